// Operand-feed probe for gfx950: how many bytes per cycle one CU can pull into LDS with
// global_load_lds when nothing else runs, as a function of where the data comes from and of how
// many 32 KiB stages are in flight. One 256-thread workgroup per CU (128 KiB of LDS keeps a second
// one out), each wave issues 8 x 1 KiB loads per stage, exactly the traffic of one k-stage of the
// 256 x 256 weight-gradient tile.
//   hit     all workgroups of an XCD walk the same 1 MiB region over and over (L2 hits)
//   share5  groups of 5 workgroups of an XCD walk one long stream together (first touch misses, rest hit)
//   miss    every workgroup walks a long stream of its own (every line comes from HBM / MALL)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/feed_probe tools/micro/feed_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int kStageBytes = 32 * 1024;
constexpr int kSlots = 4;

__device__ __forceinline__ void load_1k(const char* src, char* dst) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)dst, 16, 0, 0);
}

// stream_bytes: length of the stream one workgroup walks (wraps around); every stage advances by 32 KiB.
// kDepth: stages in flight (1..4).
template <int kDepth>
__global__ __launch_bounds__(256) void feed(const char* base, int64_t group_stride, int group_size, int64_t stream_bytes,
                                            int stages, unsigned long long* cycles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int group = within / group_size;
  const char* stream = base + (int64_t(xcd) * 64 + group) * group_stride;  // 64 groups per XCD at most
  const int64_t n_wrap = stream_bytes / kStageBytes;
  const int64_t lane_off = int64_t(wave) * 8192 + lane * 16;

  auto issue = [&](int s) {
    const char* src = stream + (s % n_wrap) * kStageBytes + lane_off;
    char* dst = smem + (s % kSlots) * kStageBytes + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) load_1k(src + i * 1024, dst + i * 1024);
  };
  for (int s = 0; s < kDepth - 1 && s < stages; ++s) issue(s);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < stages; ++s) {
    if (s + kDepth - 1 < stages) issue(s + kDepth - 1);
    else {
      // keep the queue length constant at the tail so that the counted wait below stays valid
#pragma unroll
      for (int i = 0; i < 8; ++i) load_1k(stream + lane_off + i * 1024, smem + ((s + kDepth - 1) % kSlots) * kStageBytes + wave * 8192 + i * 1024);
    }
    if (kDepth == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kDepth == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (kDepth == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (kDepth == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the consumer's view: a stage is usable once every wave's part has landed
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (stages < 0) sink[tid] = ((float*)smem)[tid];
}

// The weight-gradient kernel's addressing: a stage is 32 rows x 512 B of each of two operands, one
// instruction moves 2 rows x 512 B, rows are `pitch` bytes apart. All workgroups of an XCD walk the
// same 64 stages over and over (L2 hits), so only the row pitch changes between runs.
__global__ __launch_bounds__(256) void feed_rows(const char* base, int64_t pitch, int stages, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x % 8;
  const char* region = base + int64_t(xcd) * 2048 * pitch;
  auto issue = [&](int s) {
    char* dst = smem + (s % kSlots) * kStageBytes + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = ((i & 3) * 4 + wave) * 2 + (lane >> 5);
      const char* src = region + (int64_t(s % 64) * 32 + row) * pitch + (i >> 2) * 512 + (lane & 31) * 16;
      load_1k(src, dst + i * 1024);
    }
  };
  issue(0);
  issue(1);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < stages; ++s) {
    issue(s + 2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

void run_rows(const char* buf, int64_t pitch, unsigned long long* cyc) {
  const int stages = 1024;
  CHECK(hipFuncSetAttribute((const void*)feed_rows, hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kStageBytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(feed_rows, dim3(256), dim3(256), kSlots * kStageBytes, 0, buf, pitch, stages, cyc);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(feed_rows, dim3(256), dim3(256), kSlots * kStageBytes, 0, buf, pitch, stages, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[256];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += double(h[i]) / 256;
  printf("rows     pitch %6lld B: %7.1f s_memtime ticks per 32 KiB stage, %5.1f B/ns per CU\n", (long long)pitch, mean / stages,
         double(stages) * kStageBytes / (ms * 1e6));
}

template <int kDepth>
void run(const char* name, const char* buf, int64_t group_stride, int group_size, int64_t stream_bytes, int stages,
         unsigned long long* cyc, float* sink) {
  CHECK(hipFuncSetAttribute((const void*)feed<kDepth>, hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kStageBytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(feed<kDepth>, dim3(256), dim3(256), kSlots * kStageBytes, 0, buf, group_stride, group_size, stream_bytes,
                     stages, cyc, sink);  // warm
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(feed<kDepth>, dim3(256), dim3(256), kSlots * kStageBytes, 0, buf, group_stride, group_size, stream_bytes,
                     stages, cyc, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[256];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += double(h[i]) / 256;
  const double bytes = 256.0 * stages * kStageBytes;
  printf("%-8s depth %d: %7.1f s_memtime ticks per 32 KiB stage, %6.2f TB/s over the chip, %5.1f B/ns per CU\n", name, kDepth,
         mean / stages, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e6));
}

int main() {
  // 8 XCDs x 64 groups x 48 MiB would be 24 GiB: size the buffer for the largest layout used below instead.
  const int64_t stream = 48ll << 20;       // 1,536 stages before a private stream wraps
  const int64_t total = 8ll * 64 * stream / 2;  // miss mode: 32 groups per XCD (group_size 1); stride = stream
  char* buf;
  unsigned long long* cyc;
  float* sink;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMemset(buf, 1, total));
  CHECK(hipMalloc(&cyc, 256 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 1024));
  const int stages = 1024;
  // hit: one group of 32 per XCD, 1 MiB stream
#define ALL_DEPTHS(NAME, STRIDE, GSIZE, STREAM)                                     \
  run<1>(NAME, buf, STRIDE, GSIZE, STREAM, stages, cyc, sink);                       \
  run<2>(NAME, buf, STRIDE, GSIZE, STREAM, stages, cyc, sink);                       \
  run<3>(NAME, buf, STRIDE, GSIZE, STREAM, stages, cyc, sink);                       \
  run<4>(NAME, buf, STRIDE, GSIZE, STREAM, stages, cyc, sink)
  ALL_DEPTHS("hit", stream / 2, 32, 1ll << 20);
  ALL_DEPTHS("share5", stream / 2, 5, stream / 2);
  ALL_DEPTHS("miss", stream / 2, 1, stream / 2);
  for (int64_t pitch : {1024, 1536, 2048, 2560, 7680, 8192, 10240, 10368, 12800}) run_rows(buf, pitch, cyc);
  return 0;
}
