// Where a stage of the weight-gradient kernel spends its time: compiles the shipped kernel source
// with PDA_WGRAD_STAMPS (s_memtime at four points of every stage, read only behind waits the kernel
// performs anyway) and prints the mean ticks per stage and wave in each phase.
// Build: hipcc --offload-arch=gfx950 -O3 -I pytorch_distributed_amd/csrc/kernels -o tools/micro/bin/wgrad_stamp tools/micro/wgrad_stamp.hip
// Run:   wgrad_stamp [M I J]        (default 32768 5120 1280, the c_fc gradient of GPT-2 large at 32 x 1024 tokens)
#define PDA_WGRAD_STAMPS 1
#include "../../pytorch_distributed_amd/csrc/kernels/gemm_wgrad.hip"

#include <stdio.h>
#include <vector>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void fill(unsigned short* p, int64_t n, unsigned seed) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    p[i] = (unsigned short)(0x3c00 + (h & 0x3ff));  // bf16 in [0.0078, 0.0156): any finite values do
  }
}

// Round trip of one s_memtime (issue -> value usable): what every "stamp, then wait" pair above includes.
__global__ void memtime_latency(unsigned long long* out) {
  unsigned long long sum = 0;
  for (int i = 0; i < 64; ++i) {
    const unsigned long long a = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long b = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    sum += b - a;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sum / 64;
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 32768, I = argc > 3 ? atoi(argv[2]) : 5120, J = argc > 3 ? atoi(argv[3]) : 1280;
  if (!pda::wgrad_supported(M, I, J)) { fprintf(stderr, "unsupported shape\n"); return 1; }
  const int S = pda::wgrad_pick_slices(M, I, J);
  unsigned short *A, *B;
  float *partial, *bias;
  CHECK(hipMalloc(&A, int64_t(M) * I * 2));
  CHECK(hipMalloc(&B, int64_t(M) * J * 2));
  CHECK(hipMalloc(&partial, int64_t(S) * I * J * 4));
  CHECK(hipMalloc(&bias, int64_t(S) * I * 4));
  hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, int64_t(M) * I, 1u);
  hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, B, int64_t(M) * J, 2u);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) {
    CHECK(hipEventRecord(e0));
    pda::wgrad_gemm(A, I, B, J, partial, bias, M, I, J, S, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
  }
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int n_tasks = ((I + 255) / 256) * ((J + 255) / 256) * S;
  std::vector<unsigned long long> h(1024 * 32);
  CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(pda::g_wgrad_stamps), h.size() * 8));
  const double stages = double(M / 32) / S;
  double sum[7] = {0, 0, 0, 0, 0, 0, 0};
  const int n = n_tasks < 1024 ? n_tasks : 1024;
  for (int w = 0; w < n * 4; ++w)
    for (int p = 0; p < 7; ++p) sum[p] += double(h[w * 8 + p]) / (n * 4) / (p < 4 ? stages : 1.0);
  printf("M=%d I=%d J=%d slices=%d tasks=%d: %.1f us, %.0f TFLOP/s\n", M, I, J, S, n_tasks, ms * 1e3, 2.0 * M * I * J / (ms * 1e-3) / 1e12);
  printf("ticks per stage and wave: wait+barrier %.0f | k-step 0 (16 MFMA + 16 fragment reads + 8 loads) %.0f | fragment wait %.0f | "
         "k-step 1 (16 MFMA + 16 fragment reads) %.0f | total %.0f  (MFMA alone: 1024 cycles)\n",
         sum[0], sum[1], sum[2], sum[3], sum[0] + sum[1] + sum[2] + sum[3]);
  // The counters of different XCDs are not aligned, so only differences within a wave mean anything.
  const double rounds = double((n_tasks + 255) / 256);
  printf("ticks per task and wave: start -> loop %.0f | loop %.0f | loop -> last partial store retired %.0f | %.0f round(s) of tasks: "
         "%.2f ticks per ns of kernel time\n", sum[4], sum[5], sum[6], rounds, rounds * (sum[4] + sum[5] + sum[6]) / (ms * 1e6));
  unsigned long long* lat;
  CHECK(hipMalloc(&lat, 8));
  hipLaunchKernelGGL(memtime_latency, dim3(1), dim3(64), 0, 0, lat);
  unsigned long long hl;
  CHECK(hipMemcpy(&hl, lat, 8, hipMemcpyDeviceToHost));
  printf("s_memtime round trip on an idle CU: %llu ticks (inside 'wait+barrier' and 'fragment wait' above)\n", hl);
  return 0;
}
