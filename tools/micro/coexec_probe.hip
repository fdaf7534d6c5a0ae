// Can the matrix pipe and the vector ALU of one SIMD work for two different waves at once? One
// workgroup of 8 waves per CU (2 per SIMD): waves 0-3 issue independent v_mfma_f32_32x32x16_bf16
// back to back, waves 4-7 run the element-wise part of an attention-backward tile (32 v_exp_f32,
// 16 v_pk_mul_f32, 16 v_cvt_pk_bf16_f32 per trip). Each kind is timed alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/coexec_probe tools/micro/coexec_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ float mfma_stream(int iters, unsigned long long* ticks) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (threadIdx.x + i));
    b[i] = (__bf16)(1.0f + 0.01f * i);
  }
  f32x16 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  *ticks = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][15];
  return s;
}

__device__ __forceinline__ float valu_stream(int iters, unsigned long long* ticks) {
  float s[32], d[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    s[i] = -0.001f * (threadIdx.x + i);
    d[i] = 1.0f + 0.001f * i;
  }
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float p[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) p[i] = __builtin_amdgcn_exp2f(s[i]);
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      const f32x2 r = f32x2{p[i], p[i + 1]} * f32x2{d[i], d[i + 1]};
      bf16x2 c = __builtin_convertvector(r, bf16x2);
      acc ^= __builtin_bit_cast(uint32_t, c);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(s[i]));
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  *ticks = t1 - t0;
  return __uint_as_float(acc);
}

// mode bit 0: the MFMA waves work, bit 1: the VALU waves work, bit 2: the VALU waves are the OLDER ones (0-3),
// bit 3: the VALU waves raise their priority (s_setprio 3), bit 4: the MFMA waves raise theirs
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* ticks, int mfma_iters, int valu_iters, int mode) {
  const int wave = threadIdx.x >> 6;
  unsigned long long t = 0;
  float r = 0.f;
  const bool is_mfma = (mode & 4) ? wave >= 4 : wave < 4;
  __syncthreads();
  if (is_mfma) {
    if (mode & 16) __builtin_amdgcn_s_setprio(3);
    if (mode & 1) r = mfma_stream(mfma_iters, &t);
  } else {
    if (mode & 8) __builtin_amdgcn_s_setprio(3);
    if (mode & 2) r = valu_stream(valu_iters, &t);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t;
}

int main() {
  float* out;
  unsigned long long* ticks;
  CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  CHECK(hipMalloc(&ticks, 64));
  const int mfma_iters = 4000, valu_iters = 600;  // ~1.0 M and ~0.3 M ticks alone
  struct { int mode; const char* name; } runs[] = {
      {1, "MFMA waves alone"}, {2, "VALU waves alone"}, {3, "both, MFMA waves older"}, {7, "both, VALU waves older"},
      {11, "both, MFMA older, VALU waves at priority 3"}, {23, "both, VALU older, MFMA waves at priority 3"}};
  for (auto& run : runs) {
    const int mode = run.mode;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, ticks, mfma_iters, valu_iters, mode);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, ticks, mfma_iters, valu_iters, mode);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[8];
    CHECK(hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost));
    const int mw = (mode & 4) ? 4 : 0, vw = (mode & 4) ? 0 : 4;
    printf("%-44s: %6.2f ticks per MFMA, %7.1f ticks per element-wise trip, kernel %.1f us\n", run.name,
           double(h[mw]) / (8.0 * mfma_iters), double(h[vw]) / valu_iters, ms * 1e3);
  }
  return 0;
}
