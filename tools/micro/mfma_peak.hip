// MFMA issue-rate microbenchmark for gfx950: register-only loops of v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_16x16x32_bf16 with 1, 2 or 4 waves per SIMD. Prints achieved TFLOP/s and the
// implied cycles per instruction at the measured wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int kAcc>
__global__ __launch_bounds__(1024) void mfma32_loop(float* out, int iters, int rnd) {
  bf16x8 a, b;
  unsigned seed = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    seed = seed * 1664525u + 1013904223u; a[i] = rnd ? (__bf16)(((seed >> 8) & 0xffff) * (1.f / 32768.f) - 1.f) : (__bf16)1.f;
    seed = seed * 1664525u + 1013904223u; b[i] = rnd ? (__bf16)(((seed >> 8) & 0xffff) * (1.f / 32768.f) - 1.f) : (__bf16)0.f;
  }
  f32x16 acc[kAcc];
  for (int k = 0; k < kAcc; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < kAcc; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kAcc>
__global__ __launch_bounds__(1024) void mfma16_loop(float* out, int iters, int rnd) {
  bf16x8 a, b;
  unsigned seed = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    seed = seed * 1664525u + 1013904223u; a[i] = rnd ? (__bf16)(((seed >> 8) & 0xffff) * (1.f / 32768.f) - 1.f) : (__bf16)1.f;
    seed = seed * 1664525u + 1013904223u; b[i] = rnd ? (__bf16)(((seed >> 8) & 0xffff) * (1.f / 32768.f) - 1.f) : (__bf16)0.f;
  }
  f32x4 acc[kAcc];
  for (int k = 0; k < kAcc; ++k) for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < kAcc; ++k) for (int e = 0; e < 4; ++e) s += acc[k][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops_per_inst, int insts_per_iter, int waves_per_cu, int rnd) {
  const int iters = 40000;
  float* out;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, 100, rnd);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, iters, rnd);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = double(iters) * insts_per_iter * waves_per_cu * 256;
  const double tf = insts * flops_per_inst / (ms * 1e-3) / 1e12;
  const double ns_per_inst_per_simd = ms * 1e6 / (double(iters) * insts_per_iter * waves_per_cu / 4.0);
  printf("%-28s %8.1f TFLOP/s  %6.2f ns per instruction per SIMD\n", name, tf, ns_per_inst_per_simd);
  hipFree(out);
}

int main() {
  for (int rnd : {0, 1})
  for (int wpc : {4, 8}) {
    printf("operands: %s\n", rnd ? "random" : "ones x zeros");
    char name[64];
    snprintf(name, sizeof name, "32x32x16 bf16, %d waves/SIMD", wpc / 4);
    run(name, [&](float* o, int it, int rnd) { hipLaunchKernelGGL(mfma32_loop<4>, dim3(256), dim3(64 * wpc), 0, 0, o, it, rnd); },
        32.0 * 32 * 16 * 2, 4, wpc, rnd);
    snprintf(name, sizeof name, "16x16x32 bf16, %d waves/SIMD", wpc / 4);
    run(name, [&](float* o, int it, int rnd) { hipLaunchKernelGGL(mfma16_loop<8>, dim3(256), dim3(64 * wpc), 0, 0, o, it, rnd); },
        16.0 * 16 * 32 * 2, 8, wpc, rnd);
  }
  return 0;
}
