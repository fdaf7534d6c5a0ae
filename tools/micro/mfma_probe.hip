// Matrix-pipe issue probe for gfx950: clock ticks (s_memtime) per v_mfma_f32_32x32x16_bf16 and per
// v_mfma_f32_16x16x32_bf16 when one wave per SIMD issues independent MFMAs back to back, and when two
// waves per SIMD do. One workgroup per CU, 8 independent accumulators per wave, operands in registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/mfma_probe tools/micro/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// kExtra: what is issued between two MFMAs: 0 nothing, 1 two ds_read_b64 (consumed a loop trip later), 2 two v_add_u32,
// 3 four s_add_u32, 4 one ds_read_b64, 5 four ds_read_b64, 6 four v_exp_f32, 7 two v_pk_mul_f32 + two v_cvt_pk_bf16_f32,
// 8 all eight of 6 and 7
template <int kShape, int kExtra>  // 32: 32x32x16, 16: 16x16x32
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cycles, int iters) {
  using acc_t = typename std::conditional<kShape >= 32, f32x16, f32x4>::type;  // 33: 32x32x16 with the accumulators in AGPRs
  constexpr int kElems = kShape >= 32 ? 16 : 4;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (threadIdx.x + i));
    b[i] = (__bf16)(1.0f + 0.01f * i);
  }
  acc_t acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < kElems; ++e) acc[k][e] = 0.f;
  __shared__ uint64_t lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x + 512] = threadIdx.x;
  const unsigned lds_at = (unsigned)(uintptr_t)(lds + (threadIdx.x & 63));
  uint64_t rd[4] = {0, 0, 0, 0};
  unsigned va = threadIdx.x, vb = 1, sa = iters, sb = 3;
  float fx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fy[4] = {-0.5f, -1.5f, -2.5f, -0.25f * threadIdx.x};
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 px[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, py[2] = {{1.5f, 2.5f}, {0.5f, 0.25f}};
  unsigned cv[4] = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (kShape == 33) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
      else if constexpr (kShape == 32) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
      else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
      if constexpr (kExtra == 1 || kExtra == 4 || kExtra == 5) {
        constexpr int n = kExtra == 1 ? 2 : kExtra == 4 ? 1 : 4;
#pragma unroll
        for (int q = 0; q < n; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rd[q]) : "v"(lds_at), "n"(512 * 0) : "memory");
      }
      if constexpr (kExtra == 2) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(va) : "v"(vb));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(vb) : "v"(va));
      }
      if constexpr (kExtra == 6) {  // four independent v_exp_f32
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("v_exp_f32 %0, %1" : "=v"(fx[4 * (k & 1) + q]) : "v"(fy[q]));
      }
      if constexpr (kExtra == 7) {  // two v_pk_mul_f32 and two v_cvt_pk_bf16_f32, all to different registers
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(px[2 * (k & 1)]) : "v"(py[0]), "v"(py[1]));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(px[2 * (k & 1) + 1]) : "v"(py[1]), "v"(py[0]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[2 * (k & 1)]) : "v"(fy[0]), "v"(fy[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[2 * (k & 1) + 1]) : "v"(fy[2]), "v"(fy[3]));
      }
      if constexpr (kExtra == 8) {  // eight: 4 v_exp_f32, 2 v_pk_mul_f32, 2 v_cvt_pk_bf16_f32
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("v_exp_f32 %0, %1" : "=v"(fx[4 * (k & 1) + q]) : "v"(fy[q]));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(px[2 * (k & 1)]) : "v"(py[0]), "v"(py[1]));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(px[2 * (k & 1) + 1]) : "v"(py[1]), "v"(py[0]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[2 * (k & 1)]) : "v"(fy[0]), "v"(fy[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[2 * (k & 1) + 1]) : "v"(fy[2]), "v"(fy[3]));
      }
      if constexpr (kExtra == 9) {  // eight v_fma_f32 to different registers
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(fx[q]) : "v"(fy[q & 3]), "v"(fy[(q + 1) & 3]), "v"(fy[(q + 2) & 3]));
      }
      if constexpr (kExtra == 3) {
        asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0" : "+s"(sa), "+s"(sb));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kExtra == 1 || kExtra == 4 || kExtra == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = float(rd[0] + rd[1] + rd[2] + rd[3]) + float(va + vb + sa + sb);
#pragma unroll
  for (int q = 0; q < 8; ++q) s += fx[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) s += px[q][0] + px[q][1] + float(cv[q]);
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][kElems - 1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0 && blockIdx.x == 0) cycles[threadIdx.x / 64] = t1 - t0;
}

template <int kShape, int kExtra>
void run(const char* what, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  for (int waves = 4; waves <= 8; waves += 4) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<kShape, kExtra>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<kShape, kExtra>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hh[8];
    CHECK(hipMemcpy(hh, cyc, 8 * waves, hipMemcpyDeviceToHost));
    const unsigned long long h = hh[0], h_last = hh[waves - 1];
    const double n = 8.0 * iters;  // MFMAs per wave
    const double flop = (kShape >= 32 ? 32768.0 : 16384.0) * n * waves * 256;
    printf("%-13s + %-22s %d wave(s) per SIMD: %6.2f ticks per MFMA in wave 0, %6.2f in the last wave, %6.0f TFLOP/s\n",
           kShape == 33 ? "32x32x16 AGPR" : kShape == 32 ? "32x32x16" : "16x16x32", what, waves / 4, double(h) / n, double(h_last) / n, flop / (ms * 1e-3) / 1e12);
  }
}

int main() {
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 64));
  run<32, 0>("nothing,", out, cyc);
  run<32, 4>("1 ds_read_b64,", out, cyc);
  run<32, 1>("2 ds_read_b64,", out, cyc);
  run<32, 5>("4 ds_read_b64,", out, cyc);
  run<32, 2>("2 v_add_u32,", out, cyc);
  run<32, 3>("4 s_add_u32,", out, cyc);
  run<32, 6>("4 v_exp_f32,", out, cyc);
  run<32, 7>("2 pk_mul + 2 cvt_pk,", out, cyc);
  run<32, 8>("4 exp+2 pk_mul+2 cvt,", out, cyc);
  run<32, 9>("8 v_fma_f32,", out, cyc);
  run<33, 0>("nothing,", out, cyc);
  run<33, 2>("2 v_add_u32,", out, cyc);
  run<33, 9>("8 v_fma_f32,", out, cyc);
  run<33, 8>("4 exp+2 pk_mul+2 cvt,", out, cyc);
  run<33, 1>("2 ds_read_b64,", out, cyc);
  run<16, 0>("nothing,", out, cyc);
  run<16, 4>("1 ds_read_b64,", out, cyc);
  run<16, 1>("2 ds_read_b64,", out, cyc);
  run<16, 2>("2 v_add_u32,", out, cyc);
  return 0;
}
