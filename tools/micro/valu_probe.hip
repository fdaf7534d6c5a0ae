// Vector-instruction cost probe for gfx950: cycles (s_memtime) per iteration of the element-wise
// part of an attention-backward tile — 32 v_exp_f32, 16 packed (or 32 scalar) multiplies, 16
// v_cvt_pk_bf16_f32 — run from registers by 1, 2 or 3 waves per SIMD, one workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/valu_probe tools/micro/valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

// kMode: 0 = 32 exp; 1 = 32 exp + 16 pk_mul; 2 = 32 exp + 32 scalar mul; 3 = mode 1 + 16 cvt_pk; 4 = 32 fma only;
// 5 = 16 pk_mul only; 6 = 16 cvt only
template <int kMode>
__global__ void probe(float* out, unsigned long long* cycles, int iters) {
  float s[32], d[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    s[i] = -0.001f * (threadIdx.x + i);
    d[i] = 1.0f + 0.001f * i;
  }
  uint32_t acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float p[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (kMode <= 3) p[i] = __builtin_amdgcn_exp2f(s[i]);
      else if (kMode == 4) p[i] = __builtin_fmaf(s[i], d[i], 0.5f);
      else p[i] = s[i];
    }
    if (kMode == 1 || kMode == 3 || kMode == 5) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        const f32x2 r = f32x2{p[i], p[i + 1]} * f32x2{d[i], d[i + 1]};
        p[i] = r[0];
        p[i + 1] = r[1];
      }
    }
    if (kMode == 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(d[i]));
    }
    if (kMode == 3 || kMode == 6) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        f32x2 f = {p[i], p[i + 1]};
        bf16x2 b = __builtin_convertvector(f, bf16x2);
        acc ^= __builtin_bit_cast(uint32_t, b);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc ^= __float_as_uint(p[i]);  // every value stays live (32 v_xor on top)
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(s[i]));  // a fresh dependency per iteration, no extra instruction
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0 && blockIdx.x == 0) cycles[threadIdx.x / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc);
}

template <int kMode>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int waves = 4; waves <= 12; waves += 4) {  // 1, 2, 3 waves per SIMD
    hipLaunchKernelGGL(probe<kMode>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[16];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-44s %d wave(s)/SIMD: %7.1f cycles per iteration and wave\n", name, waves / 4, (double)h[0] / iters);
  }
}

int main() {
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 16 * sizeof(unsigned long long)));
  run<0>("32 v_exp_f32", out, cyc);
  run<1>("32 v_exp_f32 + 16 v_pk_mul_f32", out, cyc);
  run<2>("32 v_exp_f32 + 32 v_mul_f32", out, cyc);
  run<3>("32 v_exp_f32 + 16 v_pk_mul_f32 + 16 v_cvt_pk", out, cyc);
  run<4>("32 v_fma_f32", out, cyc);
  run<5>("16 v_pk_mul_f32", out, cyc);
  run<6>("16 v_cvt_pk_bf16_f32", out, cyc);
  return 0;
}
