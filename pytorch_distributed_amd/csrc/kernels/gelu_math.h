// tanh-approximated GELU ("gelu_new") and its derivative in the form the gfx950 kernels use:
//   0.5 (1 + tanh(z)) = sigmoid(2z), so gelu_new(u) = u * sigmoid(2 sqrt(2/pi) (u + c u^3)),
// with sigmoid(x) = 1 / (1 + exp2(-x log2 e)) on the hardware exp2 and reciprocal (1 ulp each; the
// result is rounded to bf16): nine VALU instructions per element where tanh through __expf and an
// IEEE division took twenty-four. exp2 overflow (very negative x) gives 1 / inf = 0 and underflow
// gives 1, the correct limits. Shared by the stand-alone activation kernels and the GEMM epilogues.
//
// gelu_new is the activation every GPT-2 config selects through ACT2FN[config.activation_function]
// (reference: assignments/assignment1/model/my_gpt2.py:90,96).
#pragma once

#include <hip/hip_runtime.h>

namespace pda {

constexpr float kGeluSqrt2OverPi = 0.7978845608028654f;
constexpr float kGeluC = 0.044715f;
constexpr float kGeluLog2e = 1.4426950408889634f;

__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kGeluLog2e * x));
}

__device__ __forceinline__ float gelu_f(float u) {
  const float u2 = u * u;
  // exp2 argument: -2 log2(e) sqrt(2/pi) (u + c u^3) = u * (a + b u^2)
  constexpr float a = -2.f * kGeluLog2e * kGeluSqrt2OverPi, b = a * kGeluC;
  return u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * (a + b * u2)));
}

// d/du [u s(u)] with s = sigmoid(2 inner(u)): s + u s (1 - s) 2 inner'(u),
// inner'(u) = sqrt(2/pi) (1 + 3 c u^2).
__device__ __forceinline__ float gelu_grad_f(float u) {
  const float u2 = u * u;
  constexpr float a = -2.f * kGeluLog2e * kGeluSqrt2OverPi, b = a * kGeluC;
  const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * (a + b * u2)));
  const float dinner2 = 2.f * kGeluSqrt2OverPi + (6.f * kGeluSqrt2OverPi * kGeluC) * u2;
  return s + u * (s - s * s) * dinner2;
}

}  // namespace pda
