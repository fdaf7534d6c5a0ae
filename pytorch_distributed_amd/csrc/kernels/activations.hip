// GELU (tanh approximation, "gelu_new") and SwiGLU forward / backward for gfx950.
//
// Pure streaming kernels: 16-byte vector accesses, fp32 math in registers,
// grid capped at 8 workgroups per CU with a grid-stride loop. The backward
// kernels recompute the activation from the saved pre-activation, so the
// forward never has to keep tanh / sigmoid values.
//
// The activation math lives in gelu_math.h (shared with the GEMM epilogues of gemm_fused.hip).
#include "common.h"
#include "gelu_math.h"
#include "pda_kernels.h"

namespace pda {
namespace {

constexpr int kThreads = 256;
inline unsigned stream_grid(int64_t nvec) {
  int64_t blocks = ceil_div(nvec, kThreads);
  if (blocks > 2048) blocks = 2048;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ y,
                                                            int64_t n) {
  constexpr int kVec = VecIO<T>::kVec;
  const int64_t stride = int64_t(gridDim.x) * kThreads * kVec;
  for (int64_t i = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * kVec; i < n; i += stride) {
    if (i + kVec <= n) {
      float a[kVec], o[kVec];
      VecIO<T>::load(u + i, a);
#pragma unroll
      for (int k = 0; k < kVec; ++k) o[k] = gelu_f(a[k]);
      VecIO<T>::store(y + i, o);
    } else {
      for (int64_t j = i; j < n; ++j) VecIO<T>::store1(y + j, gelu_f(VecIO<T>::load1(u + j)));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gelu_bwd_kernel(const T* __restrict__ dy,
                                                            const T* __restrict__ u,
                                                            T* __restrict__ du, int64_t n) {
  constexpr int kVec = VecIO<T>::kVec;
  const int64_t stride = int64_t(gridDim.x) * kThreads * kVec;
  for (int64_t i = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * kVec; i < n; i += stride) {
    if (i + kVec <= n) {
      float a[kVec], g[kVec], o[kVec];
      VecIO<T>::load(u + i, a);
      VecIO<T>::load(dy + i, g);
#pragma unroll
      for (int k = 0; k < kVec; ++k) o[k] = g[k] * gelu_grad_f(a[k]);
      VecIO<T>::store(du + i, o);
    } else {
      for (int64_t j = i; j < n; ++j)
        VecIO<T>::store1(du + j, VecIO<T>::load1(dy + j) * gelu_grad_f(VecIO<T>::load1(u + j)));
    }
  }
}

// x is [rows, 2f] = [gate | up]; y is [rows, f]. f % kVec == 0.
template <typename T>
__global__ __launch_bounds__(kThreads) void swiglu_fwd_kernel(const T* __restrict__ x,
                                                              T* __restrict__ y, int64_t rows, int f) {
  constexpr int kVec = VecIO<T>::kVec;
  const int fv = f / kVec;
  const int64_t total = rows * fv;
  const int64_t stride = int64_t(gridDim.x) * kThreads;
  for (int64_t v = int64_t(blockIdx.x) * kThreads + threadIdx.x; v < total; v += stride) {
    const int64_t r = v / fv;
    const int c = (int)(v - r * fv) * kVec;
    float g[kVec], up[kVec], o[kVec];
    VecIO<T>::load(x + r * 2 * f + c, g);
    VecIO<T>::load(x + r * 2 * f + f + c, up);
#pragma unroll
    for (int k = 0; k < kVec; ++k) o[k] = g[k] * sigmoid_f(g[k]) * up[k];
    VecIO<T>::store(y + r * f + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void swiglu_bwd_kernel(const T* __restrict__ dy,
                                                              const T* __restrict__ x,
                                                              T* __restrict__ dx, int64_t rows, int f) {
  constexpr int kVec = VecIO<T>::kVec;
  const int fv = f / kVec;
  const int64_t total = rows * fv;
  const int64_t stride = int64_t(gridDim.x) * kThreads;
  for (int64_t v = int64_t(blockIdx.x) * kThreads + threadIdx.x; v < total; v += stride) {
    const int64_t r = v / fv;
    const int c = (int)(v - r * fv) * kVec;
    float g[kVec], up[kVec], d[kVec], dg[kVec], dup[kVec];
    VecIO<T>::load(x + r * 2 * f + c, g);
    VecIO<T>::load(x + r * 2 * f + f + c, up);
    VecIO<T>::load(dy + r * f + c, d);
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      const float s = sigmoid_f(g[k]);
      const float silu = g[k] * s;
      dup[k] = d[k] * silu;
      dg[k] = d[k] * up[k] * (s + silu * (1.f - s));
    }
    VecIO<T>::store(dx + r * 2 * f + c, dg);
    VecIO<T>::store(dx + r * 2 * f + f + c, dup);
  }
}

}  // namespace

void gelu_forward(DType dtype, const void* u, void* y, int64_t n, hipStream_t stream) {
  if (n <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(gelu_fwd_kernel<bf16_raw>, dim3(stream_grid(ceil_div(n, 8))), dim3(kThreads), 0,
                       stream, (const bf16_raw*)u, (bf16_raw*)y, n);
  else
    hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(stream_grid(ceil_div(n, 4))), dim3(kThreads), 0,
                       stream, (const float*)u, (float*)y, n);
}

void gelu_backward(DType dtype, const void* dy, const void* u, void* du, int64_t n, hipStream_t stream) {
  if (n <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16_raw>, dim3(stream_grid(ceil_div(n, 8))), dim3(kThreads), 0,
                       stream, (const bf16_raw*)dy, (const bf16_raw*)u, (bf16_raw*)du, n);
  else
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(stream_grid(ceil_div(n, 4))), dim3(kThreads), 0,
                       stream, (const float*)dy, (const float*)u, (float*)du, n);
}

void swiglu_forward(DType dtype, const void* x, void* y, int64_t rows, int f, hipStream_t stream) {
  if (rows <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(swiglu_fwd_kernel<bf16_raw>, dim3(stream_grid(rows * (f / 8))), dim3(kThreads),
                       0, stream, (const bf16_raw*)x, (bf16_raw*)y, rows, f);
  else
    hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(stream_grid(rows * (f / 4))), dim3(kThreads), 0,
                       stream, (const float*)x, (float*)y, rows, f);
}

void swiglu_backward(DType dtype, const void* dy, const void* x, void* dx, int64_t rows, int f,
                     hipStream_t stream) {
  if (rows <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(swiglu_bwd_kernel<bf16_raw>, dim3(stream_grid(rows * (f / 8))), dim3(kThreads),
                       0, stream, (const bf16_raw*)dy, (const bf16_raw*)x, (bf16_raw*)dx, rows, f);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(stream_grid(rows * (f / 4))), dim3(kThreads), 0,
                       stream, (const float*)dy, (const float*)x, (float*)dx, rows, f);
}

}  // namespace pda
