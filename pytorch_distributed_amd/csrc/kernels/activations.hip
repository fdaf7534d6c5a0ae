// GELU (tanh approximation, "gelu_new") and SwiGLU forward / backward for gfx950.
//
// Pure streaming kernels: 16-byte vector accesses, fp32 math in registers. The GELU kernels
// launch ONE 16-byte vector per thread (a grid of tens of thousands of workgroups) and mark
// their loads and stores non-temporal: on MI355X that streams at 6.1-6.5 TB/s where a grid capped
// at 8 workgroups per CU with a grid-stride loop reaches 5.0-5.3 TB/s (backward 201 -> 155 us,
// forward 126 -> 107 us on a [32768, 5120] bf16 activation; tools/micro/stream_probe.hip —
// more vectors in flight per thread, contiguous chunks per workgroup and packed fp32 math all
// measured slower or equal). The backward kernels recompute the activation from the saved
// pre-activation, so the forward never has to keep tanh / sigmoid values.
//
// The activation math lives in gelu_math.h (shared with the GEMM epilogues of gemm_fused.hip).
#include "common.h"
#include "gelu_math.h"
#include "pda_kernels.h"

namespace pda {
namespace {

constexpr int kThreads = 256;
// one vector per thread
inline unsigned flat_grid(int64_t nvec) {
  const int64_t blocks = ceil_div(nvec, kThreads);
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ y,
                                                            int64_t n) {
  constexpr int kVec = VecIO<T>::kVec;
  const int64_t i = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * kVec;
  if (i + kVec <= n) {
    float a[kVec], o[kVec];
    VecIO<T>::unpack(VecIO<T>::load_raw_nt(u + i), a);
#pragma unroll
    for (int k = 0; k < kVec; ++k) o[k] = gelu_f(a[k]);
    VecIO<T>::store_raw_nt(y + i, VecIO<T>::pack(o));
  } else {
    for (int64_t j = i; j < n; ++j) VecIO<T>::store1(y + j, gelu_f(VecIO<T>::load1(u + j)));
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gelu_bwd_kernel(const T* __restrict__ dy,
                                                            const T* __restrict__ u,
                                                            T* __restrict__ du, int64_t n) {
  constexpr int kVec = VecIO<T>::kVec;
  const int64_t i = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * kVec;
  if (i + kVec <= n) {
    float a[kVec], g[kVec], o[kVec];
    const typename VecIO<T>::Raw ur = VecIO<T>::load_raw_nt(u + i), gr = VecIO<T>::load_raw_nt(dy + i);
    VecIO<T>::unpack(ur, a);
    VecIO<T>::unpack(gr, g);
#pragma unroll
    for (int k = 0; k < kVec; ++k) o[k] = g[k] * gelu_grad_f(a[k]);
    VecIO<T>::store_raw_nt(du + i, VecIO<T>::pack(o));
  } else {
    for (int64_t j = i; j < n; ++j)
      VecIO<T>::store1(du + j, VecIO<T>::load1(dy + j) * gelu_grad_f(VecIO<T>::load1(u + j)));
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
    const typename VecIO<T>::Raw gr = VecIO<T>::load_raw_nt(x + r * 2 * f + c), ur = VecIO<T>::load_raw_nt(x + r * 2 * f + f + c);
    VecIO<T>::unpack(gr, g);
    VecIO<T>::unpack(ur, up);
#pragma unroll
    for (int k = 0; k < kVec; ++k) o[k] = g[k] * sigmoid_f(g[k]) * up[k];
    VecIO<T>::store_raw_nt(y + r * f + c, VecIO<T>::pack(o));
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
    const typename VecIO<T>::Raw gr = VecIO<T>::load_raw_nt(x + r * 2 * f + c), ur = VecIO<T>::load_raw_nt(x + r * 2 * f + f + c),
                                 dr = VecIO<T>::load_raw_nt(dy + r * f + c);
    VecIO<T>::unpack(gr, g);
    VecIO<T>::unpack(ur, up);
    VecIO<T>::unpack(dr, d);
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      const float s = sigmoid_f(g[k]);
      const float silu = g[k] * s;
      dup[k] = d[k] * silu;
      dg[k] = d[k] * up[k] * (s + silu * (1.f - s));
    }
    VecIO<T>::store_raw_nt(dx + r * 2 * f + c, VecIO<T>::pack(dg));
    VecIO<T>::store_raw_nt(dx + r * 2 * f + f + c, VecIO<T>::pack(dup));
  }
}

}  // namespace

void gelu_forward(DType dtype, const void* u, void* y, int64_t n, hipStream_t stream) {
  if (n <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(gelu_fwd_kernel<bf16_raw>, dim3(flat_grid(ceil_div(n, 8))), dim3(kThreads), 0,
                       stream, (const bf16_raw*)u, (bf16_raw*)y, n);
  else
    hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(flat_grid(ceil_div(n, 4))), dim3(kThreads), 0,
                       stream, (const float*)u, (float*)y, n);
}

void gelu_backward(DType dtype, const void* dy, const void* u, void* du, int64_t n, hipStream_t stream) {
  if (n <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16_raw>, dim3(flat_grid(ceil_div(n, 8))), dim3(kThreads), 0,
                       stream, (const bf16_raw*)dy, (const bf16_raw*)u, (bf16_raw*)du, n);
  else
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(flat_grid(ceil_div(n, 4))), dim3(kThreads), 0,
                       stream, (const float*)dy, (const float*)u, (float*)du, n);
}

void swiglu_forward(DType dtype, const void* x, void* y, int64_t rows, int f, hipStream_t stream) {
  if (rows <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(swiglu_fwd_kernel<bf16_raw>, dim3(flat_grid(rows * (f / 8))), dim3(kThreads),
                       0, stream, (const bf16_raw*)x, (bf16_raw*)y, rows, f);
  else
    hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(flat_grid(rows * (f / 4))), dim3(kThreads), 0,
                       stream, (const float*)x, (float*)y, rows, f);
}

void swiglu_backward(DType dtype, const void* dy, const void* x, void* dx, int64_t rows, int f,
                     hipStream_t stream) {
  if (rows <= 0) return;
  if (dtype == kBF16)
    hipLaunchKernelGGL(swiglu_bwd_kernel<bf16_raw>, dim3(flat_grid(rows * (f / 8))), dim3(kThreads),
                       0, stream, (const bf16_raw*)dy, (const bf16_raw*)x, (bf16_raw*)dx, rows, f);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(flat_grid(rows * (f / 4))), dim3(kThreads), 0,
                       stream, (const float*)dy, (const float*)x, (float*)dx, rows, f);
}

}  // namespace pda
