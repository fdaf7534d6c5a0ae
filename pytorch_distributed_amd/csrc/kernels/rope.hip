// Rotary position embedding on the packed QKV activation for gfx950.
//
// qkv is [B, T, 3, H, D]; the Q and K thirds are rotated pair-wise in the "rotate half"
// convention — (x[i], x[i + D/2]) -> (x[i] cos - x[i + D/2] sin, x[i + D/2] cos + x[i] sin) with
// the angle of (position t, frequency i) taken from fp32 tables cos / sin [T, D/2] — and the V
// third passes through. One streaming kernel, 16-byte vector accesses, fp32 math; in place
// when src == dst (then V is not touched at all). The backward of a rotation is the rotation
// by the opposite angle, so the same kernel serves both directions (`inverse`).
//
// Replaces ~10 element-wise PyTorch kernels per layer and direction (slice, cast to fp32,
// four multiplies, add / subtract, concatenate, cast back, copy of V), which were 13 % of a
// Llama-1B training step.
#include "common.h"
#include "pda_kernels.h"

namespace pda {
namespace {

constexpr int kThreads = 256;

// A workgroup walks tokens (blockIdx.x, grid-stride); its threads own the vector pairs of a token.
// The decomposition of a pair index into (section, head, frequency) does not depend on the
// token, so it is done once per thread outside the token loop, and every index inside it is
// 32-bit apart from one 64-bit multiply-add for the token's row address (a flat 64-bit element
// index, divided per element, had made this streaming kernel VALU-bound at 40 % of HBM speed).
template <typename T>
__global__ __launch_bounds__(kThreads) void rope_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                        const float* __restrict__ cos_tab,
                                                        const float* __restrict__ sin_tab, int tokens, int T_len,
                                                        int H, int D, float sign, bool copy_v) {
  constexpr int kVec = VecIO<T>::kVec;
  const int half = D / 2;
  const int vec_per_head = half / kVec;            // vector pairs per (section, head)
  const int vec_per_token = 2 * H * vec_per_head;  // Q and K sections
  const int64_t row = int64_t(3) * H * D;
  for (int v = threadIdx.x; v < vec_per_token; v += kThreads) {
    const int section = v / (H * vec_per_head);
    const int rem = v - section * H * vec_per_head;
    const int head = rem / vec_per_head;
    const int i0 = (rem - head * vec_per_head) * kVec;
    const int in_token = (section * H + head) * D + i0;
    for (int tok = blockIdx.x; tok < tokens; tok += gridDim.x) {
      const int t = tok % T_len;  // workgroup-uniform
      const int64_t off = int64_t(tok) * row + in_token;
      float x1[kVec], x2[kVec], c[kVec], s[kVec], y1[kVec], y2[kVec];
      VecIO<T>::load(src + off, x1);
      VecIO<T>::load(src + off + half, x2);
#pragma unroll
      for (int k = 0; k < kVec; k += 4) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(cos_tab + t * half + i0 + k);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sin_tab + t * half + i0 + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          c[k + e] = cv[e];
          s[k + e] = sv[e] * sign;
        }
      }
#pragma unroll
      for (int k = 0; k < kVec; ++k) {
        y1[k] = x1[k] * c[k] - x2[k] * s[k];
        y2[k] = x2[k] * c[k] + x1[k] * s[k];
      }
      VecIO<T>::store(dst + off, y1);
      VecIO<T>::store(dst + off + half, y2);
    }
  }
  if (copy_v) {  // out of place: the V third is copied through
    const int v_vecs = H * D / kVec;
    for (int v = threadIdx.x; v < v_vecs; v += kThreads)
      for (int tok = blockIdx.x; tok < tokens; tok += gridDim.x) {
        const int64_t off = int64_t(tok) * row + int64_t(2) * H * D + v * kVec;
        *reinterpret_cast<typename VecIO<T>::Raw*>(dst + off) = VecIO<T>::load_raw(src + off);
      }
  }
}

}  // namespace

void rope_apply(DType dtype, const void* src, void* dst, const float* cos_tab, const float* sin_tab, int B, int T,
                int H, int D, bool inverse, hipStream_t stream) {
  const int64_t tokens = int64_t(B) * T;  // the binding checks tokens < 2^31 and T * D / 2 < 2^31
  int64_t blocks = tokens < 4096 ? tokens : 4096;
  if (blocks < 1) return;
  const float sign = inverse ? -1.f : 1.f;
  const bool copy_v = src != dst;
  if (dtype == kBF16)
    hipLaunchKernelGGL(rope_kernel<bf16_raw>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, (const bf16_raw*)src,
                       (bf16_raw*)dst, cos_tab, sin_tab, (int)tokens, T, H, D, sign, copy_v);
  else
    hipLaunchKernelGGL(rope_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, (const float*)src,
                       (float*)dst, cos_tab, sin_tab, (int)tokens, T, H, D, sign, copy_v);
}

}  // namespace pda
