// LayerNorm / RMSNorm forward and backward for gfx950.
//
// Mapping: one row per 64-lane wave, 4 waves per workgroup. A lane owns the
// 16-byte vectors lane, lane+64, lane+128, ... of its row and keeps them in
// registers between the statistics pass and the normalise pass, so every row
// is read from HBM exactly once and written once. Row statistics are wave
// reductions (no LDS, no barrier). The optional residual add is fused into the
// forward (emits the new residual stream too) and the incoming residual
// gradient into the backward, which removes the stand-alone elementwise adds
// of the pre-norm transformer block
// (reference block structure: assignments/assignment1/model/my_gpt2.py:121-134).
//
// Backward additionally needs column sums over all rows (dweight, dbias). Each
// wave accumulates its columns in registers while it walks its rows, the four
// waves of a block are combined through LDS, and every block writes one fp32
// partial row; norm_reduce_partials folds the <= 512 partial rows.
#include "common.h"
#include "pda_kernels.h"
#include "philox.h"

namespace pda {
namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / kWave;
constexpr int kMaxPartialRows = 512;

using attn::DropoutParams;

// Dropout on the branch input x of the fused "x + residual" (the residual / embedding dropout
// of the reference, assignments/assignment1/model/my_gpt2.py:55-56,97-98,172): element k of
// 16-byte vector `vi` of row `row` is kept when byte k of Philox(row, vi; seed) >= threshold.
// Forward and backward call this with the same coordinates, so no mask is stored.
template <int kVec>
__device__ __forceinline__ uint32_t dropout_keep_bits(const DropoutParams& dp, int64_t row, int vi) {
  const u32x4 r = attn::philox4x32_7(uint32_t(row), uint32_t(uint64_t(row) >> 32), uint32_t(vi), 0x6e6f726du,
                                     dp.seed_lo, dp.seed_hi);
  uint32_t bits = 0;
#pragma unroll
  for (int k = 0; k < kVec; ++k)
    if (((r[k >> 2] >> (8 * (k & 3))) & 0xffu) >= dp.threshold) bits |= 1u << k;
  return bits;
}

template <typename T, int kMaxIt, bool kRms, bool kHasRes, bool kDrop>
__global__ __launch_bounds__(kThreads) void norm_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const T* __restrict__ w,
    const T* __restrict__ b, T* __restrict__ y, T* __restrict__ sum_out, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int64_t rows, int d, float eps, DropoutParams drop) {
  constexpr int kVec = VecIO<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int nvec = d / kVec;
  const float inv_d = 1.f / (float)d;
  const int64_t wave_global = int64_t(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t wave_stride = int64_t(gridDim.x) * kWavesPerBlock;

  for (int64_t row = wave_global; row < rows; row += wave_stride) {
    const T* xr = x + row * d;
    float v[kMaxIt][kVec];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int vi = lane + it * 64;
      if (vi < nvec) {
        VecIO<T>::load(xr + vi * kVec, v[it]);
        if (kDrop) {
          const uint32_t keep = dropout_keep_bits<kVec>(drop, row, vi);
#pragma unroll
          for (int k = 0; k < kVec; ++k) v[it][k] = (keep >> k) & 1u ? v[it][k] * drop.keep_scale : 0.f;
        }
        if (kHasRes) {
          float r[kVec];
          VecIO<T>::load(res + row * d + vi * kVec, r);
#pragma unroll
          for (int k = 0; k < kVec; ++k) v[it][k] += r[k];
        }
        if (kHasRes || kDrop) {  // the normalised row differs from x: emit it as the new residual stream
          VecIO<T>::store(sum_out + row * d + vi * kVec, v[it]);
          if constexpr (sizeof(T) == 2) {
            // Normalise what later kernels will read back: the bf16-rounded sum.
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[it][k] = bf16_to_f32(f32_to_bf16(v[it][k]));
          }
        }
#pragma unroll
        for (int k = 0; k < kVec; ++k) s += kRms ? v[it][k] * v[it][k] : v[it][k];
      }
    }
    float mean = 0.f, rstd;
    if (kRms) {
      rstd = rsqrtf(wave_sum(s) * inv_d + eps);
    } else {
      mean = wave_sum(s) * inv_d;
      float q = 0.f;
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        if (lane + it * 64 < nvec) {
#pragma unroll
          for (int k = 0; k < kVec; ++k) {
            const float c = v[it][k] - mean;
            q += c * c;
          }
        }
      }
      rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    }
    if (lane == 0) {
      if (!kRms) mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int vi = lane + it * 64;
      if (vi < nvec) {
        float wv[kVec], o[kVec];
        VecIO<T>::load(w + vi * kVec, wv);
        if (kRms) {
#pragma unroll
          for (int k = 0; k < kVec; ++k) o[k] = v[it][k] * rstd * wv[k];
        } else {
          float bv[kVec];
          VecIO<T>::load(b + vi * kVec, bv);
#pragma unroll
          for (int k = 0; k < kVec; ++k) o[k] = (v[it][k] - mean) * rstd * wv[k] + bv[k];
        }
        VecIO<T>::store(y + row * d + vi * kVec, o);
      }
    }
  }
}

// With kDrop the gradient of the normalised row goes to `dx` (the residual stream's share) and
// its masked, rescaled copy to `dx_branch` (the share of the branch that was dropped out).
template <typename T, int kMaxIt, bool kRms, bool kHasRes, bool kDrop>
__global__ __launch_bounds__(kThreads) void norm_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ w,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const T* __restrict__ dres, T* __restrict__ dx, T* __restrict__ dx_branch, float* __restrict__ dw_partial,
    float* __restrict__ db_partial, int64_t rows, int d, DropoutParams drop) {
  constexpr int kVec = VecIO<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int nvec = d / kVec;
  const float inv_d = 1.f / (float)d;
  const int64_t wave_global = int64_t(blockIdx.x) * kWavesPerBlock + wid;
  const int64_t wave_stride = int64_t(gridDim.x) * kWavesPerBlock;

  float dw_acc[kMaxIt][kVec];
  float db_acc[kRms ? 1 : kMaxIt][kVec];
  float wv[kMaxIt][kVec];
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      dw_acc[it][k] = 0.f;
      if constexpr (!kRms) db_acc[it][k] = 0.f;
      wv[it][k] = 0.f;
    }
    const int vi = lane + it * 64;
    if (vi < nvec) VecIO<T>::load(w + vi * kVec, wv[it]);
  }

  // The row loop is software-pipelined: the loads of the wave's NEXT row are issued before the
  // two reductions of the current one, so each wave keeps two rows of dy / x / dresidual in
  // flight (the grid is capped by the number of partial rows, i.e. only ~2 waves per SIMD, and a
  // load -> reduce -> store chain per row left the memory system idle half of the time).
  typedef typename VecIO<T>::Raw Raw;
  Raw dy_raw[kMaxIt], x_raw[kMaxIt], res_raw[kHasRes ? kMaxIt : 1];
  float mean = 0.f, rstd = 0.f;
  auto fetch = [&](int64_t row, Raw (&dyr)[kMaxIt], Raw (&xr)[kMaxIt], Raw (&rr)[kHasRes ? kMaxIt : 1], float& m,
                   float& rs) {
    m = kRms ? 0.f : mean_in[row];
    rs = rstd_in[row];
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int vi = lane + it * 64;
      if (vi < nvec) {
        dyr[it] = VecIO<T>::load_raw(dy + row * d + vi * kVec);
        xr[it] = VecIO<T>::load_raw(x + row * d + vi * kVec);
        if (kHasRes) rr[it] = VecIO<T>::load_raw(dres + row * d + vi * kVec);
      }
    }
  };
  if (wave_global < rows) fetch(wave_global, dy_raw, x_raw, res_raw, mean, rstd);
  for (int64_t row = wave_global; row < rows; row += wave_stride) {
    Raw dy_nxt[kMaxIt], x_nxt[kMaxIt], res_nxt[kHasRes ? kMaxIt : 1];
    float mean_nxt = 0.f, rstd_nxt = 0.f;
    const bool has_next = row + wave_stride < rows;  // wave-uniform
    if (has_next) fetch(row + wave_stride, dy_nxt, x_nxt, res_nxt, mean_nxt, rstd_nxt);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int vi = lane + it * 64;
      if (vi < nvec) {
        float dyv[kVec], xv[kVec];
        VecIO<T>::unpack(dy_raw[it], dyv);
        VecIO<T>::unpack(x_raw[it], xv);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
          const float xhat = (xv[k] - mean) * rstd;
          const float gg = dyv[k] * wv[it][k];
          c1 += gg * xhat;
          if (!kRms) c2 += gg;
          dw_acc[it][k] += dyv[k] * xhat;
          if constexpr (!kRms) db_acc[it][k] += dyv[k];
        }
      }
    }
    c1 = wave_sum(c1) * inv_d;
    if (!kRms) c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int vi = lane + it * 64;
      if (vi < nvec) {
        float dyv[kVec], xv[kVec], o[kVec];
        VecIO<T>::unpack(dy_raw[it], dyv);
        VecIO<T>::unpack(x_raw[it], xv);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
          const float xhat = (xv[k] - mean) * rstd;
          o[k] = (dyv[k] * wv[it][k] - c2 - xhat * c1) * rstd;
        }
        if (kHasRes) {
          float r[kVec];
          VecIO<T>::unpack(res_raw[it], r);
#pragma unroll
          for (int k = 0; k < kVec; ++k) o[k] += r[k];
        }
        VecIO<T>::store(dx + row * d + vi * kVec, o);
        if (kDrop) {
          const uint32_t keep = dropout_keep_bits<kVec>(drop, row, vi);
#pragma unroll
          for (int k = 0; k < kVec; ++k) o[k] = (keep >> k) & 1u ? o[k] * drop.keep_scale : 0.f;
          VecIO<T>::store(dx_branch + row * d + vi * kVec, o);
        }
      }
    }
    if (has_next) {
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        dy_raw[it] = dy_nxt[it];
        x_raw[it] = x_nxt[it];
        if (kHasRes) res_raw[it] = res_nxt[it];
      }
      mean = mean_nxt;
      rstd = rstd_nxt;
    }
  }

  // Combine the 4 waves' column sums through LDS, then one partial row per block.
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][d]
  for (int pass = 0; pass < (kRms ? 1 : 2); ++pass) {
    __syncthreads();
    if (wid > 0) {
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        const int vi = lane + it * 64;
        if (vi < nvec) {
#pragma unroll
          for (int k = 0; k < kVec; ++k)
            smem[(wid - 1) * d + vi * kVec + k] = pass == 0 ? dw_acc[it][k] : db_acc[kRms ? 0 : it][k];
        }
      }
    }
    __syncthreads();
    if (wid == 0) {
      float* outp = (pass == 0 ? dw_partial : db_partial) + int64_t(blockIdx.x) * d;
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        const int vi = lane + it * 64;
        if (vi < nvec) {
#pragma unroll
          for (int k = 0; k < kVec; ++k) {
            const int c = vi * kVec + k;
            float a = pass == 0 ? dw_acc[it][k] : db_acc[kRms ? 0 : it][k];
            a += smem[c] + smem[d + c] + smem[2 * d + c];
            outp[c] = a;
          }
        }
      }
    }
  }
}

// out[c] (+)= sum_r partial[r][c]. Block = 64 columns x 16 row groups (1024 threads): the
// partial matrix is small (<= 512 x d fp32) and latency-bound, so every thread keeps
// several independent loads in flight and the row groups are folded through LDS.
constexpr int kReduceGroups = 16;
template <typename OutT>
__global__ __launch_bounds__(64 * kReduceGroups) void reduce_partials_kernel(
    const float* __restrict__ partial0, OutT* __restrict__ out0, bool accumulate0,
    const float* __restrict__ partial1, OutT* __restrict__ out1, bool accumulate1, int partial_rows, int d) {
  // blockIdx.y picks the (partial matrix, output, accumulate) triple: weight and bias gradients
  // of one LayerNorm are folded by one launch.
  const float* __restrict__ partial = blockIdx.y == 0 ? partial0 : partial1;
  OutT* __restrict__ out = blockIdx.y == 0 ? out0 : out1;
  const bool accumulate = blockIdx.y == 0 ? accumulate0 : accumulate1;
  __shared__ float sm[kReduceGroups][64];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane;
  const int rg = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < d) {
    int r = rg;
    for (; r + 3 * kReduceGroups < partial_rows; r += 4 * kReduceGroups) {
      a0 += partial[int64_t(r) * d + c];
      a1 += partial[int64_t(r + kReduceGroups) * d + c];
      a2 += partial[int64_t(r + 2 * kReduceGroups) * d + c];
      a3 += partial[int64_t(r + 3 * kReduceGroups) * d + c];
    }
    for (; r < partial_rows; r += kReduceGroups) a0 += partial[int64_t(r) * d + c];
  }
  sm[rg][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rg == 0 && c < d) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kReduceGroups; ++i) t += sm[i][lane];
    if (accumulate) t += VecIO<OutT>::load1(out + c);
    VecIO<OutT>::store1(out + c, t);
  }
}

inline unsigned fwd_grid(int64_t rows) {
  int64_t blocks = ceil_div(rows, kWavesPerBlock);
  if (blocks > 4096) blocks = 4096;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <typename T, int kMaxIt>
void launch_fwd(bool rms, const void* x, const void* res, const void* w, const void* b, void* y,
                void* sum_out, float* mean, float* rstd, int64_t rows, int d, float eps,
                const DropoutParams& drop, hipStream_t stream) {
  dim3 grid(fwd_grid(rows)), block(kThreads);
#define PDA_FWD(RMS, RES, DROP)                                                                          \
  hipLaunchKernelGGL((norm_fwd_kernel<T, kMaxIt, RMS, RES, DROP>), grid, block, 0, stream, (const T*)x, \
                     (const T*)res, (const T*)w, (const T*)b, (T*)y, (T*)sum_out, mean, rstd,           \
                     rows, d, eps, drop)
  if (rms) {  // RMSNorm models (Llama-style) have no dropout
    if (res) PDA_FWD(true, true, false); else PDA_FWD(true, false, false);
  } else if (drop.threshold) {
    if (res) PDA_FWD(false, true, true); else PDA_FWD(false, false, true);
  } else {
    if (res) PDA_FWD(false, true, false); else PDA_FWD(false, false, false);
  }
#undef PDA_FWD
}

template <typename T, int kMaxIt>
void launch_bwd(bool rms, const void* dy, const void* x, const void* w, const float* mean,
                const float* rstd, const void* dres, void* dx, void* dx_branch, float* dwp, float* dbp,
                int64_t rows, int d, const DropoutParams& drop, hipStream_t stream) {
  dim3 grid(norm_backward_partial_rows(rows)), block(kThreads);
  const size_t lds = size_t(3) * d * sizeof(float);
#define PDA_BWD(RMS, RES, DROP)                                                                    \
  do {                                                                                             \
    if (lds > 65536)                                                                               \
      (void)hipFuncSetAttribute((const void*)norm_bwd_kernel<T, kMaxIt, RMS, RES, DROP>,           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL((norm_bwd_kernel<T, kMaxIt, RMS, RES, DROP>), grid, block, lds, stream,     \
                       (const T*)dy, (const T*)x, (const T*)w, mean, rstd, (const T*)dres, (T*)dx, \
                       (T*)dx_branch, dwp, dbp, rows, d, drop);                                    \
  } while (0)
  if (rms) {
    if (dres) PDA_BWD(true, true, false); else PDA_BWD(true, false, false);
  } else if (drop.threshold) {
    if (dres) PDA_BWD(false, true, true); else PDA_BWD(false, false, true);
  } else {
    if (dres) PDA_BWD(false, true, false); else PDA_BWD(false, false, false);
  }
#undef PDA_BWD
}

template <typename T>
int iters_for(int d) {
  const int nvec = d / VecIO<T>::kVec;
  return (nvec + 63) / 64;
}

}  // namespace

int norm_backward_partial_rows(int64_t rows) {
  int64_t blocks = ceil_div(rows, kWavesPerBlock);
  if (blocks > kMaxPartialRows) blocks = kMaxPartialRows;
  return (int)(blocks < 1 ? 1 : blocks);
}

void norm_forward(DType dtype, bool rms, const void* x, const void* residual, const void* weight,
                  const void* bias, void* y, void* sum_out, float* mean, float* rstd, int64_t rows,
                  int d, float eps, float dropout_p, uint64_t seed, hipStream_t stream) {
  if (rows <= 0) return;
  const DropoutParams drop = attn::make_dropout_params(rms ? 0.f : dropout_p, seed);
#define PDA_DISPATCH(T)                                                                              \
  do {                                                                                               \
    const int it = iters_for<T>(d);                                                                  \
    if (it <= 2) launch_fwd<T, 2>(rms, x, residual, weight, bias, y, sum_out, mean, rstd, rows, d, eps, drop, stream); \
    else if (it <= 3) launch_fwd<T, 3>(rms, x, residual, weight, bias, y, sum_out, mean, rstd, rows, d, eps, drop, stream); \
    else if (it <= 4) launch_fwd<T, 4>(rms, x, residual, weight, bias, y, sum_out, mean, rstd, rows, d, eps, drop, stream); \
    else if (it <= 8) launch_fwd<T, 8>(rms, x, residual, weight, bias, y, sum_out, mean, rstd, rows, d, eps, drop, stream); \
    else launch_fwd<T, 16>(rms, x, residual, weight, bias, y, sum_out, mean, rstd, rows, d, eps, drop, stream); \
  } while (0)
  if (dtype == kBF16) PDA_DISPATCH(bf16_raw); else PDA_DISPATCH(float);
#undef PDA_DISPATCH
}

void norm_backward(DType dtype, bool rms, const void* dy, const void* x, const void* weight,
                   const float* mean, const float* rstd, const void* dresidual, void* dx, void* dx_branch,
                   float* dw_partial, float* db_partial, int64_t rows, int d, float dropout_p, uint64_t seed,
                   hipStream_t stream) {
  if (rows <= 0) return;
  const DropoutParams drop = attn::make_dropout_params(rms ? 0.f : dropout_p, seed);
#define PDA_DISPATCH(T)                                                                              \
  do {                                                                                               \
    const int it = iters_for<T>(d);                                                                  \
    if (it <= 2) launch_bwd<T, 2>(rms, dy, x, weight, mean, rstd, dresidual, dx, dx_branch, dw_partial, db_partial, rows, d, drop, stream); \
    else if (it <= 3) launch_bwd<T, 3>(rms, dy, x, weight, mean, rstd, dresidual, dx, dx_branch, dw_partial, db_partial, rows, d, drop, stream); \
    else if (it <= 4) launch_bwd<T, 4>(rms, dy, x, weight, mean, rstd, dresidual, dx, dx_branch, dw_partial, db_partial, rows, d, drop, stream); \
    else if (it <= 8) launch_bwd<T, 8>(rms, dy, x, weight, mean, rstd, dresidual, dx, dx_branch, dw_partial, db_partial, rows, d, drop, stream); \
    else launch_bwd<T, 16>(rms, dy, x, weight, mean, rstd, dresidual, dx, dx_branch, dw_partial, db_partial, rows, d, drop, stream); \
  } while (0)
  if (dtype == kBF16) PDA_DISPATCH(bf16_raw); else PDA_DISPATCH(float);
#undef PDA_DISPATCH
}

void norm_reduce_partials(const float* dw_partial, const float* db_partial, int partial_rows, int d,
                          DType dw_dtype, void* dweight, bool accumulate_w, DType db_dtype, void* dbias,
                          bool accumulate_b, hipStream_t stream) {
  const dim3 block(64 * kReduceGroups);
  auto launch = [&](DType dt, const float* p0, void* o0, bool a0, const float* p1, void* o1, bool a1, int n) {
    const dim3 grid((d + 63) / 64, n);
    if (dt == kBF16)
      hipLaunchKernelGGL(reduce_partials_kernel<bf16_raw>, grid, block, 0, stream, p0, (bf16_raw*)o0, a0, p1,
                         (bf16_raw*)o1, a1, partial_rows, d);
    else
      hipLaunchKernelGGL(reduce_partials_kernel<float>, grid, block, 0, stream, p0, (float*)o0, a0, p1, (float*)o1,
                         a1, partial_rows, d);
  };
  const bool has_w = dw_partial && dweight, has_b = db_partial && dbias;
  if (has_w && has_b && dw_dtype == db_dtype) {
    launch(dw_dtype, dw_partial, dweight, accumulate_w, db_partial, dbias, accumulate_b, 2);
  } else {
    if (has_w) launch(dw_dtype, dw_partial, dweight, accumulate_w, nullptr, nullptr, false, 1);
    if (has_b) launch(db_dtype, db_partial, dbias, accumulate_b, nullptr, nullptr, false, 1);
  }
}

}  // namespace pda
