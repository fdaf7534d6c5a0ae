// Fused softmax cross-entropy forward + backward for gfx950.
//
// One 512-thread workgroup owns one logits row at a time. The row (up to
// 32 x 512 sixteen-byte vectors, i.e. 131072 bf16 or 65536 fp32 columns) is
// loaded ONCE into registers, reduced to max and sum-of-exponentials with two
// workgroup reductions, and written back in place as the gradient
// (softmax - onehot) * scale. HBM traffic is therefore one read and one write
// of the logits, against the four-plus passes of log_softmax + nll_loss +
// their backward (the reference's loss:
// assignments/assignment1/train/trainer.py:53-56).
//
// The row stride `ld` may exceed the vocabulary size: the LM head pads the
// logits row to a multiple of 128 columns so that every row starts 16-byte
// aligned even for V = 50257; padding columns never enter the softmax and
// receive a zero gradient.
#include "common.h"
#include "pda_kernels.h"

namespace pda {
namespace {

constexpr int kThreads = 512;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// kMaxIt > 0: row cached in registers. kMaxIt == 0: row re-read from memory
// (rows longer than the register budget). kScalar: unaligned rows, one
// element per access (always re-reads).
template <typename T, int kMaxIt, bool kScalar>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(kMaxIt <= 16 ? 4 : 2))) void ce_kernel(T* __restrict__ logits,
                                                      const int64_t* __restrict__ targets,
                                                      float* __restrict__ row_loss,
                                                      const float* __restrict__ grad_scale,
                                                      int64_t rows, int vocab, int64_t ld,
                                                      int64_t ignore_index) {
  constexpr int kVec = kScalar ? 1 : VecIO<T>::kVec;
  __shared__ float scratch[16];
  const int tid = threadIdx.x;
  const int nvec = (int)(ld / kVec);  // vectors per row incl. padding columns
  const float scale = *grad_scale;

  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    T* rp = logits + row * ld;
    const int64_t tgt = targets[row];
    // Out-of-range targets are treated like ignore_index rather than read out of bounds.
    if (tgt == ignore_index || tgt < 0 || tgt >= vocab) {
      for (int vi = tid; vi < nvec; vi += kThreads) {
        if constexpr (kScalar) {
          VecIO<T>::store1(rp + vi, 0.f);
        } else {
          float z[kVec];
#pragma unroll
          for (int k = 0; k < kVec; ++k) z[k] = 0.f;
          VecIO<T>::store(rp + vi * kVec, z);
        }
      }
      if (tid == 0) row_loss[row] = 0.f;
      continue;
    }
    const float x_tgt = VecIO<T>::load1(rp + tgt);  // read before the row is overwritten

    // ---- pass 1: load, mask the padding columns, row maximum -----------------
    // Columns >= vocab are replaced by -inf as they are loaded, so the later passes need
    // no per-element column test: exp2(-inf) = 0 keeps them out of the sum and gives
    // them a zero gradient.
    typename VecIO<T>::Raw cache[kMaxIt > 0 ? kMaxIt : 1];
    float m = -INFINITY;
    if constexpr (kMaxIt > 0) {
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        const int vi = tid + it * kThreads;
        if (vi < nvec) {
          cache[it] = VecIO<T>::load_raw(rp + vi * kVec);
          float v[kVec];
          VecIO<T>::unpack(cache[it], v);
          if (vi * kVec + kVec > vocab) {  // only the vector(s) at the vocabulary edge
#pragma unroll
            for (int k = 0; k < kVec; ++k)
              if (vi * kVec + k >= vocab) v[k] = -INFINITY;
            cache[it] = VecIO<T>::pack(v);
          }
#pragma unroll
          for (int k = 0; k < kVec; ++k) m = fmaxf(m, v[k]);
        }
      }
    } else {
      for (int vi = tid; vi < nvec; vi += kThreads) {
        if constexpr (kScalar) {
          if (vi < vocab) m = fmaxf(m, VecIO<T>::load1(rp + vi));
        } else {
          float v[kVec];
          VecIO<T>::load(rp + vi * kVec, v);
#pragma unroll
          for (int k = 0; k < kVec; ++k)
            if (vi * kVec + k < vocab) m = fmaxf(m, v[k]);
        }
      }
    }
    m = block_max<kThreads>(m, scratch);
    if constexpr (kMaxIt > 0) {
      // Keep the row packed between passes (the unpacked fp32 copy would double its footprint).
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) asm volatile("" : "+v"(cache[it]));
    }
    const float m2 = m * kLog2e;

    // ---- pass 2: sum of exponentials ----------------------------------------
    float s = 0.f;
    if constexpr (kMaxIt > 0) {
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        if (tid + it * kThreads < nvec) {
          float v[kVec];
          VecIO<T>::unpack(cache[it], v);
#pragma unroll
          for (int k = 0; k < kVec; ++k) s += exp2f(v[k] * kLog2e - m2);
        }
        // One cached vector at a time: without this fence the scheduler interleaves all
        // kMaxIt iterations and holds kMaxIt * kVec fp32 temporaries.
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (int vi = tid; vi < nvec; vi += kThreads) {
        if constexpr (kScalar) {
          if (vi < vocab) s += exp2f(VecIO<T>::load1(rp + vi) * kLog2e - m2);
        } else {
          float v[kVec];
          VecIO<T>::load(rp + vi * kVec, v);
#pragma unroll
          for (int k = 0; k < kVec; ++k)
            if (vi * kVec + k < vocab) s += exp2f(v[k] * kLog2e - m2);
        }
      }
    }
    s = block_sum<kThreads>(s, scratch);
    if constexpr (kMaxIt > 0) {
      // Make the cached row opaque here: otherwise the compiler keeps every
      // exponential of pass 2 alive for pass 3 (8 fp32 per cached vector) and spills.
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) asm volatile("" : "+v"(cache[it]));
    }
    const float lse = m + log2f(s) * kLn2;
    if (tid == 0) row_loss[row] = lse - x_tgt;
    const float inv_s_scaled = scale / s;

    // ---- pass 3: gradient, in place -------------------------------------------
    // Every column gets softmax * scale; the target column is then corrected by the
    // thread that stored it (same-thread stores to one address stay in program order).
    if constexpr (kMaxIt > 0) {
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        const int vi = tid + it * kThreads;
        if (vi < nvec) {
          float v[kVec], g[kVec];
          VecIO<T>::unpack(cache[it], v);
#pragma unroll
          for (int k = 0; k < kVec; ++k) g[k] = exp2f(v[k] * kLog2e - m2) * inv_s_scaled;
          VecIO<T>::store(rp + vi * kVec, g);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (int vi = tid; vi < nvec; vi += kThreads) {
        if constexpr (kScalar) {
          const float p = exp2f(VecIO<T>::load1(rp + vi) * kLog2e - m2) * inv_s_scaled;
          VecIO<T>::store1(rp + vi, vi < vocab ? p : 0.f);
        } else {
          float v[kVec], g[kVec];
          VecIO<T>::load(rp + vi * kVec, v);
#pragma unroll
          for (int k = 0; k < kVec; ++k)
            g[k] = vi * kVec + k < vocab ? exp2f(v[k] * kLog2e - m2) * inv_s_scaled : 0.f;
          VecIO<T>::store(rp + vi * kVec, g);
        }
      }
    }
    if (tid == (int)((tgt / kVec) % kThreads))
      VecIO<T>::store1(rp + tgt, exp2f(x_tgt * kLog2e - m2) * inv_s_scaled - scale);
  }
}

template <typename T>
void launch(void* logits, const int64_t* targets, float* row_loss, const float* grad_scale,
            int64_t rows, int vocab, int64_t ld, int64_t ignore_index, hipStream_t stream) {
  int64_t blocks = rows < 2048 ? rows : 2048;
  dim3 grid((unsigned)blocks), block(kThreads);
  constexpr int kVec = VecIO<T>::kVec;
  const bool aligned = (ld % kVec == 0) && (reinterpret_cast<uintptr_t>(logits) % 16 == 0);
#define PDA_CE(IT, SC)                                                                         \
  hipLaunchKernelGGL((ce_kernel<T, IT, SC>), grid, block, 0, stream, (T*)logits, targets, row_loss, \
                     grad_scale, rows, vocab, ld, ignore_index)
  if (!aligned) {
    PDA_CE(0, true);
    return;
  }
  const int64_t its = ceil_div(ld / kVec, kThreads);
  if (its <= 4) PDA_CE(4, false);
  else if (its <= 8) PDA_CE(8, false);
  else if (its <= 16) PDA_CE(16, false);
  else if (its <= 32) PDA_CE(32, false);
  else PDA_CE(0, false);
#undef PDA_CE
}

}  // namespace

void cross_entropy_fwd_bwd(DType dtype, void* logits, const int64_t* targets, float* row_loss,
                           const float* grad_scale, int64_t rows, int vocab, int64_t ld,
                           int64_t ignore_index, hipStream_t stream) {
  if (rows <= 0) return;
  if (dtype == kBF16)
    launch<bf16_raw>(logits, targets, row_loss, grad_scale, rows, vocab, ld, ignore_index, stream);
  else
    launch<float>(logits, targets, row_loss, grad_scale, rows, vocab, ld, ignore_index, stream);
}

}  // namespace pda
