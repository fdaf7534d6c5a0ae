// Host-side launch API of the hand-written gfx950 kernels.
//
// The kernels know nothing about PyTorch: every entry point takes raw device
// pointers, sizes and a HIP stream, so each can be driven from the Python
// binding (bindings.cpp) or from a stand-alone C++ benchmark under rocprofv3.
// No entry point allocates, frees or synchronises, so all of them are safe
// inside a hipGraph capture.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace pda {

enum DType : int { kF32 = 0, kBF16 = 1 };

// ------------------------------------------------------------------ AdamW --
struct AdamWHyper {
  float lr;
  float beta1;
  float beta2;
  float eps;
  float weight_decay;
  float bias_correction1;  // 1 - beta1^t
  float bias_correction2;  // 1 - beta2^t
  float grad_scale;        // gradients are multiplied by this before use
  // Optional device array {1 - lr*wd, lr / bias_correction1, 1 / sqrt(bias_correction2)} that
  // overrides lr / weight_decay / bias_correction* above (HIP-graph capture: see adamw.hip).
  const float* dynamic = nullptr;
};

// One tensor of a multi-tensor AdamW step. `param` is the fp32 value that is
// updated in place (the parameter itself for fp32 models, the master copy for
// bf16 models); `lowp` (optional) receives a bf16 copy of the updated value.
struct AdamWTensor {
  float* param;
  void* lowp;        // bf16 mirror or nullptr
  void* grad;        // fp32 or bf16, see grad_dtype
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
};

// Decoupled-weight-decay Adam over `n` tensors in as few launches as possible
// (tensors of any size are packed 48 to a launch).
// If zero_grad is set the gradient is overwritten with zeros once consumed.
void adamw_step(const AdamWTensor* tensors, int n, DType grad_dtype, const AdamWHyper& hp,
                bool zero_grad, hipStream_t stream);

// ----------------------------------------------------- LayerNorm / RMSNorm --
// y = LN(x [+ residual]) * w + b over rows of length d. If `residual` is given
// the sum x + residual is also written to `sum_out` (the new residual stream).
// mean / rstd are fp32 [rows]. RMSNorm passes bias = nullptr, mean = nullptr
// and rms = true. Requires d % (16 / sizeof(T)) == 0 and 16-byte aligned rows.
void norm_forward(DType dtype, bool rms, const void* x, const void* residual, const void* weight,
                  const void* bias, void* y, void* sum_out, float* mean, float* rstd,
                  int64_t rows, int d, float eps, float dropout_p, uint64_t seed, hipStream_t stream);

// dx = dLN/dx (+ dresidual if given). Column reductions for dweight / dbias
// are produced as fp32 partial sums [norm_backward_partial_rows(rows), d] and
// folded by norm_reduce_partials into fp32 dweight / dbias (accumulate=true
// adds to the existing contents instead of overwriting).
int norm_backward_partial_rows(int64_t rows);
void norm_backward(DType dtype, bool rms, const void* dy, const void* x, const void* weight,
                   const float* mean, const float* rstd, const void* dresidual, void* dx, void* dx_branch,
                   float* dw_partial, float* db_partial, int64_t rows, int d, float dropout_p, uint64_t seed,
                   hipStream_t stream);
void norm_reduce_partials(const float* dw_partial, const float* db_partial, int partial_rows, int d,
                          DType dw_dtype, void* dweight, bool accumulate_w, DType db_dtype, void* dbias,
                          bool accumulate_b, hipStream_t stream);

// -------------------------------------------------------------- Activations --
// tanh-approximated GELU ("gelu_new"): y = 0.5 u (1 + tanh(sqrt(2/pi)(u + 0.044715 u^3))).
void gelu_forward(DType dtype, const void* u, void* y, int64_t n, hipStream_t stream);
void gelu_backward(DType dtype, const void* dy, const void* u, void* du, int64_t n, hipStream_t stream);
// SwiGLU over x = [gate | up] rows of length 2*f: y = silu(gate) * up.
void swiglu_forward(DType dtype, const void* x, void* y, int64_t rows, int f, hipStream_t stream);
void swiglu_backward(DType dtype, const void* dy, const void* x, void* dx, int64_t rows, int f,
                     hipStream_t stream);

// --------------------------------------------------- Softmax cross-entropy --
// One pass over logits [rows, ld] (only the first `vocab` columns are real):
// writes per-row loss = logsumexp - logit[target] (0 where target == ignore_index)
// and replaces the logits IN PLACE with (softmax - onehot) * (*grad_scale)
// (0 for ignored rows and for the padding columns vocab..ld-1).
void cross_entropy_fwd_bwd(DType dtype, void* logits, const int64_t* targets, float* row_loss,
                           const float* grad_scale, int64_t rows, int vocab, int64_t ld,
                           int64_t ignore_index, hipStream_t stream);

// ----------------------------------------------------------------- Attention --
// Causal self-attention over a packed QKV activation [B, T, 3, H, D] (bf16),
// D in {64, 128}. out is [B, T, H, D] bf16, lse is [B, H, T] fp32 (natural log).
// dropout_p > 0 drops attention probabilities with a counter-based (Philox) mask derived from
// `seed`; the backward must be given the same dropout_p and seed.
void attention_forward(const void* qkv, void* out, float* lse, int B, int T, int H, int D,
                       float softmax_scale, float dropout_p, uint64_t seed, hipStream_t stream);
// dqkv has the layout of qkv. delta is an fp32 [B, H, T] workspace.
void attention_backward(const void* qkv, const void* out, const void* dout, const float* lse,
                        float* delta, void* dqkv, int B, int T, int H, int D, float softmax_scale,
                        float dropout_p, uint64_t seed, hipStream_t stream);
// keep[B*H][T][T] (bytes): 1 where the dropout mask of (dropout_p, seed) keeps the probability.
void attention_dropout_mask(uint8_t* keep, int B, int T, int H, float dropout_p, uint64_t seed, hipStream_t stream);

// ------------------------------------------------------------------- Rotary --
// Rotates the Q and K thirds of a packed [B, T, 3, H, D] activation (rotate-half pairing) by
// the angles of the fp32 tables cos / sin [>= T, D/2]; `inverse` rotates back (the backward).
// src == dst works in place; otherwise the V third is copied through. D/2 must be a multiple of
// the 16-byte vector width (8 bf16 / 4 fp32).
void rope_apply(DType dtype, const void* src, void* dst, const float* cos_tab, const float* sin_tab, int B, int T,
                int H, int D, bool inverse, hipStream_t stream);

// ------------------------------------------------------------------- Transpose --
// dst[c][r] = src[r][c] for a bf16 [rows, cols] matrix (cols % 8 == 0, row pitches in elements,
// multiples of 8). Rows rows..rows_padded-1 of the source are taken as zeros, i.e. every
// destination row is written up to column rows_padded (a multiple of 8, <= ld_dst).
void transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, int rows_padded,
                    hipStream_t stream);

// ------------------------------------------------------------- Weight gradient --
// C[I, J] = A[M, I]^T . B[M, J] (bf16 operands, row strides lda / ldb in elements) with the
// contraction M split over S slices: fp32 partials [S][I][J] (+ optional column sums of A,
// [S][I], the bias gradient), folded by wgrad_reduce. Needs M % 32 == 0, I % 8 == 0, J % 8 == 0
// (wgrad_supported; output tiles are 256 x 256, partial at the edges); wgrad_pick_slices
// chooses S to fill the 256 CUs.
bool wgrad_supported(int M, int I, int J);
int wgrad_pick_slices(int M, int I, int J);
void wgrad_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, float* c_partial,
                float* bias_partial, int M, int I, int J, int S, hipStream_t stream);
// Folds the first `numel` elements of every slice (slices `slice_stride` elements apart; numel ==
// slice_stride for a whole result, smaller when only the leading rows of C are wanted).
// bias_partial (optional, [S][bias_numel]) is folded into bias_out (bf16) by the same launch.
// The same product over rows that are the concatenation of `n_segments` (<= wgrad_max_segments())
// equally long pieces at unrelated addresses: A[k] is [rows_per_segment, I], B[k] [rows_per_segment, J],
// all with the row pitches lda / ldb. One long-K GEMM for the (dY, X) pairs of several micro-batches.
int wgrad_max_segments();
void wgrad_gemm_segments(const void* const* A, int64_t lda, const void* const* B, int64_t ldb, int n_segments,
                         float* c_partial, float* bias_partial, int rows_per_segment, int I, int J, int S,
                         hipStream_t stream);
void wgrad_reduce(const float* partial, int S, int64_t numel, int64_t slice_stride, DType out_dtype, void* out, bool accumulate,
                  const float* bias_partial, int64_t bias_numel, DType bias_dtype, void* bias_out,
                  bool bias_accumulate, hipStream_t stream);

// ------------------------------------------------------------- Fused-epilogue GEMM --
// out[M, N] = epilogue(X[M, K] . W[N, K]^T), bf16 operands with row strides in elements (multiples
// of 8, 16-byte aligned bases), fp32 accumulation. K % 32 == 0, N % 8 == 0 (gemm_fused_supported).
//   kEpiNone          out = acc
//   kEpiBias          out = acc + bias[N]
//   kEpiBiasGelu      out = u = acc + bias,  aux = gelu_new(u)           (aux: second output [M, N])
//   kEpiMulGeluGrad   out = acc * gelu_new'(aux)                         (aux: input  [M, N])
enum GemmEpilogue : int { kEpiNone = 0, kEpiBias = 1, kEpiBiasGelu = 2, kEpiMulGeluGrad = 3 };
bool gemm_fused_supported(int M, int N, int K);
void gemm_fused(const void* X, int64_t ldx, const void* W, int64_t ldw, const void* bias, void* out, int64_t ldo, void* aux,
                int64_t ldaux, int M, int N, int K, int epilogue, hipStream_t stream);

// ----------------------------------------------------------------- Embedding --
// out[n, :] = wte[ids[n], :] (+ wpe[n mod T, :] when wpe != nullptr). d % (16 / sizeof(T)) == 0.
void embedding_forward(DType dtype, const int64_t* ids, const void* wte, const void* wpe, void* out, int64_t n_tokens,
                       int T_len, int d, int vocab, hipStream_t stream);
// dwte[v, :] (+)= sum of dout rows whose token is v: deterministic (stable sort by token, one
// writer per row), launch shapes depend on n_tokens only. Rows of dwte whose token does not occur
// are left untouched (zero them first unless accumulating). `workspace` needs
// embedding_backward_workspace_bytes(n_tokens, vocab) bytes.
size_t embedding_backward_workspace_bytes(int n_tokens, int vocab);
void embedding_backward_wte(DType dtype, const int64_t* ids, const void* dout, void* dwte, bool accumulate, int n_tokens,
                            int d, int vocab, void* workspace, size_t workspace_bytes, hipStream_t stream);
// dwpe[t, :] (+)= sum over the batch of dout[b, t, :].
void embedding_backward_wpe(DType dtype, const void* dout, void* dwpe, bool accumulate, int batch, int T_len, int d,
                            hipStream_t stream);

// ----------------------------------------------------------------- Utilities --
// dst(fp32) += src(bf16) * scale, and plain casts, for main-grad accumulation.
void accumulate_bf16_into_f32(const void* src_bf16, float* dst, int64_t n, float scale, hipStream_t stream);
void cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, hipStream_t stream);

}  // namespace pda
