// PyTorch-facing glue for the gfx950 kernels in csrc/kernels.
//
// Each function validates shapes / dtypes / alignment on the host (a kernel
// that reads out of bounds can take the whole node down, so nothing is
// launched on operands the kernels were not written for), allocates outputs
// through the caching allocator, and launches on the current HIP stream.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <optional>
#include <vector>

#include "kernels/pda_kernels.h"

namespace py = pybind11;

namespace {

pda::DType dtype_of(const at::Tensor& t) {
  if (t.scalar_type() == at::kBFloat16) return pda::kBF16;
  TORCH_CHECK(t.scalar_type() == at::kFloat, "expected a float32 or bfloat16 tensor, got ", t.scalar_type());
  return pda::kF32;
}

void check_gpu(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.defined(), name, " is undefined");
  TORCH_CHECK(t.is_cuda(), name, " must live on the GPU");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, " must be 16-byte aligned");
}

// Kernels launch on PyTorch's current stream of the current device (set by the DeviceGuard
// each binding takes on its first tensor argument).
hipStream_t current_stream() { return c10::hip::getCurrentHIPStream().stream(); }

void check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  TORCH_CHECK(err == hipSuccess, what, " launch failed: ", hipGetErrorString(err));
}

// ------------------------------------------------------------------ norms --
std::vector<at::Tensor> norm_forward(const at::Tensor& x, const std::optional<at::Tensor>& residual,
                                     const at::Tensor& weight, const std::optional<at::Tensor>& bias,
                                     double eps, bool rms, double dropout_p, uint64_t seed) {
  check_gpu(x, "x");
  TORCH_CHECK(dropout_p >= 0.0 && dropout_p < 1.0, "norm: dropout probability must be in [0, 1)");
  TORCH_CHECK(dropout_p == 0.0 || !rms, "norm: fused dropout is implemented for LayerNorm only");
  check_gpu(weight, "weight");
  const c10::DeviceGuard guard(x.device());
  const int64_t d = x.size(-1);
  const int64_t rows = x.numel() / d;
  const auto dt = dtype_of(x);
  const int vec = dt == pda::kBF16 ? 8 : 4;
  TORCH_CHECK(d % vec == 0 && d <= 64 * 16 * vec, "norm: unsupported row length ", d);
  TORCH_CHECK(weight.numel() == d && weight.scalar_type() == x.scalar_type(), "norm: bad weight");
  const void* bias_ptr = nullptr;
  if (!rms) {
    TORCH_CHECK(bias.has_value(), "layernorm needs a bias");
    check_gpu(*bias, "bias");
    TORCH_CHECK(bias->numel() == d && bias->scalar_type() == x.scalar_type(), "norm: bad bias");
    bias_ptr = bias->data_ptr();
  }
  const void* res_ptr = nullptr;
  at::Tensor sum_out;
  if (residual.has_value()) {
    check_gpu(*residual, "residual");
    TORCH_CHECK(residual->sizes() == x.sizes() && residual->scalar_type() == x.scalar_type(),
                "norm: residual must match x");
    res_ptr = residual->data_ptr();
    sum_out = at::empty_like(x);
  } else if (dropout_p > 0.0) {
    sum_out = at::empty_like(x);  // the normalised row is dropout(x), not x: it is emitted too
  }
  at::Tensor y = at::empty_like(x);
  auto stat_opts = x.options().dtype(at::kFloat);
  at::Tensor rstd = at::empty({rows}, stat_opts);
  at::Tensor mean = rms ? at::Tensor() : at::empty({rows}, stat_opts);
  pda::norm_forward(dt, rms, x.data_ptr(), res_ptr, weight.data_ptr(), bias_ptr, y.data_ptr(),
                    sum_out.defined() ? sum_out.data_ptr() : nullptr,
                    rms ? nullptr : mean.data_ptr<float>(), rstd.data_ptr<float>(), rows, (int)d,
                    (float)eps, (float)dropout_p, seed, current_stream());
  check_launch("norm_forward");
  return {y, sum_out, mean, rstd};
}

std::vector<at::Tensor> norm_backward(const at::Tensor& dy, const at::Tensor& x,
                                      const at::Tensor& weight, const std::optional<at::Tensor>& mean,
                                      const at::Tensor& rstd, const std::optional<at::Tensor>& dresidual,
                                      bool rms, const std::optional<at::Tensor>& dweight_out, bool accumulate_w,
                                      const std::optional<at::Tensor>& dbias_out, bool accumulate_b,
                                      double dropout_p, uint64_t seed) {
  TORCH_CHECK(dropout_p >= 0.0 && dropout_p < 1.0, "norm: dropout probability must be in [0, 1)");
  TORCH_CHECK(dropout_p == 0.0 || !rms, "norm: fused dropout is implemented for LayerNorm only");
  check_gpu(dy, "dy");
  check_gpu(x, "x");
  check_gpu(weight, "weight");
  check_gpu(rstd, "rstd");
  const c10::DeviceGuard guard(x.device());
  const int64_t d = x.size(-1);
  const int64_t rows = x.numel() / d;
  const auto dt = dtype_of(x);
  const int vec = dt == pda::kBF16 ? 8 : 4;
  TORCH_CHECK(d % vec == 0 && d <= 64 * 16 * vec, "norm: unsupported row length ", d);
  TORCH_CHECK(dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type(), "norm: dy must match x");
  TORCH_CHECK(weight.numel() == d && weight.scalar_type() == x.scalar_type(), "norm: bad weight");
  TORCH_CHECK(rstd.numel() == rows && rstd.scalar_type() == at::kFloat, "norm: bad rstd");
  const float* mean_ptr = nullptr;
  if (!rms) {
    TORCH_CHECK(mean.has_value() && mean->defined(), "layernorm backward needs mean");
    check_gpu(*mean, "mean");
    TORCH_CHECK(mean->numel() == rows && mean->scalar_type() == at::kFloat, "norm: bad mean");
    mean_ptr = mean->data_ptr<float>();
  }
  const void* dres_ptr = nullptr;
  if (dresidual.has_value() && dresidual->defined()) {
    check_gpu(*dresidual, "dresidual");
    TORCH_CHECK(dresidual->sizes() == x.sizes() && dresidual->scalar_type() == x.scalar_type(),
                "norm: dresidual must match x");
    dres_ptr = dresidual->data_ptr();
  }
  at::Tensor dx = at::empty_like(x);
  // With dropout the branch input gets the masked gradient, the residual stream the full one.
  at::Tensor dx_branch = dropout_p > 0.0 ? at::empty_like(x) : at::Tensor();
  const int prow = pda::norm_backward_partial_rows(rows);
  auto f32 = x.options().dtype(at::kFloat);
  at::Tensor dwp = at::empty({prow, d}, f32);
  at::Tensor dbp = rms ? at::Tensor() : at::empty({prow, d}, f32);
  hipStream_t stream = current_stream();
  pda::norm_backward(dt, rms, dy.data_ptr(), x.data_ptr(), weight.data_ptr(), mean_ptr,
                     rstd.data_ptr<float>(), dres_ptr, dx.data_ptr(),
                     dx_branch.defined() ? dx_branch.data_ptr() : nullptr, dwp.data_ptr<float>(),
                     rms ? nullptr : dbp.data_ptr<float>(), rows, (int)d, (float)dropout_p, seed, stream);
  check_launch("norm_backward");
  // Parameter gradients go into fresh tensors, or (written / accumulated) into the tensors the
  // caller supplies: slices of a reducer's flat gradient buffer, or existing .grad tensors.
  auto sink = [&](const std::optional<at::Tensor>& out, bool accumulate, const char* what) {
    if (!out.has_value() || !out->defined()) {
      TORCH_CHECK(!accumulate, "norm: accumulate needs ", what);
      return at::empty_like(weight);
    }
    TORCH_CHECK(out->is_cuda() && out->is_contiguous() && out->numel() == d &&
                    (out->scalar_type() == at::kBFloat16 || out->scalar_type() == at::kFloat),
                "norm: ", what, " must be a contiguous bf16/fp32 GPU tensor with ", d, " elements");
    return *out;
  };
  at::Tensor dw = sink(dweight_out, accumulate_w, "dweight_out");
  at::Tensor db = rms ? at::Tensor() : sink(dbias_out, accumulate_b, "dbias_out");
  pda::norm_reduce_partials(dwp.data_ptr<float>(), rms ? nullptr : dbp.data_ptr<float>(), prow, (int)d,
                            dtype_of(dw), dw.data_ptr(), accumulate_w, rms ? pda::kF32 : dtype_of(db),
                            rms ? nullptr : db.data_ptr(), accumulate_b, stream);
  check_launch("norm_reduce_partials");
  return {dx, dw, db, dx_branch};
}

// ------------------------------------------------------------ activations --
at::Tensor gelu_forward(const at::Tensor& u) {
  check_gpu(u, "u");
  const c10::DeviceGuard guard(u.device());
  at::Tensor y = at::empty_like(u);
  pda::gelu_forward(dtype_of(u), u.data_ptr(), y.data_ptr(), u.numel(), current_stream());
  check_launch("gelu_forward");
  return y;
}

at::Tensor gelu_backward(const at::Tensor& dy, const at::Tensor& u) {
  check_gpu(dy, "dy");
  check_gpu(u, "u");
  TORCH_CHECK(dy.sizes() == u.sizes() && dy.scalar_type() == u.scalar_type(), "gelu: dy must match u");
  const c10::DeviceGuard guard(u.device());
  at::Tensor du = at::empty_like(u);
  pda::gelu_backward(dtype_of(u), dy.data_ptr(), u.data_ptr(), du.data_ptr(), u.numel(), current_stream());
  check_launch("gelu_backward");
  return du;
}

at::Tensor swiglu_forward(const at::Tensor& x) {
  check_gpu(x, "x");
  const c10::DeviceGuard guard(x.device());
  const int64_t f2 = x.size(-1);
  const auto dt = dtype_of(x);
  const int vec = dt == pda::kBF16 ? 8 : 4;
  TORCH_CHECK(f2 % 2 == 0 && (f2 / 2) % vec == 0, "swiglu: last dim must be 2*f with f % ", vec, " == 0");
  auto sizes = x.sizes().vec();
  sizes.back() = f2 / 2;
  at::Tensor y = at::empty(sizes, x.options());
  pda::swiglu_forward(dt, x.data_ptr(), y.data_ptr(), x.numel() / f2, (int)(f2 / 2), current_stream());
  check_launch("swiglu_forward");
  return y;
}

at::Tensor swiglu_backward(const at::Tensor& dy, const at::Tensor& x) {
  check_gpu(dy, "dy");
  check_gpu(x, "x");
  const c10::DeviceGuard guard(x.device());
  const int64_t f2 = x.size(-1);
  const auto dt = dtype_of(x);
  const int vec = dt == pda::kBF16 ? 8 : 4;
  TORCH_CHECK(f2 % 2 == 0 && (f2 / 2) % vec == 0, "swiglu: bad last dim");
  TORCH_CHECK(dy.scalar_type() == x.scalar_type() && dy.numel() * 2 == x.numel(), "swiglu: dy mismatch");
  at::Tensor dx = at::empty_like(x);
  pda::swiglu_backward(dt, dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel() / f2, (int)(f2 / 2),
                       current_stream());
  check_launch("swiglu_backward");
  return dx;
}

// ---------------------------------------------------------- cross entropy --
// logits: [rows, ld] view (stride(0) = ld >= vocab, stride(1) = 1); overwritten with the gradient.
at::Tensor cross_entropy_fwd_bwd(at::Tensor logits, const at::Tensor& targets, int64_t vocab,
                                 const at::Tensor& grad_scale, int64_t ignore_index) {
  TORCH_CHECK(logits.is_cuda() && logits.dim() == 2 && logits.stride(1) == 1, "ce: bad logits layout");
  TORCH_CHECK(targets.is_cuda() && targets.scalar_type() == at::kLong && targets.is_contiguous(),
              "ce: targets must be contiguous int64 on the GPU");
  TORCH_CHECK(grad_scale.is_cuda() && grad_scale.scalar_type() == at::kFloat && grad_scale.numel() == 1,
              "ce: grad_scale must be a 1-element float32 GPU tensor");
  const int64_t rows = logits.size(0);
  const int64_t ld = logits.stride(0);
  TORCH_CHECK(targets.numel() == rows, "ce: one target per row");
  TORCH_CHECK(vocab > 0 && vocab <= ld && vocab <= logits.size(1) + (ld - logits.size(1)),
              "ce: vocab exceeds the row");
  // The kernel rewrites the whole padded row [0, ld); the storage must own it.
  const int64_t elt = logits.element_size();
  const int64_t end_byte = (logits.storage_offset() + (rows - 1) * ld + ld) * elt;
  TORCH_CHECK(rows == 0 || end_byte <= (int64_t)logits.storage().nbytes(),
              "ce: padded rows run past the end of the logits storage");
  const c10::DeviceGuard guard(logits.device());
  at::Tensor row_loss = at::empty({rows}, logits.options().dtype(at::kFloat));
  pda::cross_entropy_fwd_bwd(dtype_of(logits), logits.data_ptr(), targets.data_ptr<int64_t>(),
                             row_loss.data_ptr<float>(), grad_scale.data_ptr<float>(), rows, (int)vocab,
                             ld, ignore_index, current_stream());
  check_launch("cross_entropy_fwd_bwd");
  return row_loss;
}

// ------------------------------------------------------------------ AdamW --
void adamw_step(const std::vector<at::Tensor>& params, const std::vector<std::optional<at::Tensor>>& lowp,
                const std::vector<at::Tensor>& grads, const std::vector<at::Tensor>& exp_avg,
                const std::vector<at::Tensor>& exp_avg_sq, double lr, double beta1, double beta2,
                double eps, double weight_decay, int64_t step, double grad_scale, bool zero_grad,
                const std::optional<at::Tensor>& dynamic) {
  const size_t n = params.size();
  TORCH_CHECK(grads.size() == n && exp_avg.size() == n && exp_avg_sq.size() == n && lowp.size() == n,
              "adamw: list lengths differ");
  if (n == 0) return;
  TORCH_CHECK(step >= 1, "adamw: step counts from 1");
  const c10::DeviceGuard guard(params[0].device());
  std::vector<pda::AdamWTensor> ts;
  ts.reserve(n);
  const pda::DType gdt = dtype_of(grads[0]);
  for (size_t i = 0; i < n; ++i) {
    check_gpu(params[i], "param");
    check_gpu(grads[i], "grad");
    check_gpu(exp_avg[i], "exp_avg");
    check_gpu(exp_avg_sq[i], "exp_avg_sq");
    TORCH_CHECK(params[i].scalar_type() == at::kFloat && exp_avg[i].scalar_type() == at::kFloat &&
                    exp_avg_sq[i].scalar_type() == at::kFloat,
                "adamw: param / state must be float32");
    TORCH_CHECK(dtype_of(grads[i]) == gdt, "adamw: mixed gradient dtypes in one call");
    const int64_t numel = params[i].numel();
    TORCH_CHECK(grads[i].numel() == numel && exp_avg[i].numel() == numel && exp_avg_sq[i].numel() == numel,
                "adamw: size mismatch at tensor ", i);
    pda::AdamWTensor t;
    t.param = params[i].data_ptr<float>();
    t.lowp = nullptr;
    if (lowp[i].has_value() && lowp[i]->defined()) {
      check_gpu(*lowp[i], "lowp");
      TORCH_CHECK(lowp[i]->scalar_type() == at::kBFloat16 && lowp[i]->numel() == numel, "adamw: bad bf16 mirror");
      t.lowp = lowp[i]->data_ptr();
    }
    t.grad = grads[i].data_ptr();
    t.exp_avg = exp_avg[i].data_ptr<float>();
    t.exp_avg_sq = exp_avg_sq[i].data_ptr<float>();
    t.numel = numel;
    ts.push_back(t);
  }
  pda::AdamWHyper hp;
  hp.lr = (float)lr;
  hp.beta1 = (float)beta1;
  hp.beta2 = (float)beta2;
  hp.eps = (float)eps;
  hp.weight_decay = (float)weight_decay;
  hp.bias_correction1 = (float)(1.0 - std::pow(beta1, (double)step));
  hp.bias_correction2 = (float)(1.0 - std::pow(beta2, (double)step));
  hp.grad_scale = (float)grad_scale;
  if (dynamic.has_value()) {  // device-resident {1 - lr*wd, lr/bc1, 1/sqrt(bc2)}: HIP-graph capture
    TORCH_CHECK(dynamic->is_cuda() && dynamic->scalar_type() == at::kFloat && dynamic->numel() >= 3 &&
                    dynamic->is_contiguous(),
                "adamw: dynamic must be a contiguous float32 GPU tensor with 3 elements");
    hp.dynamic = dynamic->data_ptr<float>();
  }
  pda::adamw_step(ts.data(), (int)n, gdt, hp, zero_grad, current_stream());
  check_launch("adamw_step");
}

void accumulate_bf16_into_f32(const at::Tensor& src, at::Tensor dst, double scale) {
  check_gpu(src, "src");
  check_gpu(dst, "dst");
  TORCH_CHECK(src.scalar_type() == at::kBFloat16 && dst.scalar_type() == at::kFloat &&
                  src.numel() == dst.numel(),
              "accumulate: expected bf16 src and float32 dst of equal size");
  const c10::DeviceGuard guard(src.device());
  pda::accumulate_bf16_into_f32(src.data_ptr(), dst.data_ptr<float>(), src.numel(), (float)scale,
                                current_stream());
  check_launch("accumulate_bf16_into_f32");
}

// -------------------------------------------------------------- attention --
std::vector<at::Tensor> attention_forward(const at::Tensor& qkv, int64_t num_heads, double softmax_scale,
                                          double dropout_p, uint64_t seed) {
  check_gpu(qkv, "qkv");
  TORCH_CHECK(qkv.scalar_type() == at::kBFloat16, "attention: bf16 only");
  TORCH_CHECK(qkv.dim() == 3, "attention: qkv must be [B, T, 3*H*D]");
  const int64_t B = qkv.size(0), T = qkv.size(1), three_hd = qkv.size(2);
  TORCH_CHECK(three_hd % (3 * num_heads) == 0, "attention: bad qkv width");
  const int64_t D = three_hd / (3 * num_heads);
  TORCH_CHECK(D == 64 || D == 128, "attention: head_dim must be 64 or 128, got ", D);
  TORCH_CHECK(T >= 1 && B >= 1, "attention: empty input");
  TORCH_CHECK(dropout_p >= 0.0 && dropout_p < 1.0, "attention: dropout_p must be in [0, 1)");
  const c10::DeviceGuard guard(qkv.device());
  at::Tensor out = at::empty({B, T, num_heads * D}, qkv.options());
  at::Tensor lse = at::empty({B, num_heads, T}, qkv.options().dtype(at::kFloat));
  pda::attention_forward(qkv.data_ptr(), out.data_ptr(), lse.data_ptr<float>(), (int)B, (int)T,
                         (int)num_heads, (int)D, (float)softmax_scale, (float)dropout_p, seed, current_stream());
  check_launch("attention_forward");
  return {out, lse};
}

at::Tensor attention_backward(const at::Tensor& qkv, const at::Tensor& out, const at::Tensor& dout,
                              const at::Tensor& lse, int64_t num_heads, double softmax_scale,
                              double dropout_p, uint64_t seed) {
  check_gpu(qkv, "qkv");
  check_gpu(out, "out");
  check_gpu(dout, "dout");
  check_gpu(lse, "lse");
  TORCH_CHECK(qkv.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16 &&
                  dout.scalar_type() == at::kBFloat16 && lse.scalar_type() == at::kFloat,
              "attention backward: dtype mismatch");
  const int64_t B = qkv.size(0), T = qkv.size(1);
  const int64_t D = qkv.size(2) / (3 * num_heads);
  TORCH_CHECK(D == 64 || D == 128, "attention: head_dim must be 64 or 128");
  TORCH_CHECK(out.numel() == B * T * num_heads * D && dout.numel() == out.numel() &&
                  lse.numel() == B * num_heads * T,
              "attention backward: shape mismatch");
  const c10::DeviceGuard guard(qkv.device());
  at::Tensor dqkv = at::empty_like(qkv);
  at::Tensor delta = at::empty_like(lse);
  pda::attention_backward(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr<float>(),
                          delta.data_ptr<float>(), dqkv.data_ptr(), (int)B, (int)T, (int)num_heads,
                          (int)D, (float)softmax_scale, (float)dropout_p, seed, current_stream());
  check_launch("attention_backward");
  return dqkv;
}

// keep[B, H, T, T] (bool): the attention-dropout mask the kernels apply for (dropout_p, seed).
at::Tensor attention_dropout_mask(int64_t B, int64_t H, int64_t T, double dropout_p, uint64_t seed,
                                  const at::Device& device) {
  TORCH_CHECK(device.is_cuda(), "attention_dropout_mask: GPU device expected");
  const c10::DeviceGuard guard(device);
  at::Tensor keep = at::empty({B, H, T, T}, at::TensorOptions().dtype(at::kByte).device(device));
  pda::attention_dropout_mask(keep.data_ptr<uint8_t>(), (int)B, (int)T, (int)H, (float)dropout_p, seed,
                              current_stream());
  check_launch("attention_dropout_mask");
  return keep.to(at::kBool);
}

// ------------------------------------------------------------------- rotary --
// [rows, cols] bf16 (unit column stride, any 8-aligned row pitch) -> [cols, rows_padded], rows_padded
// = rows rounded up to pad_rows_to, with the padding columns zero.
at::Tensor transpose_2d(const at::Tensor& src, int64_t pad_rows_to) {
  TORCH_CHECK(src.defined() && src.is_cuda() && src.dim() == 2, "transpose_2d: expected a 2-D GPU tensor");
  TORCH_CHECK(src.scalar_type() == at::kBFloat16, "transpose_2d: bf16 only");
  const int64_t rows = src.size(0), cols = src.size(1);
  TORCH_CHECK(rows > 0 && cols > 0 && rows < (int64_t(1) << 30) && cols < (int64_t(1) << 30), "transpose_2d: bad shape");
  TORCH_CHECK(src.stride(1) == 1 && src.stride(0) >= cols && src.stride(0) % 8 == 0 && cols % 8 == 0,
              "transpose_2d: rows must be contiguous, 16-byte aligned and a multiple of 8 long");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(src.data_ptr()) % 16 == 0, "transpose_2d: source must be 16-byte aligned");
  TORCH_CHECK(pad_rows_to >= 8 && pad_rows_to % 8 == 0, "transpose_2d: pad_rows_to must be a multiple of 8");
  const c10::DeviceGuard guard(src.device());
  const int64_t rows_padded = (rows + pad_rows_to - 1) / pad_rows_to * pad_rows_to;
  at::Tensor out = at::empty({cols, rows_padded}, src.options());
  pda::transpose_bf16(src.data_ptr(), src.stride(0), out.data_ptr(), rows_padded, (int)rows, (int)cols,
                      (int)rows_padded, current_stream());
  check_launch("transpose_2d");
  return out;
}

at::Tensor rope_apply(const at::Tensor& qkv, const at::Tensor& cos, const at::Tensor& sin, int64_t num_heads,
                      bool inverse, bool in_place) {
  check_gpu(qkv, "qkv");
  check_gpu(cos, "cos");
  check_gpu(sin, "sin");
  TORCH_CHECK(qkv.dim() == 3, "rope: qkv must be [B, T, 3*H*D]");
  const int64_t B = qkv.size(0), T = qkv.size(1);
  TORCH_CHECK(qkv.size(2) % (3 * num_heads) == 0, "rope: bad qkv width");
  const int64_t D = qkv.size(2) / (3 * num_heads);
  const auto dt = dtype_of(qkv);
  const int vec = dt == pda::kBF16 ? 8 : 4;
  TORCH_CHECK(D % 2 == 0 && (D / 2) % vec == 0, "rope: head_dim / 2 must be a multiple of ", vec);
  TORCH_CHECK(cos.scalar_type() == at::kFloat && sin.scalar_type() == at::kFloat && cos.dim() == 2 &&
                  sin.sizes() == cos.sizes() && cos.size(0) >= T && cos.size(1) == D / 2,
              "rope: cos / sin must be float32 [>= T, D/2] tables");
  TORCH_CHECK(B * T < (int64_t(1) << 31) && T * D < (int64_t(1) << 31) && 3 * num_heads * D < (int64_t(1) << 31),
              "rope: token count and row length must fit 32-bit indices");
  const c10::DeviceGuard guard(qkv.device());
  at::Tensor out = in_place ? qkv : at::empty_like(qkv);
  pda::rope_apply(dt, qkv.data_ptr(), out.data_ptr(), cos.data_ptr<float>(), sin.data_ptr<float>(), (int)B, (int)T,
                  (int)num_heads, (int)D, inverse, current_stream());
  check_launch("rope_apply");
  return out;
}

// ---------------------------------------------------------------- weight grad --
bool wgrad_supported(int64_t M, int64_t I, int64_t J) { return pda::wgrad_supported((int)M, (int)I, (int)J); }

// Can `n` pairs of `rows` rows each go through linear_wgrad_segments?
bool wgrad_segments_supported(int64_t rows, int64_t I, int64_t J, int64_t n) {
  if (n < 1 || n > pda::wgrad_max_segments() || rows < 32 || rows % 32 != 0) return false;
  const int64_t steps = rows / 32;
  if ((steps & (steps - 1)) != 0 || rows * n >= (int64_t(1) << 31)) return false;
  return pda::wgrad_supported((int)(rows * n), (int)I, (int)J);
}

// dW[I, J] = dy[M, I]^T . x[M, J] and (optionally) db[I] = column sums of dy, both bf16.
// With `out` the weight gradient is written into that tensor (a slice of a reducer's flat
// gradient buffer) instead of a fresh one — added to its contents when `accumulate` is set.
// `out_rows` (default: all I): only the leading rows of dW are produced — for a dy whose trailing
// columns are zero padding (the LM head's logits gradient, V = 50257 in a 50304-column buffer).
std::vector<at::Tensor> linear_wgrad(const at::Tensor& dy, const at::Tensor& x, bool want_bias,
                                     const std::optional<at::Tensor>& out, bool accumulate,
                                     const std::optional<at::Tensor>& bias_out, bool bias_accumulate,
                                     int64_t out_rows) {
  TORCH_CHECK(dy.is_cuda() && x.is_cuda() && dy.dim() == 2 && x.dim() == 2, "wgrad: 2-D GPU tensors");
  TORCH_CHECK(dy.scalar_type() == at::kBFloat16 && x.scalar_type() == at::kBFloat16, "wgrad: bf16 only");
  TORCH_CHECK(dy.stride(1) == 1 && x.stride(1) == 1, "wgrad: rows must be contiguous");
  const int64_t M = dy.size(0), I = dy.size(1), J = x.size(1);
  TORCH_CHECK(x.size(0) == M, "wgrad: row counts differ");
  TORCH_CHECK(pda::wgrad_supported((int)M, (int)I, (int)J), "wgrad: unsupported shape ", M, "x", I, "x", J);
  TORCH_CHECK(dy.stride(0) % 8 == 0 && x.stride(0) % 8 == 0 &&
                  reinterpret_cast<uintptr_t>(dy.data_ptr()) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0,
              "wgrad: operands must be 16-byte aligned with row strides that are multiples of 8");
  if (out_rows < 0) out_rows = I;
  TORCH_CHECK(out_rows >= 1 && out_rows <= I && (out_rows * J) % 4 == 0, "wgrad: bad out_rows ", out_rows);
  TORCH_CHECK(out_rows == I || !want_bias, "wgrad: out_rows and the bias gradient are not combined");
  const c10::DeviceGuard guard(dy.device());
  const int S = pda::wgrad_pick_slices((int)M, (int)I, (int)J);
  auto f32 = dy.options().dtype(at::kFloat);
  at::Tensor partial = at::empty({S, I, J}, f32);
  at::Tensor bpartial = want_bias ? at::empty({S, I}, f32) : at::Tensor();
  hipStream_t stream = current_stream();
  pda::wgrad_gemm(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), partial.data_ptr<float>(),
                  want_bias ? bpartial.data_ptr<float>() : nullptr, (int)M, (int)I, (int)J, S, stream);
  check_launch("wgrad_gemm");
  at::Tensor dw;
  if (out.has_value()) {
    dw = *out;
    TORCH_CHECK(dw.is_cuda() && dw.is_contiguous() && dw.numel() == out_rows * J &&
                    (dw.scalar_type() == at::kBFloat16 || dw.scalar_type() == at::kFloat),
                "wgrad: out must be a contiguous bf16/fp32 GPU tensor with ", out_rows * J, " elements");
  } else {
    TORCH_CHECK(!accumulate, "wgrad: accumulate needs an out tensor");
    dw = at::empty({out_rows, J}, dy.options());
  }
  at::Tensor db;
  if (want_bias) {
    if (bias_out.has_value() && bias_out->defined()) {
      db = *bias_out;
      TORCH_CHECK(db.is_cuda() && db.is_contiguous() && db.numel() == I &&
                      (db.scalar_type() == at::kBFloat16 || db.scalar_type() == at::kFloat),
                  "wgrad: bias_out must be a contiguous bf16/fp32 GPU tensor with ", I, " elements");
    } else {
      TORCH_CHECK(!bias_accumulate, "wgrad: bias_accumulate needs a bias_out tensor");
      db = at::empty({I}, dy.options());
    }
  }
  pda::wgrad_reduce(partial.data_ptr<float>(), S, out_rows * J, I * J, dtype_of(dw), dw.data_ptr(), accumulate,
                    want_bias ? bpartial.data_ptr<float>() : nullptr, I, want_bias ? dtype_of(db) : pda::kBF16,
                    want_bias ? db.data_ptr() : nullptr, want_bias && bias_accumulate, stream);
  check_launch("wgrad_reduce");
  return {dw, db};
}

// The same product over the row-wise concatenation of up to wgrad_max_segments() (dy, x) pairs that
// live in separate buffers (the micro-batches of a gradient-accumulation window): ONE long-K GEMM and
// one slice reduction instead of one of each per micro-batch. Every pair has the same shape and row
// strides; rows per pair / 32 must be a power of two (the kernel finds a stage's pair with a shift).
std::vector<at::Tensor> linear_wgrad_segments(const std::vector<at::Tensor>& dys, const std::vector<at::Tensor>& xs,
                                              bool want_bias, const std::optional<at::Tensor>& out, bool accumulate,
                                              const std::optional<at::Tensor>& bias_out, bool bias_accumulate) {
  const int64_t n = (int64_t)dys.size();
  TORCH_CHECK(n >= 1 && n <= pda::wgrad_max_segments() && (int64_t)xs.size() == n, "wgrad segments: need 1..",
              pda::wgrad_max_segments(), " (dy, x) pairs, got ", n, " and ", xs.size());
  if (n == 1) return linear_wgrad(dys[0], xs[0], want_bias, out, accumulate, bias_out, bias_accumulate, -1);
  const at::Tensor& dy0 = dys[0];
  const at::Tensor& x0 = xs[0];
  TORCH_CHECK(dy0.is_cuda() && dy0.dim() == 2 && x0.dim() == 2, "wgrad segments: 2-D GPU tensors");
  const int64_t rows = dy0.size(0), I = dy0.size(1), J = x0.size(1);
  TORCH_CHECK(wgrad_segments_supported(rows, I, J, n), "wgrad segments: unsupported shape ", n, "x", rows, "x", I, "x", J);
  const void* a[4];
  const void* b[4];
  for (int64_t k = 0; k < n; ++k) {
    const at::Tensor& dy = dys[k];
    const at::Tensor& x = xs[k];
    TORCH_CHECK(dy.is_cuda() && x.is_cuda() && dy.device() == dy0.device() && x.device() == dy0.device() &&
                    dy.scalar_type() == at::kBFloat16 && x.scalar_type() == at::kBFloat16,
                "wgrad segments: bf16 tensors on one GPU");
    TORCH_CHECK(dy.dim() == 2 && x.dim() == 2 && dy.size(0) == rows && x.size(0) == rows && dy.size(1) == I &&
                    x.size(1) == J && dy.stride(1) == 1 && x.stride(1) == 1 && dy.stride(0) == dy0.stride(0) &&
                    x.stride(0) == x0.stride(0),
                "wgrad segments: pair ", k, " differs in shape or strides from pair 0");
    TORCH_CHECK(dy.stride(0) % 8 == 0 && x.stride(0) % 8 == 0 && reinterpret_cast<uintptr_t>(dy.data_ptr()) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0,
                "wgrad segments: operands must be 16-byte aligned with row strides that are multiples of 8");
    a[k] = dy.data_ptr();
    b[k] = x.data_ptr();
  }
  const c10::DeviceGuard guard(dy0.device());
  const int64_t M = rows * n;
  const int S = pda::wgrad_pick_slices((int)M, (int)I, (int)J);
  auto f32 = dy0.options().dtype(at::kFloat);
  at::Tensor partial = at::empty({S, I, J}, f32);
  at::Tensor bpartial = want_bias ? at::empty({S, I}, f32) : at::Tensor();
  hipStream_t stream = current_stream();
  pda::wgrad_gemm_segments(a, dy0.stride(0), b, x0.stride(0), (int)n, partial.data_ptr<float>(),
                           want_bias ? bpartial.data_ptr<float>() : nullptr, (int)rows, (int)I, (int)J, S, stream);
  check_launch("wgrad_gemm_segments");
  at::Tensor dw;
  if (out.has_value()) {
    dw = *out;
    TORCH_CHECK(dw.is_cuda() && dw.is_contiguous() && dw.numel() == I * J &&
                    (dw.scalar_type() == at::kBFloat16 || dw.scalar_type() == at::kFloat),
                "wgrad segments: out must be a contiguous bf16/fp32 GPU tensor with ", I * J, " elements");
  } else {
    TORCH_CHECK(!accumulate, "wgrad segments: accumulate needs an out tensor");
    dw = at::empty({I, J}, dy0.options());
  }
  at::Tensor db;
  if (want_bias) {
    if (bias_out.has_value() && bias_out->defined()) {
      db = *bias_out;
      TORCH_CHECK(db.is_cuda() && db.is_contiguous() && db.numel() == I &&
                      (db.scalar_type() == at::kBFloat16 || db.scalar_type() == at::kFloat),
                  "wgrad segments: bias_out must be a contiguous bf16/fp32 GPU tensor with ", I, " elements");
    } else {
      TORCH_CHECK(!bias_accumulate, "wgrad segments: bias_accumulate needs a bias_out tensor");
      db = at::empty({I}, dy0.options());
    }
  }
  pda::wgrad_reduce(partial.data_ptr<float>(), S, I * J, I * J, dtype_of(dw), dw.data_ptr(), accumulate,
                    want_bias ? bpartial.data_ptr<float>() : nullptr, I, want_bias ? dtype_of(db) : pda::kBF16,
                    want_bias ? db.data_ptr() : nullptr, want_bias && bias_accumulate, stream);
  check_launch("wgrad_reduce");
  return {dw, db};
}

}  // namespace

// ------------------------------------------------------- fused-epilogue GEMM --
void check_bf16_matrix(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.defined() && t.is_cuda() && t.scalar_type() == at::kBFloat16, name, " must be a bf16 GPU tensor");
  TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1 && t.stride(0) >= t.size(1) && t.stride(0) % 8 == 0, name,
              " must be a row-major matrix with unit column stride and a row stride that is a multiple of 8");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, " must be 16-byte aligned");
}

bool gemm_fused_supported(int64_t M, int64_t N, int64_t K) {
  return M < (int64_t(1) << 31) && N < (int64_t(1) << 31) && pda::gemm_fused_supported((int)M, (int)N, (int)K);
}

// epilogue: "none" | "bias" | "bias_gelu" (returns [u, gelu(u)]) | "mul_gelu_grad" (aux = u, returns [acc * gelu'(u)]).
std::vector<at::Tensor> gemm_fused(const at::Tensor& x, const at::Tensor& w, const std::optional<at::Tensor>& bias,
                                   const std::string& epilogue, const std::optional<at::Tensor>& aux) {
  check_bf16_matrix(x, "x");
  check_bf16_matrix(w, "w");
  const c10::DeviceGuard guard(x.device());
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  TORCH_CHECK(w.size(1) == K, "gemm_fused: x is [M, K], w must be [N, K]");
  TORCH_CHECK(gemm_fused_supported(M, N, K), "gemm_fused: unsupported shape M=", M, " N=", N, " K=", K,
              " (needs K % 32 == 0 and N % 8 == 0)");
  int epi = pda::kEpiNone;
  if (epilogue == "bias") epi = pda::kEpiBias;
  else if (epilogue == "bias_gelu") epi = pda::kEpiBiasGelu;
  else if (epilogue == "mul_gelu_grad") epi = pda::kEpiMulGeluGrad;
  else TORCH_CHECK(epilogue == "none", "gemm_fused: unknown epilogue '", epilogue, "'");
  const void* bias_ptr = nullptr;
  if (epi == pda::kEpiBias || epi == pda::kEpiBiasGelu) {
    TORCH_CHECK(bias.has_value() && bias->is_cuda() && bias->scalar_type() == at::kBFloat16 && bias->is_contiguous() &&
                    bias->numel() == N && reinterpret_cast<uintptr_t>(bias->data_ptr()) % 8 == 0,
                "gemm_fused: bias must be a contiguous, 8-byte aligned bf16 GPU vector of length ", N);
    bias_ptr = bias->data_ptr();
  }
  at::Tensor out = at::empty({M, N}, x.options());
  std::vector<at::Tensor> result{out};
  void* aux_ptr = nullptr;
  int64_t ldaux = N;
  if (epi == pda::kEpiBiasGelu) {
    at::Tensor act = at::empty({M, N}, x.options());
    aux_ptr = act.data_ptr();
    result.push_back(act);
  } else if (epi == pda::kEpiMulGeluGrad) {
    TORCH_CHECK(aux.has_value(), "gemm_fused: 'mul_gelu_grad' needs the pre-activation as aux");
    check_bf16_matrix(*aux, "aux");
    TORCH_CHECK(aux->size(0) == M && aux->size(1) == N, "gemm_fused: aux must be [M, N]");
    aux_ptr = aux->data_ptr();
    ldaux = aux->stride(0);
  }
  pda::gemm_fused(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), bias_ptr, out.data_ptr(), N, aux_ptr, ldaux, (int)M,
                  (int)N, (int)K, epi, current_stream());
  check_launch("gemm_fused");
  return result;
}

// -------------------------------------------------------------- embedding --
at::Tensor embedding_forward(const at::Tensor& ids, const at::Tensor& wte, const std::optional<at::Tensor>& wpe) {
  check_gpu(wte, "wte");
  TORCH_CHECK(ids.is_cuda() && ids.is_contiguous() && ids.scalar_type() == at::kLong, "embedding: ids must be a contiguous int64 GPU tensor");
  TORCH_CHECK(ids.dim() == 2 && wte.dim() == 2, "embedding: ids [B, T], wte [V, d]");
  const c10::DeviceGuard guard(wte.device());
  const auto dt = dtype_of(wte);
  const int64_t d = wte.size(1), T = ids.size(1);
  TORCH_CHECK(d % (dt == pda::kBF16 ? 8 : 4) == 0, "embedding: row length ", d, " is not a multiple of the 16-byte vector");
  TORCH_CHECK(wte.size(0) < (int64_t(1) << 31) && ids.numel() < (int64_t(1) << 31), "embedding: table or batch too large");
  const void* wpe_ptr = nullptr;
  if (wpe.has_value()) {
    check_gpu(*wpe, "wpe");
    TORCH_CHECK(wpe->dim() == 2 && wpe->size(1) == d && wpe->size(0) >= T && wpe->scalar_type() == wte.scalar_type(),
                "embedding: wpe must be [>= T, d] in the dtype of wte");
    wpe_ptr = wpe->data_ptr();
  }
  at::Tensor out = at::empty({ids.size(0), T, d}, wte.options());
  pda::embedding_forward(dt, ids.data_ptr<int64_t>(), wte.data_ptr(), wpe_ptr, out.data_ptr(), ids.numel(), (int)T, (int)d,
                         (int)wte.size(0), current_stream());
  check_launch("embedding_forward");
  return out;
}

// Gradients of embedding_forward. dwte_out / dwpe_out (optional): existing gradient buffers to
// write (accumulate = false) or add into (accumulate = true); without them fresh tensors are
// returned (dwte zero-initialised, then the rows of occurring tokens written).
std::vector<at::Tensor> embedding_backward(const at::Tensor& ids, const at::Tensor& dout, int64_t vocab, bool need_wpe,
                                           int64_t wpe_rows, const std::optional<at::Tensor>& dwte_out, bool wte_accumulate,
                                           const std::optional<at::Tensor>& dwpe_out, bool wpe_accumulate) {
  check_gpu(dout, "dout");
  TORCH_CHECK(ids.is_cuda() && ids.is_contiguous() && ids.scalar_type() == at::kLong && ids.dim() == 2,
              "embedding: ids must be a contiguous int64 [B, T] GPU tensor");
  const c10::DeviceGuard guard(dout.device());
  const auto dt = dtype_of(dout);
  const int64_t B = ids.size(0), T = ids.size(1), d = dout.size(-1);
  TORCH_CHECK(dout.numel() == B * T * d, "embedding: dout must be [B, T, d]");
  TORCH_CHECK(d % (dt == pda::kBF16 ? 8 : 4) == 0, "embedding: bad row length ", d);
  TORCH_CHECK(vocab > 0 && vocab < (int64_t(1) << 31) && B * T < (int64_t(1) << 31), "embedding: table or batch too large");
  at::Tensor dwte;
  bool acc_w = false;
  if (dwte_out.has_value()) {
    dwte = *dwte_out;
    check_gpu(dwte, "dwte_out");
    TORCH_CHECK(dwte.scalar_type() == dout.scalar_type() && dwte.numel() == vocab * d, "embedding: dwte_out must be [V, d] in dout's dtype");
    acc_w = wte_accumulate;
    if (!acc_w) dwte.zero_();
  } else {
    dwte = at::zeros({vocab, d}, dout.options());
  }
  const size_t ws_bytes = pda::embedding_backward_workspace_bytes((int)(B * T), (int)vocab);
  at::Tensor ws = at::empty({(int64_t)ws_bytes}, dout.options().dtype(at::kByte));
  pda::embedding_backward_wte(dt, ids.data_ptr<int64_t>(), dout.data_ptr(), dwte.data_ptr(), acc_w, (int)(B * T), (int)d,
                              (int)vocab, ws.data_ptr(), ws_bytes, current_stream());
  check_launch("embedding_backward_wte");
  std::vector<at::Tensor> result{dwte};
  if (need_wpe) {
    TORCH_CHECK(wpe_rows >= T, "embedding: position table shorter than the sequence");
    at::Tensor dwpe;
    bool acc_p = false;
    if (dwpe_out.has_value()) {
      dwpe = *dwpe_out;
      check_gpu(dwpe, "dwpe_out");
      TORCH_CHECK(dwpe.scalar_type() == dout.scalar_type() && dwpe.numel() == wpe_rows * d, "embedding: dwpe_out must be [n_ctx, d]");
      acc_p = wpe_accumulate;
      if (!acc_p && wpe_rows > T) dwpe.narrow(0, T, wpe_rows - T).zero_();
    } else {
      dwpe = wpe_rows > T ? at::zeros({wpe_rows, d}, dout.options()) : at::empty({wpe_rows, d}, dout.options());
    }
    pda::embedding_backward_wpe(dt, dout.data_ptr(), dwpe.data_ptr(), acc_p, (int)B, (int)T, (int)d, current_stream());
    check_launch("embedding_backward_wpe");
    result.push_back(dwpe);
  }
  return result;
}

void bind_ops(py::module_& m) {
  m.def("gemm_fused_supported", &gemm_fused_supported);
  m.def("gemm_fused", &gemm_fused, py::arg("x"), py::arg("w"), py::arg("bias") = py::none(), py::arg("epilogue") = "none",
        py::arg("aux") = py::none());
  m.def("embedding_forward", &embedding_forward, py::arg("ids"), py::arg("wte"), py::arg("wpe") = py::none());
  m.def("embedding_backward", &embedding_backward, py::arg("ids"), py::arg("dout"), py::arg("vocab"), py::arg("need_wpe"),
        py::arg("wpe_rows") = 0, py::arg("dwte_out") = py::none(), py::arg("wte_accumulate") = false,
        py::arg("dwpe_out") = py::none(), py::arg("wpe_accumulate") = false);
  m.def("rope_apply", &rope_apply, py::arg("qkv"), py::arg("cos"), py::arg("sin"), py::arg("num_heads"),
        py::arg("inverse"), py::arg("in_place"));
  m.def("transpose_2d", &transpose_2d, py::arg("src"), py::arg("pad_rows_to") = 8);
  m.def("wgrad_supported", &wgrad_supported);
  m.def("linear_wgrad", &linear_wgrad, py::arg("dy"), py::arg("x"), py::arg("want_bias"),
        py::arg("out") = py::none(), py::arg("accumulate") = false, py::arg("bias_out") = py::none(),
        py::arg("bias_accumulate") = false, py::arg("out_rows") = -1);
  m.def("wgrad_segments_supported", &wgrad_segments_supported, py::arg("rows"), py::arg("I"), py::arg("J"), py::arg("n"));
  m.def("wgrad_max_segments", []() { return pda::wgrad_max_segments(); });
  m.def("linear_wgrad_segments", &linear_wgrad_segments, py::arg("dys"), py::arg("xs"), py::arg("want_bias"),
        py::arg("out") = py::none(), py::arg("accumulate") = false, py::arg("bias_out") = py::none(),
        py::arg("bias_accumulate") = false);
  m.def("norm_forward", &norm_forward, py::arg("x"), py::arg("residual"), py::arg("weight"),
        py::arg("bias"), py::arg("eps"), py::arg("rms"), py::arg("dropout_p") = 0.0, py::arg("seed") = 0);
  m.def("norm_backward", &norm_backward, py::arg("dy"), py::arg("x"), py::arg("weight"), py::arg("mean"),
        py::arg("rstd"), py::arg("dresidual"), py::arg("rms"), py::arg("dweight_out") = py::none(),
        py::arg("accumulate_w") = false, py::arg("dbias_out") = py::none(), py::arg("accumulate_b") = false,
        py::arg("dropout_p") = 0.0, py::arg("seed") = 0);
  m.def("gelu_forward", &gelu_forward);
  m.def("gelu_backward", &gelu_backward);
  m.def("swiglu_forward", &swiglu_forward);
  m.def("swiglu_backward", &swiglu_backward);
  m.def("cross_entropy_fwd_bwd", &cross_entropy_fwd_bwd, py::arg("logits"), py::arg("targets"),
        py::arg("vocab"), py::arg("grad_scale"), py::arg("ignore_index"));
  m.def("adamw_step", &adamw_step, py::arg("params"), py::arg("lowp"), py::arg("grads"),
        py::arg("exp_avg"), py::arg("exp_avg_sq"), py::arg("lr"), py::arg("beta1"), py::arg("beta2"),
        py::arg("eps"), py::arg("weight_decay"), py::arg("step"), py::arg("grad_scale"),
        py::arg("zero_grad"), py::arg("dynamic") = py::none());
  m.def("accumulate_bf16_into_f32", &accumulate_bf16_into_f32);
  m.def("attention_forward", &attention_forward, py::arg("qkv"), py::arg("num_heads"),
        py::arg("softmax_scale"), py::arg("dropout_p") = 0.0, py::arg("seed") = 0);
  m.def("attention_backward", &attention_backward, py::arg("qkv"), py::arg("out"), py::arg("dout"),
        py::arg("lse"), py::arg("num_heads"), py::arg("softmax_scale"), py::arg("dropout_p") = 0.0,
        py::arg("seed") = 0);
  m.def("attention_dropout_mask", &attention_dropout_mask, py::arg("B"), py::arg("H"), py::arg("T"),
        py::arg("dropout_p"), py::arg("seed"), py::arg("device"));
}
