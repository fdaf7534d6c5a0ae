"""GPT-2 MLP (``c_fc -> gelu_new -> c_proj``) with the activation fused into the GEMMs.

The reference runs the activation as five unfused element-wise ops between two ``nn.Linear`` calls
(assignments/assignment1/model/my_gpt2.py:94-99). Here the activation never has a kernel of its
own on the GPU (``csrc/kernels/gemm_fused.hip``):

* forward: the ``c_fc`` GEMM adds the bias and writes both the pre-activation ``u`` (kept for the
  backward) and ``gelu(u)`` (the input of ``c_proj``) from its accumulators;
* backward: the input gradient of ``c_proj`` is formed on the transposed weight and multiplied by
  ``gelu'(u)`` in the epilogue, producing ``du`` directly — the ``[tokens, 4d]`` gradient of the
  activation output is never written.

Weight and bias gradients use the native weight-gradient kernel with gradient sinks, exactly as in
``ops.linear``.

Status: OPT-IN (``PDA_FUSED_MLP=1``). Measured on MI355X at 32 K tokens, d = 1280
(``tools/gemm_fused_probe.py``, ``profiles/r2/gemm_fused_probe*.log``): with the four-wave persistent
kernel the forward pair (u and gelu(u) from one GEMM) takes 499 us against 439 us for library GEMM +
GELU kernel, the backward form (GEMM * gelu'(u)) 489 us against 508 us. The activation's arithmetic
(~77 us of the chip's vector throughput per call) runs in the epilogue while the matrix pipe waits;
the stand-alone kernels hide it under their memory traffic instead. Until the kernel overlaps the two
(wave specialisation), the default is the library GEMM with the stand-alone activation kernels; the
fused path stays available, tested, for further work on it.
"""
from __future__ import annotations

import os

import torch

from .. import _ext
from .activations import gelu_new
from .linear import input_grad_mm, linear, planned_mm, weight_grads
from .regen import GeluRegen


class _GeluMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_fc, b_fc, w_proj, b_proj):
        ext = _ext.require()
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if _mode() == "bwd":  # library GEMM + activation kernel forward, fused epilogue in backward only
            u = planned_mm(x2, w_fc, trans_b=True, bias=b_fc)
            g = ext.gelu_forward(u)
        else:
            u, g = ext.gemm_fused(x2, w_fc, b_fc, "bias_gelu", None)
        y = torch.empty(*x.shape[:-1], w_proj.shape[0], dtype=x.dtype, device=x.device)
        planned_mm(g, w_proj, trans_b=True, bias=b_proj, out=y.view(-1, w_proj.shape[0]))
        ctx.save_for_backward(x2, u, g, w_fc, w_proj)
        ctx.x_shape = x.shape
        ctx.params = (w_fc, b_fc, w_proj, b_proj)  # consulted for .grad / gradient sinks only
        ctx.sinks = tuple(getattr(p, "_pda_grad_sink", None) for p in ctx.params)
        return y

    @staticmethod
    def backward(ctx, dy):
        ext = _ext.require()
        x2, u, g, w_fc, w_proj = ctx.saved_tensors
        _, b_fc, _, b_proj = ctx.params
        s_wfc, s_bfc, s_wproj, s_bproj = ctx.sinks
        need = ctx.needs_input_grad
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dw_proj, db_proj = weight_grads(dy2, g, w_proj, b_proj, s_wproj, s_bproj, need[3], need[4])
        # du = (dy . W_proj) * gelu'(u): contraction over W_proj's row index, so the kernel runs on the
        # transposed weight (a few microseconds against a GEMM of hundreds)
        (du,) = ext.gemm_fused(dy2, ext.transpose_2d(w_proj, 8), None, "mul_gelu_grad", u)
        dw_fc, db_fc = weight_grads(du, x2, w_fc, b_fc, s_wfc, s_bfc, need[1], need[2])
        dx = input_grad_mm(du, w_fc).view(ctx.x_shape) if need[0] else None
        return dx, dw_fc, db_fc, dw_proj, db_proj


def _mode() -> str:
    """"0" (off), "1" (both epilogues) or "bwd" (only the gelu' multiply of the backward)."""
    return os.environ.get("PDA_FUSED_MLP", "0")


def _fused_ok(x, w_fc, b_fc, w_proj, b_proj) -> bool:
    if _mode() not in ("1", "bwd") or not _ext.use_hip(x):
        return False
    if x.dtype != torch.bfloat16 or b_fc is None or b_proj is None:
        return False
    tokens = x.numel() // x.shape[-1]
    ext = _ext.require()
    d, f = w_fc.shape[1], w_fc.shape[0]
    return (w_proj.shape == (d, f) and w_fc.is_contiguous() and w_proj.is_contiguous()
            and ext.gemm_fused_supported(tokens, f, d) and ext.gemm_fused_supported(tokens, f, w_proj.shape[0])
            and d % 8 == 0)


def gelu_mlp(x: torch.Tensor, w_fc: torch.Tensor, b_fc, w_proj: torch.Tensor, b_proj, *, input_regen=None,
             regen_activation: bool = False) -> torch.Tensor:
    """``linear(gelu_new(linear(x, w_fc, b_fc)), w_proj, b_proj)``.

    ``input_regen``: recipe rebuilding ``x`` in backward (``ops.regen``), so ``c_fc`` does not keep
    it; ``regen_activation``: ``c_proj`` rebuilds ``gelu(u)`` from the kept ``u`` instead of keeping
    it. The fused-epilogue path keeps all three tensors and is skipped when either is asked for."""
    if input_regen is None and not regen_activation and _fused_ok(x, w_fc, b_fc, w_proj, b_proj):
        return _GeluMLPFn.apply(x, w_fc, b_fc, w_proj, b_proj)
    u = linear(x, w_fc, b_fc, regen=input_regen)
    return linear(gelu_new(u), w_proj, b_proj, regen=GeluRegen(u) if regen_activation else None)
