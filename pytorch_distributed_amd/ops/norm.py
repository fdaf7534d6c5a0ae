"""LayerNorm / RMSNorm with the residual add fused in.

``fused_add_layer_norm(x, residual, weight, bias, eps)`` returns
``(LN(x + residual), x + residual)``: the normalised activation for the next
GEMM and the updated residual stream, produced by one kernel that reads each
row once. In the backward the gradient arriving through the residual stream
is added inside the LayerNorm-backward kernel, so a pre-norm transformer block
(reference: assignments/assignment1/model/my_gpt2.py:121-134) runs without any
stand-alone elementwise add. With ``dropout_p`` the kernel also applies the
residual / embedding dropout of the reference (my_gpt2.py:55-56,97-98,172) to
``x`` on the way in — ``LN(dropout(x) + residual)`` — from a counter-based mask
that the backward kernel regenerates, so no dropout kernel and no stored mask.

GPU tensors run the gfx950 kernels of ``csrc/kernels/norm.hip``; CPU tensors
(and ``PDA_FORCE_REFERENCE=1``) run the PyTorch reference below, which is also
the oracle of the kernel parity tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .. import _ext
from .attention import _draw_seed
from .linear import _grad_target


# --------------------------------------------------------------------------- reference
def layer_norm_reference(x, residual, weight, bias, eps):
    h = x if residual is None else x + residual
    y = F.layer_norm(h.float(), (h.shape[-1],), weight.float(), bias.float(), eps).to(h.dtype)
    return y, h


def rms_norm_reference(x, residual, weight, eps):
    h = x if residual is None else x + residual
    hf = h.float()
    y = hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()
    return y.to(h.dtype), h


# --------------------------------------------------------------------------- HIP path
class _FusedNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps, rms, dropout_p=0.0, seed=0):
        ext = _ext.require()
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        y, h, mean, rstd = ext.norm_forward(x, res, weight, bias, eps, rms, dropout_p, seed)
        if res is None and dropout_p == 0.0:
            h = x
        ctx.dropout = (dropout_p, seed)
        ctx.rms = rms
        ctx.has_residual = res is not None
        # Gradient sinks (see ops.linear): the owner of a flat gradient buffer tags its
        # parameters with a callable returning (slice_of_buffer, accumulate).
        ctx.params = (weight, bias)
        ctx.sinks = (getattr(weight, "_pda_grad_sink", None), getattr(bias, "_pda_grad_sink", None))
        ctx.save_for_backward(h, weight, mean if not rms else None, rstd)
        ctx.set_materialize_grads(False)  # an unused output arrives as None, not as zeros
        return y, h

    @staticmethod
    def backward(ctx, dy, dh):
        ext = _ext.require()
        h, weight, mean, rstd = ctx.saved_tensors
        dropout_p, seed = ctx.dropout
        if dy is None and dropout_p == 0.0:  # only the residual stream was consumed downstream
            return dh, (dh if ctx.has_residual else None), None, None, None, None, None, None
        if dy is None:
            dy = torch.zeros_like(h)  # rare: let the kernel apply the mask to the residual gradient
        dy = dy.contiguous()
        dres = dh.contiguous() if dh is not None else None
        w_target = _grad_target(ctx.params[0], ctx.sinks[0]) if ctx.needs_input_grad[2] else None
        b_target = _grad_target(ctx.params[1], ctx.sinks[1]) if not ctx.rms and ctx.needs_input_grad[3] else None
        w_out, w_acc = w_target if w_target is not None else (None, False)
        b_out, b_acc = b_target if b_target is not None else (None, False)
        dx, dw, db, dx_branch = ext.norm_backward(dy, h, weight, mean, rstd, dres, ctx.rms, w_out, w_acc, b_out, b_acc,
                                                  dropout_p, seed)
        # A gradient written into its sink is reported to autograd as None (the accumulator
        # node still runs and fires the owner's hook).
        return ((dx_branch if dropout_p > 0.0 else dx), (dx if ctx.has_residual else None),
                (dw if w_target is None else None), (db if not ctx.rms and b_target is None else None),
                None, None, None, None)


def fused_add_layer_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                         bias: torch.Tensor, eps: float = 1e-5, dropout_p: float = 0.0, training: bool = False,
                         seed: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns ``(layer_norm(h), h)`` with ``h = dropout(x) + residual`` (``residual`` may be None;
    dropout only when ``training`` and ``dropout_p > 0``; on the GPU the probability is quantised
    to a multiple of 1/256 like the attention kernels')."""
    p = float(dropout_p) if training else 0.0
    if _ext.use_hip(x):
        if p > 0.0:
            return _FusedNormFn.apply(x, residual, weight, bias, eps, False, p,
                                      _draw_seed() if seed is None else int(seed))
        return _FusedNormFn.apply(x, residual, weight, bias, eps, False)
    if p > 0.0:
        x = F.dropout(x, p, True)
    return layer_norm_reference(x, residual, weight, bias, eps)


def fused_add_rms_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                       eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns ``(rms_norm(x + residual), x + residual)`` (``residual`` may be None)."""
    if _ext.use_hip(x):
        return _FusedNormFn.apply(x, residual, weight, None, eps, True)
    return rms_norm_reference(x, residual, weight, eps)


def layer_norm(x, weight, bias, eps=1e-5):
    return fused_add_layer_norm(x, None, weight, bias, eps)[0]


def rms_norm(x, weight, eps=1e-6):
    return fused_add_rms_norm(x, None, weight, eps)[0]
