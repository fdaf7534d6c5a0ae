"""Linear layer with a native weight-gradient kernel.

Forward and input-gradient GEMMs go to hipBLASLt through ``F.linear`` / ``torch.mm`` (their
contraction dimension is contiguous in both operands and the library handles them at
~1.2 PFLOP/s). The weight gradient ``dW = dY^T . X`` contracts over the token dimension —
the row index of both operands — and produces a small output; there the library reaches
only 0.4-0.85 PFLOP/s on GPT-2 shapes, so it is done by ``csrc/kernels/gemm_wgrad.hip``
(split-K over tokens, transposed LDS operand reads), which also returns the bias gradient
as a by-product instead of a separate reduction pass over ``dY``.

Falls back to the library for shapes the kernel does not cover (non-bf16, dimensions not
multiples of 256, e.g. the V = 50257 LM head) and on the CPU.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .. import _ext


def _native_wgrad_ok(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16):
        return False
    if os.environ.get("PDA_NO_NATIVE_WGRAD") == "1":
        return False
    return _ext.require().wgrad_supported(dy2.shape[0], dy2.shape[1], x2.shape[1])


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, weight).view(x.shape)
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if need_w and _native_wgrad_ok(dy2, x2):
            if not dy2.is_contiguous():
                dy2 = dy2.contiguous()
            dw, db_native = _ext.require().linear_wgrad(dy2, x2.contiguous(), bool(need_b))
            if need_b:
                db = db_native
        else:
            if need_w:
                dw = torch.mm(dy2.t(), x2)
            if need_b:
                db = dy2.sum(0)
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """``x @ weight.T + bias`` (``torch.nn.functional.linear`` semantics)."""
    if _ext.use_hip(x) and x.dtype == torch.bfloat16:
        return _LinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)
