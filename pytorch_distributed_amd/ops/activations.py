"""GELU (tanh approximation, HF "gelu_new") and SwiGLU.

The backward kernels recompute the activation from the saved pre-activation,
so only the GEMM output that feeds the activation is kept for the backward —
the same save-matmul-outputs / recompute-the-rest policy as the reference's
selective activation checkpointing
(assignments/assignment1/model/pytorch_utils.py:5-17), realised inside the op.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .. import _ext


def gelu_new_reference(u: torch.Tensor) -> torch.Tensor:
    uf = u.float()
    y = 0.5 * uf * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (uf + 0.044715 * uf.pow(3))))
    return y.to(u.dtype)


def swiglu_reference(x: torch.Tensor) -> torch.Tensor:
    gate, up = x.float().chunk(2, dim=-1)
    return (F.silu(gate) * up).to(x.dtype)


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        u = u.contiguous()
        ctx.save_for_backward(u)
        return _ext.require().gelu_forward(u)

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        return _ext.require().gelu_backward(dy.contiguous(), u)


class _SwigluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return _ext.require().swiglu_forward(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _ext.require().swiglu_backward(dy.contiguous(), x)


def gelu_new(u: torch.Tensor) -> torch.Tensor:
    """tanh-approximated GELU, elementwise."""
    if _ext.use_hip(u):
        return _GeluFn.apply(u)
    return gelu_new_reference(u)


def swiglu(x: torch.Tensor) -> torch.Tensor:
    """``silu(x[..., :f]) * x[..., f:]`` for ``x`` of width ``2f``."""
    if _ext.use_hip(x):
        return _SwigluFn.apply(x)
    return swiglu_reference(x)
