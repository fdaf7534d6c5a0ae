"""Activations that are rebuilt in backward instead of being kept.

A linear layer keeps its input for the weight gradient. When that input is the output of a
cheap element-wise kernel whose own input is kept anyway — ``LayerNorm(h)`` (``h`` is the
residual stream, saved by the LayerNorm backward) or ``gelu(u)`` (``u`` is the ``c_fc`` GEMM
output, saved by the GELU backward) — the linear layer can hold a *recipe* instead: the tensors
the producer already keeps plus the kernel to run on them. The activation is then written twice
(forward, and again just before the weight-gradient GEMM) and never stored between the passes.

This is the reference's selective activation checkpointing — "save the matmul outputs, recompute
everything else" (assignments/assignment1/model/pytorch_utils.py:5-17 and
my_gpt2.py:176-196, built on ``torch.utils.checkpoint`` with a per-op policy) — expressed as
data carried by the consuming op rather than as a re-execution of the block under a policy: no
second pass over the Python forward, no saved-tensor hooks, and the GEMMs are never re-run.

Per GPT-2 block and token, what stays resident in bf16 drops from ``16 d`` elements (residual
stream x2, QKV 3d, attention output, ``u`` 4d, and LayerNorm outputs x2, ``gelu(u)`` 4d) to
``10 d``; the price is two LayerNorm and one GELU kernel per block in backward
(about 4 % of a GPT-2 large step on MI355X), which is why ``activation_checkpoint="none"``
keeps everything and is what the benchmark uses on a 288 GB device.
"""
from __future__ import annotations

import copy
from typing import Optional, Tuple

import torch

from .. import _ext


class Regen:
    """Recipe for one activation: ``tensors`` are saved by the consumer, ``rebuild`` turns them
    back into the activation."""

    tensors: Tuple[Optional[torch.Tensor], ...] = ()

    def rebuild(self, *tensors) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError

    def recipe(self) -> "Regen":
        """The same recipe without its tensors (those travel through ``save_for_backward`` so that
        autograd releases them with the graph)."""
        bare = copy.copy(self)
        bare.tensors = ()
        return bare


class NormRegen(Regen):
    """``layer_norm(h)`` / ``rms_norm(h)`` of an already formed residual stream ``h``."""

    def __init__(self, h: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float,
                 rms: bool = False):
        self.tensors = (h, weight, bias)
        self.eps = float(eps)
        self.rms = bool(rms)

    def rebuild(self, h, weight, bias):
        with torch.no_grad():
            if _ext.use_hip(h):
                return _ext.require().norm_forward(h, None, weight, bias, self.eps, self.rms)[0]
            from .norm import layer_norm_reference, rms_norm_reference
            if self.rms:
                return rms_norm_reference(h, None, weight, self.eps)[0]
            return layer_norm_reference(h, None, weight, bias, self.eps)[0]


class GeluRegen(Regen):
    """``gelu_new(u)`` of a kept pre-activation ``u``."""

    def __init__(self, u: torch.Tensor):
        self.tensors = (u,)

    def rebuild(self, u):
        with torch.no_grad():
            if _ext.use_hip(u):
                return _ext.require().gelu_forward(u.contiguous())
            from .activations import gelu_new_reference
            return gelu_new_reference(u)


class SwigluRegen(Regen):
    """``silu(gate) * up`` of a kept ``[gate | up]`` projection output."""

    def __init__(self, x: torch.Tensor):
        self.tensors = (x,)

    def rebuild(self, x):
        with torch.no_grad():
            if _ext.use_hip(x):
                return _ext.require().swiglu_forward(x.contiguous())
            from .activations import swiglu_reference
            return swiglu_reference(x)
