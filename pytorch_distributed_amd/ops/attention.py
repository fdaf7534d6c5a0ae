"""Causal self-attention over a packed QKV activation.

``causal_attention(qkv, num_heads)`` takes the ``[B, T, 3*H*D]`` output of the
fused QKV projection and returns ``[B, T, H*D]``. On the GPU this is one
flash-style forward kernel and two backward kernels (``csrc/kernels/
attention_*.hip``) that read Q, K and V in place with strides; the
``[B, H, T, T]`` score tensor the reference materialises
(assignments/assignment1/model/my_gpt2.py:60-77) never exists, and no causal
mask buffer is needed.

``attention_reference`` is the reference's eager op sequence in fp32 and is the
oracle of the kernel tests. Attention-probability dropout (``attn_pdrop > 0``
in training mode) is only implemented by the reference path.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .. import _ext


def attention_reference(qkv: torch.Tensor, num_heads: int, scale: float | None = None,
                        dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    B, T, three_hd = qkv.shape
    D = three_hd // (3 * num_heads)
    scale = 1.0 / math.sqrt(D) if scale is None else scale
    q, k, v = qkv.float().view(B, T, 3, num_heads, D).permute(2, 0, 3, 1, 4)  # each [B, H, T, D]
    scores = torch.matmul(q, k.transpose(-2, -1)) * scale
    mask = torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril()
    scores = scores.masked_fill(~mask, float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    if dropout_p > 0.0 and training:
        probs = F.dropout(probs, p=dropout_p, training=True)
    out = torch.matmul(probs, v)  # [B, H, T, D]
    return out.transpose(1, 2).reshape(B, T, num_heads * D).to(qkv.dtype)


def attention_sdpa(qkv: torch.Tensor, num_heads: int, scale: float | None = None) -> torch.Tensor:
    """Memory-light CPU path (no T x T tensor in autograd) for the multi-process gloo tests."""
    B, T, three_hd = qkv.shape
    D = three_hd // (3 * num_heads)
    q, k, v = qkv.view(B, T, 3, num_heads, D).permute(2, 0, 3, 1, 4)
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True, scale=scale)
    return out.transpose(1, 2).reshape(B, T, num_heads * D)


class _CausalAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, num_heads, scale):
        ext = _ext.require()
        qkv = qkv.contiguous()
        out, lse = ext.attention_forward(qkv, num_heads, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.num_heads, ctx.scale = num_heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        dqkv = _ext.require().attention_backward(qkv, out, dout.contiguous(), lse, ctx.num_heads, ctx.scale)
        return dqkv, None, None


def causal_attention(qkv: torch.Tensor, num_heads: int, scale: float | None = None,
                     dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    D = qkv.shape[-1] // (3 * num_heads)
    scale = 1.0 / math.sqrt(D) if scale is None else float(scale)
    want_dropout = dropout_p > 0.0 and training
    if _ext.use_hip(qkv) and not want_dropout and qkv.dtype == torch.bfloat16 and D in (64, 128):
        return _CausalAttentionFn.apply(qkv, num_heads, scale)
    if want_dropout or qkv.is_cuda:
        return attention_reference(qkv, num_heads, scale, dropout_p, training)
    return attention_sdpa(qkv, num_heads, scale)
