"""Llama-style decoder (RMSNorm, rotary positions, SwiGLU, no biases, untied LM head).

Not part of the reference (its only model is GPT-2); this is the 7B-class configuration the
sharded-data-parallel runtime is sized for: 6.7 B parameters = 13.5 GB of bf16 weights plus
81 GB of fp32 master / Adam state, i.e. something that only trains on one node when the
optimizer state is sharded. It reuses the same kernels as GPT-2 — the fused residual +
RMSNorm, the causal attention kernel at head_dim 128 over a packed QKV activation, SwiGLU,
the fused LM-head cross-entropy — and the same trunk layout (``transformer.h`` list of
blocks carrying the residual stream as a ``(delta, residual)`` pair), so ``fsdp_wrap_gpt`` and
the trainers apply unchanged.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import _ext, ops
from ..ops.regen import NormRegen, SwigluRegen
from ..utils.recompute import recompute
from .config import LlamaConfig, get_config


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        return ops.rms_norm(x, self.weight, self.eps)


def rope_tables(seq_len: int, head_dim: int, theta: float, device, dtype=torch.float32):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device=device, dtype=torch.float32) / head_dim))
    ang = torch.outer(torch.arange(seq_len, device=device, dtype=torch.float32), inv)  # [T, D/2]
    return ang.cos().to(dtype), ang.sin().to(dtype)


def _rotate(x, cos, sin, inverse: bool):
    """x: [B, T, n, H, D]; rotates pairs (x[..., :D/2], x[..., D/2:]) by +-angle(t)."""
    d2 = x.shape[-1] // 2
    ct = torch.promote_types(x.dtype, torch.float32)  # fp32 math for bf16 activations
    x1, x2 = x[..., :d2].to(ct), x[..., d2:].to(ct)
    c, s = cos[None, :, None, None, :].to(ct), sin[None, :, None, None, :].to(ct)
    if inverse:
        s = -s
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)


def _native_rope_ok(qkv, cos, head_dim: int) -> bool:
    vec = 8 if qkv.dtype == torch.bfloat16 else 4
    return (_ext.use_hip(qkv) and qkv.dtype in (torch.bfloat16, torch.float32) and qkv.is_contiguous()
            and cos.dtype == torch.float32 and cos.is_cuda and cos.is_contiguous()
            and head_dim % 2 == 0 and (head_dim // 2) % vec == 0)


class _RopePacked(torch.autograd.Function):
    """Rotary embedding applied to the Q and K thirds of a packed [B, T, 3, H, D] activation."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, num_heads):
        B, T, W = qkv.shape
        D = W // (3 * num_heads)
        ctx.save_for_backward(cos, sin)
        ctx.num_heads = num_heads
        ctx.native = _native_rope_ok(qkv, cos, D)
        if ctx.native:
            # one streaming HIP kernel, in place on the projection output (nothing else reads it)
            ctx.mark_dirty(qkv)
            return _ext.require().rope_apply(qkv, cos, sin, num_heads, False, True)
        v = qkv.view(B, T, 3, num_heads, D)
        out = torch.empty_like(v)
        out[:, :, :2] = _rotate(v[:, :, :2], cos[:T], sin[:T], inverse=False)
        out[:, :, 2] = v[:, :, 2]
        return out.view(B, T, W)

    @staticmethod
    def backward(ctx, dout):
        cos, sin = ctx.saved_tensors
        if ctx.native:
            return _ext.require().rope_apply(dout.contiguous(), cos, sin, ctx.num_heads, True, False), None, None, None
        B, T, W = dout.shape
        D = W // (3 * ctx.num_heads)
        g = dout.reshape(B, T, 3, ctx.num_heads, D)
        dq = torch.empty_like(g)
        dq[:, :, :2] = _rotate(g[:, :, :2], cos[:T], sin[:T], inverse=True)  # rotation is orthogonal
        dq[:, :, 2] = g[:, :, 2]
        return dq.view(B, T, W), None, None, None


class LlamaAttention(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.num_heads = config.n_head
        self.head_dim = config.head_dim
        self.wqkv = nn.Linear(config.n_embd, 3 * config.n_embd, bias=False)
        self.c_proj = nn.Linear(config.n_embd, config.n_embd, bias=False)

    def forward(self, x, cos, sin, input_regen=None):
        qkv = _RopePacked.apply(ops.linear(x, self.wqkv.weight, regen=input_regen), cos, sin, self.num_heads)
        return ops.linear(ops.causal_attention(qkv, self.num_heads), self.c_proj.weight)


class LlamaMLP(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.gate_up = nn.Linear(config.n_embd, 2 * config.intermediate_size, bias=False)
        self.down = nn.Linear(config.intermediate_size, config.n_embd, bias=False)

    def forward(self, x, input_regen=None, regen_activation: bool = False):
        gu = ops.linear(x, self.gate_up.weight, regen=input_regen)
        return ops.linear(ops.swiglu(gu), self.down.weight, regen=SwigluRegen(gu) if regen_activation else None)


class LlamaBlock(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.ln_1 = RMSNorm(config.n_embd, config.rms_norm_eps)
        self.attn = LlamaAttention(config)
        self.ln_2 = RMSNorm(config.n_embd, config.rms_norm_eps)
        self.mlp = LlamaMLP(config)
        self.rope = None  # (cos, sin) shared tables, set by the trunk

    def forward(self, hidden, residual=None, *, fused: bool = False, regen: bool = False):
        """``regen``: the RMSNorm outputs and the SwiGLU output are rebuilt in backward instead of
        kept (``activation_checkpoint="selective"``)."""
        regen = regen and torch.is_grad_enabled()
        cos, sin = self.rope()
        ln = self.ln_1
        y, h = ops.fused_add_rms_norm(hidden, residual, ln.weight, ln.eps)
        a = self.attn(y, cos, sin, input_regen=NormRegen(h, ln.weight, None, ln.eps, rms=True) if regen else None)
        ln = self.ln_2
        y, h = ops.fused_add_rms_norm(a, h, ln.weight, ln.eps)
        out = self.mlp(y, regen_activation=regen,
                       input_regen=NormRegen(h, ln.weight, None, ln.eps, rms=True) if regen else None)
        return (out, h) if fused else out + h


class LlamaModel(nn.Module):
    def __init__(self, config: LlamaConfig, activation_checkpoint: str = "selective"):
        super().__init__()
        self.config = config
        self.activation_checkpoint = activation_checkpoint
        self.wte = nn.Embedding(config.vocab_size, config.n_embd)
        self.h = nn.ModuleList([LlamaBlock(config) for _ in range(config.n_layer)])
        self.ln_f = RMSNorm(config.n_embd, config.rms_norm_eps)
        self._rope_cache = None
        for b in self.h:
            b.rope = self._rope

    def _rope(self):
        return self._rope_cache

    def forward(self, input_ids):
        _, T = input_ids.shape
        if T > self.config.n_ctx:
            raise ValueError(f"sequence length {T} exceeds n_ctx {self.config.n_ctx}")
        c = self._rope_cache
        if c is None or c[0].device != input_ids.device or c[0].shape[0] < T:
            self._rope_cache = rope_tables(self.config.n_ctx, self.config.head_dim, self.config.rope_theta,
                                           input_ids.device)
        delta, residual = ops.embedding(input_ids, self.wte.weight), None
        mode = self.activation_checkpoint
        full = mode == "full" and self.training and torch.is_grad_enabled()
        for block in self.h:
            if full:
                delta, residual = recompute(block, delta, residual, fused=True)
            else:
                delta, residual = block(delta, residual, fused=True, regen=mode == "selective")
        out, _ = ops.fused_add_rms_norm(delta, residual, self.ln_f.weight, self.ln_f.eps)
        return out


class LlamaForCausalLM(nn.Module):
    def __init__(self, config, activation_checkpoint: str = "selective"):
        super().__init__()
        if isinstance(config, str):
            config = get_config(config)
        self.config = config
        self.transformer = LlamaModel(config, activation_checkpoint)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.transformer.wte.weight
        self.init_weights()

    @torch.no_grad()
    def init_weights(self):
        std = 0.02
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.normal_(0.0, std)
            elif isinstance(m, nn.Embedding):
                m.weight.normal_(0.0, std)
            elif isinstance(m, RMSNorm):
                m.weight.fill_(1.0)
        for b in self.transformer.h:  # residual projections scaled by depth
            b.attn.c_proj.weight.normal_(0.0, std / math.sqrt(2 * self.config.n_layer))
            b.mlp.down.weight.normal_(0.0, std / math.sqrt(2 * self.config.n_layer))

    def forward(self, input_ids, targets: Optional[torch.Tensor] = None, ignore_index: int = -100):
        hidden = self.transformer(input_ids)
        if targets is None:
            return self.lm_head(hidden)
        return ops.lm_head_cross_entropy(hidden, self.lm_head.weight, targets, ignore_index)

    @property
    def device(self):
        return next(self.parameters()).device

    def num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def save(self, path: str) -> None:
        torch.save(self.state_dict(), path)
