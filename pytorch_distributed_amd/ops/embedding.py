"""Token (+ position) embedding on the native gfx950 kernels (``csrc/kernels/embedding.hip``).

``token_position_embedding(ids, wte, wpe)`` is the reference's ``wte(ids) + wpe(arange(T))``
(assignments/assignment1/model/my_gpt2.py:166-172) as one lookup kernel; ``embedding(ids, weight)``
is the plain lookup (the Llama-style models have no position table).

Backward is the part that matters. The token-table gradient is a scatter-add with collisions;
PyTorch's ``embedding_dense_backward`` sorts and segments the indices with launches whose shape
depends on the token VALUES, which cannot be replayed from a HIP graph with other tokens. The
native backward sorts (token, position) pairs with a stable radix sort and lets exactly one wave
sum each run of equal tokens in position order: no float atomics (bit-reproducible), launch shapes
that depend on the token count only (graph-capture safe), and the result is written — or, for the
tied LM-head weight, added — straight into the owner's gradient buffer (gradient sink), so no
``[V, d]`` temporary is created under DDP / FSDP.

CPU tensors and ``PDA_FORCE_REFERENCE=1`` use ``torch.nn.functional.embedding``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .. import _ext


def _grad_target(param, sink):
    """Where a parameter's gradient should be written: ``(tensor, accumulate)`` from the owner's sink,
    else an existing contiguous ``.grad`` (accumulate), else None (fresh tensor handed to autograd).
    The embedding kernels write the parameter's own dtype only: a wider sink (fp32 accumulation of a
    bf16 table) is left alone and the owner's hook adds the returned gradient itself."""
    wide = getattr(param, "_pda_grad_sink_dtype", None)
    if wide is not None and wide != param.dtype:
        sink = None
    target = sink() if sink is not None else None
    if target is None and param.is_leaf and param.grad is not None and param.grad.is_contiguous():
        target = (param.grad, True)
    return target


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, wte, wpe):
        ext = _ext.require()
        ids = ids.contiguous()
        ctx.save_for_backward(ids)
        ctx.wte, ctx.wpe = wte, wpe  # consulted for .grad / the gradient sink only
        ctx.wte_sink = getattr(wte, "_pda_grad_sink", None)
        ctx.wpe_sink = getattr(wpe, "_pda_grad_sink", None) if wpe is not None else None
        return ext.embedding_forward(ids, wte, wpe)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        ext = _ext.require()
        wte, wpe = ctx.wte, ctx.wpe
        need_wte = ctx.needs_input_grad[1]
        need_wpe = wpe is not None and ctx.needs_input_grad[2]
        if not need_wte:  # frozen table: only the position gradient (rare); use the dense path for it
            dwpe = dout.sum(0) if need_wpe else None
            if need_wpe and dwpe.shape[0] < wpe.shape[0]:
                dwpe = F.pad(dwpe, (0, 0, 0, wpe.shape[0] - dwpe.shape[0]))
            return None, None, dwpe
        w_target = _grad_target(wte, ctx.wte_sink)
        p_target = _grad_target(wpe, ctx.wpe_sink) if need_wpe else None
        w_out, w_acc = w_target if w_target is not None else (None, False)
        p_out, p_acc = p_target if p_target is not None else (None, False)
        res = ext.embedding_backward(ids, dout.contiguous(), wte.shape[0], need_wpe,
                                     wpe.shape[0] if need_wpe else 0, w_out, w_acc, p_out, p_acc)
        dwte = res[0] if w_target is None else None
        dwpe = (res[1] if p_target is None else None) if need_wpe else None
        return None, dwte, dwpe


def token_position_embedding(ids: torch.Tensor, wte: torch.Tensor, wpe: Optional[torch.Tensor]) -> torch.Tensor:
    """``wte[ids] + wpe[arange(T)]`` for ``ids [B, T]`` (``wpe=None``: the lookup alone)."""
    if _ext.use_hip(wte) and wte.dtype in (torch.bfloat16, torch.float32) and ids.dim() == 2:
        return _EmbeddingFn.apply(ids, wte, wpe)
    out = F.embedding(ids, wte)
    if wpe is not None:
        out = out + wpe[: ids.shape[-1]]
    return out


def embedding(ids: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``weight[ids]`` (``torch.nn.functional.embedding`` semantics, no padding index)."""
    return token_position_embedding(ids, weight, None)
