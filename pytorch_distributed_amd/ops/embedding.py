"""Token-embedding lookup whose backward is safe inside a HIP graph capture.

Eager mode is PyTorch's own path (``embedding_dense_backward``: sort the indices, segment them with
a unique-by-key, reduce each segment — deterministic). Replaying that path from a captured graph
was the one place a whole-step replay was seen to fault (its segmenting kernels are sized from the
data of the capturing step), so while a capture is in progress the weight gradient is instead built
by ``index_add_`` into an fp32 buffer: plain atomic adds, nothing data-dependent in the launch.
The summation order of the atomics is not fixed, so graph replays are not bit-reproducible in
this gradient; eager runs are.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight):
        ctx.save_for_backward(ids)
        ctx.rows = weight.shape[0]
        return F.embedding(ids, weight)

    @staticmethod
    def backward(ctx, grad):
        (ids,) = ctx.saved_tensors
        d = grad.shape[-1]
        if grad.is_cuda and torch.cuda.is_current_stream_capturing():
            acc = torch.zeros(ctx.rows, d, dtype=torch.float32, device=grad.device)
            acc.index_add_(0, ids.reshape(-1), grad.reshape(-1, d).float())
            return None, acc.to(grad.dtype)
        return None, torch.ops.aten.embedding_dense_backward(grad, ids, ctx.rows, -1, False)


def embedding(ids: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``weight[ids]`` (``torch.nn.functional.embedding`` semantics, no padding index)."""
    if weight.is_cuda and weight.requires_grad and torch.is_grad_enabled():
        return _EmbeddingFn.apply(ids, weight)
    return F.embedding(ids, weight)
