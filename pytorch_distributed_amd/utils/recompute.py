"""Block recomputation: run a function without keeping its activations, re-run it in backward.

``recompute(fn, *args, **kwargs)`` returns what ``fn(*args, **kwargs)`` returns (a tensor or a
tuple of tensors) but keeps only the tensor arguments. When gradients arrive, the function is
run a second time with autograd recording, under the random-number state of the first run, and
differentiated at once; the activations of one block live only for the duration of that block's
backward. This is ``activation_checkpoint="full"`` of the model classes — the reference reaches
the same effect with ``torch.utils.checkpoint`` around each block (assignments/assignment1/
model/my_gpt2.py:176-196 with ``pytorch_utils.py:5-17`` selecting what to keep); here it is a
self-contained autograd function of a few dozen lines that knows about this framework's pieces:

* dropout masks of the fused kernels are derived from seeds drawn from the CPU generator, so
  restoring the CPU (and, when a device is in use, the device) generator state reproduces them;
* the nested backward is bracketed by ``_C.recompute_enter / recompute_exit`` so that the
  gradient reducer and the sharding runtime attach their end-of-backward callbacks to the
  model's backward, not to the block's (csrc/runtime/recompute.h);
* sharded units see the second forward through their ordinary pre-forward hook: inside a
  backward it gathers without prefetching forward and does not reshard afterwards.
"""
from __future__ import annotations

from typing import Any, Callable, Tuple

import torch

from .. import _ext
from .profiler import annotate


def _bracket():
    ext = _ext.load()
    if ext is None:
        return (lambda: None), (lambda: None)
    return ext.recompute_enter, ext.recompute_exit


class _RecomputeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fn: Callable, kwargs: dict, anchor: torch.Tensor, *args: Any):
        del anchor  # only there so that the node is part of the graph even when no argument requires grad
        ctx.fn, ctx.kwargs = fn, kwargs
        ctx.is_tensor = tuple(isinstance(a, torch.Tensor) for a in args)
        ctx.plain = tuple(None if t else a for a, t in zip(args, ctx.is_tensor))
        tensors = [a for a in args if isinstance(a, torch.Tensor)]
        ctx.needs = tuple(a.requires_grad for a in tensors)
        ctx.save_for_backward(*tensors)
        ctx.cpu_rng = torch.get_rng_state()
        ctx.device = next((a.device for a in tensors if a.is_cuda), None)
        ctx.gpu_rng = torch.cuda.get_rng_state(ctx.device) if ctx.device is not None else None
        out = fn(*args, **kwargs)  # grad mode is off inside an autograd function's forward
        ctx.single = isinstance(out, torch.Tensor)
        outs = (out,) if ctx.single else tuple(out)
        # An argument handed straight through must come back as a new tensor object, or autograd
        # would treat the output as the (already connected) input.
        ids = {id(a) for a in tensors}
        outs = tuple(o.view_as(o) if isinstance(o, torch.Tensor) and id(o) in ids else o for o in outs)
        return outs[0] if ctx.single else outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        saved = iter(ctx.saved_tensors)
        needs = iter(ctx.needs)
        args, leaves = [], []
        for is_t, plain in zip(ctx.is_tensor, ctx.plain):
            if not is_t:
                args.append(plain)
                continue
            t = next(saved).detach()
            if next(needs):
                t.requires_grad_(True)
            args.append(t)
            leaves.append(t)
        # Second forward under the generator state of the first, then put today's state back.
        now_cpu = torch.get_rng_state()
        now_gpu = torch.cuda.get_rng_state(ctx.device) if ctx.device is not None else None
        torch.set_rng_state(ctx.cpu_rng)
        if ctx.device is not None:
            torch.cuda.set_rng_state(ctx.gpu_rng, ctx.device)
        try:
            with torch.enable_grad(), annotate("recompute forward"):
                out = ctx.fn(*args, **ctx.kwargs)
        finally:
            torch.set_rng_state(now_cpu)
            if ctx.device is not None:
                torch.cuda.set_rng_state(now_gpu, ctx.device)
        outs = (out,) if isinstance(out, torch.Tensor) else tuple(out)
        pairs = [(o, g) for o, g in zip(outs, grads)
                 if isinstance(o, torch.Tensor) and o.requires_grad and g is not None]
        if pairs:
            enter, leave = _bracket()
            enter()
            try:
                torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
            finally:
                leave()
        leaf_grads = iter(t.grad if t.requires_grad else None for t in leaves)
        return (None, None, None) + tuple(next(leaf_grads) if is_t else None for is_t in ctx.is_tensor)


_ANCHOR = {}


def _anchor(device: torch.device) -> torch.Tensor:
    a = _ANCHOR.get(device)
    if a is None:
        a = _ANCHOR[device] = torch.zeros(0, device=device, requires_grad=True)
    return a


def recompute(fn: Callable, *args: Any, **kwargs: Any):
    """``fn(*args, **kwargs)`` with its activations rebuilt in backward (see the module text).

    Tensor arguments must be positional; ``kwargs`` are passed through unchanged to both runs.
    Without grad mode this is a plain call."""
    if not torch.is_grad_enabled():
        return fn(*args, **kwargs)
    device = next((a.device for a in args if isinstance(a, torch.Tensor)), torch.device("cpu"))
    return _RecomputeFn.apply(fn, kwargs, _anchor(device), *args)
