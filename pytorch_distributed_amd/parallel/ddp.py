"""DistributedDataParallel on the native gradient-bucket reducer.

``DDP(model, device_ids=[local_rank])`` keeps the constructor shape the reference uses
(assignments/assignment1/train_ddp.py:46-49) and the wrapper surface its trainer relies on:
``no_sync()``, ``parameters()``, ``state_dict()`` / ``load_state_dict()``, ``train()`` /
``eval()``, call-through ``forward`` — plus ``.device`` and unprefixed ``state_dict`` keys so
checkpoints load into the bare model and vice versa.

All per-gradient work happens in C++ (``csrc/runtime/reducer.cpp``): gradients are views of
flat buckets, buckets are reduced on the native communicator's reduction stream
(``csrc/runtime/comm``: RCCL on GPUs, the TCP mesh on CPUs) from autograd hooks while backward
is still running, and the optimizer is fenced behind the collectives by an end-of-backward
stream wait. ``bucket_cap_mb=None`` / ``decompose=None`` take the values the start-up
measurement chose (``parallel/autotune.py``); explicit values always win. Not reproduced from torch DDP because the model gives it no purpose: the
per-forward buffer broadcast (this GPT-2 has no buffers — the reference's ``causal_mask``
buffers cost 36 MiB of broadcast per forward) and the unused-parameter graph search
(buckets whose parameters got no gradient are simply flushed at the end of backward).
"""
from __future__ import annotations

import contextlib
import functools
from typing import Optional

import torch
from torch import nn

from .. import _ext
from . import comm

DEFAULT_BUCKET_CAP_MB = 64.0
DEFAULT_FIRST_BUCKET_MB = 4.0


def _bucket_slot(p):
    """(gradient tensor to accumulate into, accumulate=True) or None when ``p.grad`` was dropped."""
    g = p.grad
    return None if g is None else (g, True)


class DDP(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, process_group=None,
                 bucket_cap_mb: Optional[float] = None, first_bucket_mb: float = DEFAULT_FIRST_BUCKET_MB,
                 broadcast_parameters: bool = True, force_collectives: bool = False, reduce_fp32: bool = False,
                 decompose: Optional[bool] = None, communicator=None, grad_accum_dtype: Optional[str] = None):
        """``reduce_fp32``: all-reduce bf16 gradient buckets through an fp32 copy (twice the bytes on
        the wire, no rounding between the ranks' contributions); default: the gradient dtype.
        ``decompose``: reduce each bucket as reduce-scatter + all-gather instead of one all-reduce.
        ``process_group``: a torch.distributed group to run on instead of the native communicator.
        ``grad_accum_dtype="fp32"``: low-precision (bf16) gradients are summed over the micro-batches
        of a ``no_sync`` window in fp32 twins of the buckets instead of being rounded to bf16 after
        every addition (the reference accumulates in fp32: assignments/assignment1/train/trainer.py:57-61);
        the sum is narrowed (or, with ``reduce_fp32``, reduced as fp32 and then narrowed) at the
        synchronising backward. ``.grad`` then only exists from that backward to the next forward."""
        super().__init__()
        self.module = module
        self.communicator = comm.resolve(communicator, process_group)
        self.world_size = self.communicator.world_size
        self.rank = self.communicator.rank
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("DDP: the module has no trainable parameters")
        if device_ids is not None and len(device_ids) == 1 and params[0].is_cuda:
            want = torch.device("cuda", device_ids[0]) if isinstance(device_ids[0], int) else torch.device(device_ids[0])
            if params[0].device != want:
                raise ValueError(f"DDP: module lives on {params[0].device}, device_ids says {want}")
        self._params = params
        if params[0].is_cuda != self.communicator.on_gpu:
            raise RuntimeError(f"DDP: the '{self.communicator.backend}' communicator cannot move "
                               f"{'GPU' if params[0].is_cuda else 'CPU'} tensors")

        from . import autotune
        tuned = autotune.current_policy()
        if bucket_cap_mb is None:
            bucket_cap_mb = tuned.ddp_bucket_mb if tuned is not None else DEFAULT_BUCKET_CAP_MB
        if decompose is None:
            decompose = tuned.ddp_decompose if tuned is not None else False
        self.bucket_cap_mb, self.decompose = float(bucket_cap_mb), bool(decompose)

        # Same shapes everywhere, then rank 0's values everywhere (parameters and buffers).
        comm.assert_same_across_ranks([p.numel() for p in params], "parameter sizes", self.communicator)
        if broadcast_parameters and self.world_size > 1:
            tensors = [p.data for p in params] + [b.data for b in module.buffers()]
            comm.broadcast_coalesced(tensors, 0, self.communicator)

        ext = _ext.require()
        if grad_accum_dtype not in (None, "param", "fp32"):
            raise ValueError("grad_accum_dtype must be None / 'param' or 'fp32'")
        self.grad_accum_dtype = "fp32" if grad_accum_dtype == "fp32" else "param"
        self.reducer = ext.GradReducer(params, self.communicator, int(bucket_cap_mb * 2**20),
                                       int(first_bucket_mb * 2**20), force_collectives, reduce_fp32, self.decompose,
                                       self.grad_accum_dtype == "fp32")
        for i, p in enumerate(params):
            p._pda_grad_is_view = True  # optimizers must zero these in place, never drop them
            # ops.linear / ops.norm add their parameter gradients straight into the bucket view
            # instead of materialising them for autograd to add (the hook on the accumulator
            # still fires); under fp32 accumulation the target is the bucket's fp32 twin.
            if self.reducer.accumulates_fp32(i):
                p._pda_grad_sink = functools.partial(self.reducer.grad_slot, i)
                p._pda_grad_sink_dtype = torch.float32
            else:
                p._pda_grad_sink = functools.partial(_bucket_slot, p)

    # -- module surface ----------------------------------------------------------------
    def forward(self, *args, **kwargs):
        self.reducer.prepare_forward()  # closes a pass left open by a backward that raised
        return self.module(*args, **kwargs)

    @property
    def device(self) -> torch.device:
        return self._params[0].device

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate into the bucket views without communicating."""
        previous = self.reducer.sync_enabled()
        self.reducer.set_sync(False)
        try:
            yield
        finally:
            self.reducer.set_sync(previous)

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        state_dict = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        return self.module.load_state_dict(state_dict, *args, **kwargs)

    def zero_grad(self, set_to_none: bool = False):
        self.reducer.zero_grad()

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)
