"""Gradient accumulation for ONE process on the data-parallel reducer's buffers.

``GradAccumulator(model)`` gives a single-GPU run what DDP gives a multi-GPU one minus the
collectives: every ``.grad`` is a view into a few flat buckets (``csrc/runtime/reducer.cpp`` with no
communicator), and with ``grad_accum_dtype="fp32"`` the micro-batches of an accumulation window are
summed in fp32 twins of those buckets — weight-gradient kernels write the twins directly — and
rounded to the parameter dtype once, at the last micro-batch. That is the arithmetic of the
reference's training loop, whose parameters and gradients are fp32 throughout
(assignments/assignment1/train/trainer.py:57-61: ``(loss / accumulation_steps).backward()`` adding
into fp32 ``.grad``), for a bf16 model.

Use it like DDP: backward under ``no_sync()`` for every micro-batch but the last of a window.
"""
from __future__ import annotations

import contextlib
import functools
from typing import Optional

import torch
from torch import nn

from .. import _ext


class GradAccumulator(nn.Module):
    def __init__(self, module: nn.Module, grad_accum_dtype: Optional[str] = "fp32", bucket_cap_mb: float = 256.0):
        super().__init__()
        if grad_accum_dtype not in (None, "param", "fp32"):
            raise ValueError("grad_accum_dtype must be None / 'param' or 'fp32'")
        self.module = module
        self.grad_accum_dtype = "fp32" if grad_accum_dtype == "fp32" else "param"
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("GradAccumulator: module has no trainable parameters")
        self._params = params
        cap = int(bucket_cap_mb * 2**20)
        self.reducer = _ext.require().GradReducer(params, None, cap, cap, False, False, False,
                                                  self.grad_accum_dtype == "fp32")
        for i, p in enumerate(params):
            p._pda_grad_is_view = True
            if self.reducer.accumulates_fp32(i):
                p._pda_grad_sink = functools.partial(self.reducer.grad_slot, i)
                p._pda_grad_sink_dtype = torch.float32

    def forward(self, *args, **kwargs):
        self.reducer.prepare_forward()
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside only accumulate; the first backward outside closes the window
        (the fp32 sums are rounded into ``.grad``)."""
        previous = self.reducer.sync_enabled()
        self.reducer.set_sync(False)
        try:
            yield
        finally:
            self.reducer.set_sync(previous)

    @property
    def device(self) -> torch.device:
        return self._params[0].device

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        return self.module.load_state_dict(state_dict, *args, **kwargs)

    def zero_grad(self, set_to_none: bool = False):
        self.reducer.zero_grad()

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)
