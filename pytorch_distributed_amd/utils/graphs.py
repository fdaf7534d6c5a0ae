"""Whole-step HIP graphs for launch-bound training.

A training step of a small model (GPT-2 small at the reference's batch of 8 sequences:
assignments/assignment0/throughput.py:197-199) is a few thousand kernels of a few
microseconds each; the GPU finishes them faster than one CPU thread can launch them, so the
step time is the launch time. :class:`GraphedTrainStep` records ONE full optimizer step —
every micro-batch's forward and backward plus the fused AdamW update — into a HIP graph
(``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replays it with one launch per step.

What makes the step replayable:

* every kernel of this package launches on the current stream, allocates through the caching
  allocator and never synchronises (the GEMM planner only times candidates outside a
  capture; the warm-up steps before the capture make its choices);
* :class:`~pytorch_distributed_amd.optim.FusedAdamW` in ``capturable`` mode keeps the step
  counter, bias corrections and learning rate in device memory;
* inputs are copied into static buffers before each replay.

Limits: single process (collectives of the data-parallel wrappers are not captured),
fixed shapes, every dropout off: the Philox seeds of attention dropout and of the residual /
embedding dropout fused into the LayerNorm kernels are drawn on the host, so a capture would
bake one seed in and every replay would repeat the same masks.
"""
from __future__ import annotations

from typing import Optional

import torch


class GraphedTrainStep:
    """``step(inputs, targets) -> mean loss`` where ``inputs``/``targets`` are
    ``[accum * micro_batch, T]`` token tensors; one replayed graph per optimizer step."""

    def __init__(self, model: torch.nn.Module, optimizer, micro_batch: int, seq_len: int, accum: int = 1,
                 warmup_steps: int = 3, lr_scheduler=None, device: Optional[torch.device] = None):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FusedAdamW(..., capturable=True)")
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.micro_batch, self.seq_len, self.accum = micro_batch, seq_len, accum
        self.device = device if device is not None else next(model.parameters()).device
        shape = (accum, micro_batch, seq_len)
        self._inputs = torch.zeros(shape, dtype=torch.long, device=self.device)
        self._targets = torch.zeros(shape, dtype=torch.long, device=self.device)
        self._loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._warmup_left = warmup_steps
        if model.training:
            for name, m in model.named_modules():
                p = max(float(getattr(m, "attn_pdrop", 0.0) or 0.0), float(getattr(m, "out_pdrop", 0.0) or 0.0),
                        float(m.p) if isinstance(m, torch.nn.Dropout) else 0.0)
                if p > 0.0:
                    raise ValueError(f"GraphedTrainStep: dropout (p={p} in '{name or type(m).__name__}') draws its "
                                     "Philox seed on the host; a captured step would replay one mask for ever. "
                                     "Set every dropout probability to 0 (config.without_dropout()).")

    def _eager_step(self) -> None:
        from .. import ops
        total = None
        for i in range(self.accum):
            # the window's weight gradients are formed once, from the kept operands, inside the graph as outside it
            with ops.accumulation_window(i, self.accum):
                loss = self.model(self._inputs[i], self._targets[i]) / self.accum
                loss.backward()
            total = loss.detach().float() if total is None else total + loss.detach().float()
        self.optimizer.step(zero_grad=True)
        self._loss.copy_(total)

    def __call__(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        self._inputs.copy_(inputs.view_as(self._inputs))
        self._targets.copy_(targets.view_as(self._targets))
        if self._graph is None and self._warmup_left > 0:
            # Warm-up on a side stream (allocator and library state settle, the GEMM planner
            # picks its algorithms), exactly as torch.cuda.graphs prescribes.
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._eager_step()
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._warmup_left -= 1
        elif self._graph is None:
            self._graph = torch.cuda.CUDAGraph()
            self.optimizer.refresh_capture_scalars()
            with torch.cuda.graph(self._graph):
                self._eager_step()
            # the capture itself executed nothing: run the recorded step once
            self._graph.replay()
        else:
            self.optimizer.refresh_capture_scalars()
            self._graph.replay()
            self.optimizer.note_replayed_steps(1)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return self._loss
