"""Training loop for data-parallel (DDP / FSDP) models.

Same contract as the reference's ``DistributedTrainer``
(assignments/assignment1/train/distributed_trainer.py:33-45): the model must already be
wrapped (anything with a ``no_sync()`` context manager), the process group must be
initialised, ``grad_accumulation_steps = global // (micro * world_size)``, gradients
synchronise only on the last micro-batch of an accumulation window, the logged loss is
averaged over ranks with one scalar all-reduce issued by *all* ranks, and only rank 0
prints.

Departures: a global batch smaller than ``micro * world_size`` is an error instead of a
modulo-by-zero (the reference's hard-coded 32 / (8 x 8 GPUs) = 0); checkpoints gather
the full model / optimizer state collectively on every rank and rank 0 writes, so
sharded models can be saved (a rank-0-only ``state_dict()`` deadlocks under sharding).
"""
from __future__ import annotations

import os
import struct
from typing import Any, Optional

import torch
import torch.nn as nn

from .. import ops
from ..parallel import comm
from ..utils.profiler import annotate
from .trainer import Trainer


class DistributedTrainer(Trainer):
    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, lr_scheduler: Optional[Any] = None,
                 max_steps: int = 1000, global_batch_size: int = 32, micro_batch_size: int = 8,
                 log_every_n_steps: int = 10, save_every_n_steps: Optional[int] = None,
                 checkpoint_dir: str = "checkpoints", ddp_enabled: bool = True,
                 metrics_file: Optional[str] = None, merge_micro_batches=1, relax_recompute: bool = False):
        super().__init__(model=model, optimizer=optimizer, lr_scheduler=lr_scheduler, max_steps=max_steps,
                         global_batch_size=global_batch_size, micro_batch_size=micro_batch_size,
                         log_every_n_steps=log_every_n_steps, save_every_n_steps=save_every_n_steps,
                         checkpoint_dir=checkpoint_dir, metrics_file=metrics_file,
                         merge_micro_batches=merge_micro_batches, relax_recompute=relax_recompute)
        self.ddp_enabled = ddp_enabled
        if ddp_enabled:
            if not comm.is_initialized():
                raise RuntimeError("No communicator is initialized. Call "
                                   "pytorch_distributed_amd.parallel.init_distributed() before creating "
                                   "DistributedTrainer.")
            if not hasattr(model, "no_sync"):
                raise RuntimeError("Model does not have 'no_sync' method. Wrap it with DDP or FSDP before "
                                   "passing it to DistributedTrainer.")
            self.rank = comm.get_rank()
            self.world_size = comm.get_world_size()
        else:
            self.rank, self.world_size = 0, 1

        per_step = micro_batch_size * self.world_size
        if global_batch_size % per_step != 0 or global_batch_size < per_step:
            raise ValueError(f"global batch size {global_batch_size} must be a positive multiple of "
                             f"micro_batch_size x world_size = {per_step}")
        self.grad_accumulation_steps = global_batch_size // per_step
        print(f"DistributedTrainer initialized: rank={self.rank}, world_size={self.world_size}, "
              f"grad_acc_steps={self.grad_accumulation_steps}, ddp_enabled={ddp_enabled}")

    def training_step(self, inputs: torch.Tensor, targets: torch.Tensor, should_sync: bool = True) -> torch.Tensor:
        """Forward + backward; under ``model.no_sync()`` unless this micro-batch closes the window."""
        if self.ddp_enabled and not should_sync:
            with self.model.no_sync():
                with annotate("forward"):
                    loss = self._loss(inputs, targets)
                with annotate("backward (no_sync)"):
                    (loss * self._backward_scale()).backward()
        else:
            with annotate("forward"):
                loss = self._loss(inputs, targets)
            with annotate("backward"):
                (loss * self._backward_scale()).backward()
        return loss.detach()

    def _run_micro_batch(self, inputs, targets, is_last_microbatch: bool) -> torch.Tensor:
        if not self.ddp_enabled:  # every micro-batch synchronises: gradients are read in between
            return self.training_step(inputs, targets, should_sync=True)
        # gradients are read only at the synchronising micro-batch: weight gradients once per window
        index, total = self._window_position()
        with ops.accumulation_window(index, total):
            return self.training_step(inputs, targets, should_sync=is_last_microbatch)

    def _agree(self, merge: int, relax: bool) -> tuple:
        """The ranks' smallest choices: every rank must run the same passes, or their collectives differ."""
        c = comm.get_communicator() if self.ddp_enabled else None
        if c is None or c.world_size == 1:
            return merge, relax
        votes = [struct.unpack("<qq", blob) for blob in c.exchange(struct.pack("<qq", int(merge), int(relax)))]
        return min(v[0] for v in votes), all(v[1] for v in votes)

    def _after_first_changed_step(self) -> None:
        """The merged passes met GEMM shapes that no rank had planned before wrapping; every rank timed the
        library's candidates for itself inside that step. Adopt rank 0's choices everywhere, as the
        planning pass before wrapping does (``plan_sync``): one pace for all ranks."""
        if self.ddp_enabled and self.world_size > 1 and torch.cuda.is_available():
            from ..parallel import plan_sync
            plan_sync.sync_decisions()

    def _aggregate_loss(self, loss: float, device) -> float:
        """Mean of the ranks' window losses. Every rank calls this on logging steps."""
        if not self.ddp_enabled or self.world_size == 1:
            return loss
        return comm.all_reduce_mean_scalar(loss, device)

    def _window_loss_handle(self, device):
        """The logged loss: mean over the accumulation window and over ranks. The 4-byte all-reduce is
        issued on the communicator's reduction queue straight from the device-side accumulator by
        ALL ranks; nobody waits for it here — the line is printed a step later."""
        if not self.ddp_enabled or self.world_size == 1:
            return super()._window_loss_handle(device)
        return comm.LazyScalarMean(self.loss_accumulator / self.grad_accumulation_steps)

    # Checkpoints: state is gathered by all ranks (collective for sharded models); the base
    # class lets only rank 0 write the file.
    def _model_state_dict(self):
        return self.model.state_dict()

    def _optimizer_state_dict(self):
        gather = getattr(self.model, "full_optim_state_dict", None)
        if gather is not None:
            return gather(self.optimizer)
        return self.optimizer.state_dict()

    def _load_optimizer_state(self, state) -> None:
        scatter = getattr(self.model, "load_full_optim_state_dict", None)
        if scatter is not None:
            scatter(self.optimizer, state)
        else:
            self.optimizer.load_state_dict(state)

    def save_checkpoint(self, path) -> None:
        super().save_checkpoint(path)
        if self.ddp_enabled and self.world_size > 1:
            comm.barrier()
