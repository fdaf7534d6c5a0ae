"""Single-process training loop: gradient accumulation, logging, checkpoints.

Constructor arguments, the ``train(dataloader, profiler)`` / ``save_checkpoint`` /
``load_checkpoint`` surface, the log line
``step={n} | loss={:.4f} | lr={:.2e} | time={:.1f}s`` and the checkpoint dictionary
``{model_state_dict, optimizer_state_dict, step[, lr_scheduler_state_dict]}`` follow the
reference trainer (assignments/assignment1/train/trainer.py:10-21,93-98,117-141).

The loop itself is arranged for a GPU that should never wait for the host:

* the loss of every micro-batch is accumulated in a device tensor; on logging steps it is
  copied to pinned host memory on a side stream and the line is printed one optimizer step
  later, when the copy has long completed — the host never drains the compute stream (the
  reference synchronises on every micro-batch: trainer.py:79). Step times for tokens/s come
  from device events recorded at the logging steps;
* when the model supports it, ``model(inputs, targets)`` returns the loss through the
  fused LM-head + cross-entropy op instead of materialising logits;
* with :class:`~pytorch_distributed_amd.optim.FusedAdamW` the gradients are cleared
  inside the optimizer kernel;
* throughput (tokens/s, MFU against the MI355X dense bf16 peak) is appended to the log
  line and optionally written as JSON lines;
* ``merge_micro_batches``: the micro-batch size of the reference's recipe (8 sequences) is a
  memory limit of 80 GB devices, not part of the optimisation problem — the mean loss over 4 x 8
  sequences and its gradient are the mean over 32. With 288 GB per GPU the trainer may therefore
  run several of the loader's micro-batches as ONE forward / backward pass (larger GEMMs, one
  weight-gradient product, one set of FSDP all-gathers per window instead of one per
  micro-batch). ``"auto"`` runs the first window as given, measures what a micro-batch needs and
  merges as many as fit from the second window on; an integer merges that many; 1 keeps the
  loader's micro-batches. The optimizer sees the same gradient up to summation order; dropout
  draws different masks (one Philox stream per pass);
* ``relax_recompute``: selective activation recomputation exists to save memory and returns
  bit-identical gradients, so a model built with ``activation_checkpoint="selective"`` is switched
  to ``"none"`` after the first window when the same measurement says twice the footprint still
  fits (decided together with the merge, agreed between ranks).
"""
from __future__ import annotations

import inspect
import json
import os
import time
from pathlib import Path
from typing import Any, Optional, Union

import torch
import torch.nn as nn

from .. import ops
from ..optim import FusedAdamW
from ..utils.flops import model_flops_per_token, mfu
from ..utils.profiler import annotate


def unwrap_model(model: nn.Module) -> nn.Module:
    """Strips data-parallel wrappers (anything exposing ``.module``)."""
    while hasattr(model, "module") and isinstance(getattr(model, "module"), nn.Module):
        model = model.module
    return model


class Trainer:
    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, lr_scheduler: Optional[Any] = None,
                 max_steps: int = 1000, global_batch_size: int = 32, micro_batch_size: int = 8,
                 log_every_n_steps: int = 10, save_every_n_steps: Optional[int] = None,
                 checkpoint_dir: str = "checkpoints", metrics_file: Optional[str] = None,
                 merge_micro_batches: Union[int, str] = 1, relax_recompute: bool = False):
        self.model = model
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.max_steps = max_steps
        self.global_batch_size = global_batch_size
        self.micro_batch_size = micro_batch_size
        self.log_every_n_steps = log_every_n_steps
        self.save_every_n_steps = save_every_n_steps
        self.checkpoint_dir = checkpoint_dir
        self.metrics_file = metrics_file
        self.world_size = 1
        self.rank = 0
        self.merge_micro_batches = merge_micro_batches
        self.relax_recompute = relax_recompute
        self._merge = 1              # loader micro-batches per forward / backward pass
        self._window_merge = 1       # ... in the window that is running (fixed at its first pass)
        self._merged_now = 1         # ... in the pass that is running (smaller only when the loader ran out)
        self._merge_pending = False  # "auto": still to be decided after the first window
        self._pass_bytes = None      # what the first pass allocated above the resident state
        self._shapes_changed = False # set by the decision, cleared after the first step that ran with it

        if global_batch_size % micro_batch_size != 0:
            raise ValueError(f"Global batch size ({global_batch_size}) must be divisible by "
                             f"micro batch size ({micro_batch_size})")
        self.grad_accumulation_steps = global_batch_size // micro_batch_size

        self.current_step = 0
        self.batch_count = 0
        self.loss_accumulator = None  # device tensor, created on first use
        self.start_time = None
        self._tokens_since_log = 0
        self._time_of_last_log = None
        self._event_of_last_log = None
        self._pending_log = None  # the logging step whose line has not been printed yet
        self._dataloader = None
        self._resume_data_state = None  # stream position from a loaded checkpoint, applied by train()

        if save_every_n_steps is not None:
            os.makedirs(checkpoint_dir, exist_ok=True)

        base = unwrap_model(model)
        try:
            self._fused_loss = "targets" in inspect.signature(base.forward).parameters
        except (TypeError, ValueError):
            self._fused_loss = False
        self._flops_per_token = None

    # ------------------------------------------------------------------ one micro-batch
    def _loss(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        if self._fused_loss:
            return self.model(inputs, targets)
        logits = self.model(inputs)
        return ops.cross_entropy(logits.view(-1, logits.size(-1)), targets.view(-1))

    def training_step(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """Forward + backward of one micro-batch; returns the (unscaled) loss, detached."""
        with annotate("forward"):
            loss = self._loss(inputs, targets)
        with annotate("backward"):
            (loss * self._backward_scale()).backward()
        return loss.detach()

    def _backward_scale(self) -> float:
        """Weight of this pass's mean loss in the window's mean: its share of the window's micro-batches."""
        return self._merged_now / self.grad_accumulation_steps

    # ------------------------------------------------------------------ merged passes
    def _requested_merge(self) -> Optional[int]:
        """Micro-batches per pass asked for by ``merge_micro_batches``; None = decide by measurement."""
        m, accum = self.merge_micro_batches, self.grad_accumulation_steps
        if m in (None, False, 0, 1, "off", "1"):
            return 1
        if m in (True, "all", "on"):
            return accum
        if m == "auto":
            return None
        m = int(m)
        if m < 1 or accum % m != 0:
            raise ValueError(f"merge_micro_batches={m} must divide the {accum} micro-batches of a window")
        return m

    def _agree(self, merge: int, relax: bool) -> tuple:
        """Hook for the distributed trainer: every rank must run the same number of passes."""
        return merge, relax

    @staticmethod
    def _free_bytes(device) -> int:
        """Device memory a new allocation can still get (free at the driver plus cached by the allocator)."""
        if device.type != "cuda":
            return 0
        free, _total = torch.cuda.mem_get_info(device)
        return free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)

    @staticmethod
    def _allocated_now(device) -> Optional[int]:
        """Bytes allocated on ``device`` now (allocator statistics live on the host: no synchronisation)."""
        return torch.cuda.memory_allocated(device) if device.type == "cuda" else None

    @staticmethod
    def _allocated_peak(device) -> int:
        return torch.cuda.max_memory_allocated(device)

    def _after_first_changed_step(self) -> None:
        """Hook: the first step with merged passes (new GEMM shapes, each planned where it was met) is done."""

    def _recompute_owner(self):
        """The module whose ``activation_checkpoint`` attribute sets the model's policy, or None."""
        owner = getattr(unwrap_model(self.model), "transformer", None)
        return owner if isinstance(getattr(owner, "activation_checkpoint", None), str) else None

    def _decide_merge(self, device) -> None:
        """After the first window: the largest divisor of the window whose passes fit the memory that is
        free now (parameters, gradients and optimizer state are all allocated by then), with the first
        pass's footprint as the measure and 15 % of the room left alone; and whether the selective
        recomputation can be dropped on top of that (its activations: taken as twice the footprint)."""
        decide_merge, self._merge_pending = self._merge_pending, False
        accum, merge, relax, room = self.grad_accumulation_steps, self._merge, False, 0
        if self._pass_bytes:
            room = 0.85 * self._free_bytes(device)
            if decide_merge:
                merge = 1
                for cand in range(accum, 1, -1):
                    if accum % cand == 0 and cand * self._pass_bytes * 1.1 <= room:
                        merge = cand
                        break
            owner = self._recompute_owner()
            relax = (self.relax_recompute and owner is not None and owner.activation_checkpoint == "selective"
                     and merge * self._pass_bytes * 1.1 * 2.0 <= room)
        merge, relax = self._agree(merge, relax)
        self._shapes_changed = merge != self._merge or relax  # the next step meets GEMM shapes it has not planned
        self._merge = merge
        if relax:
            self._recompute_owner().activation_checkpoint = "none"
        if self.rank == 0:
            measured = f"[one micro-batch needs {(self._pass_bytes or 0) / 2**30:.1f} GiB, {room / 2**30:.0f} GiB usable]"
            if decide_merge and accum > 1:
                what = (f"{accum // merge} pass(es) of {merge * self.micro_batch_size} sequences"
                        if merge > 1 else "given (no merge)")
                print(f"Accumulation window of {accum} x {self.micro_batch_size} sequences runs as {what} {measured}",
                      flush=True)
            if relax:
                print(f"Selective activation recomputation switched off: everything is kept for backward {measured}",
                      flush=True)

    def _passes(self, batches):
        """The loader's micro-batches grouped into passes: ``(inputs, targets, n)`` with ``n`` micro-batches
        concatenated along the batch dimension, never across the end of an accumulation window and never
        drawn from the loader before the pass is due (a checkpoint written after a step sees the loader
        exactly ``batch_count`` batches in)."""
        it = iter(batches)
        accum = self.grad_accumulation_steps
        self._window_merge = 1  # a window entered in its middle (resumed there) runs as the loader gives it
        while self.current_step < self.max_steps:
            if self.batch_count % accum == 0:
                self._window_merge = self._merge
            group = []
            for _ in range(self._window_merge):
                nxt = next(it, None)
                if nxt is None:
                    break
                group.append(nxt)
            if not group:
                return
            if len(group) == 1:
                yield group[0][0], group[0][1], 1
            else:
                yield torch.cat([g[0] for g in group]), torch.cat([g[1] for g in group]), len(group)

    # ------------------------------------------------------------------ helpers
    def _accumulate_loss(self, loss: torch.Tensor, micro_batches: int = 1) -> None:
        if self.loss_accumulator is None:
            self.loss_accumulator = torch.zeros((), dtype=torch.float32, device=loss.device)
        if micro_batches == 1:
            self.loss_accumulator += loss.float()
        else:  # a merged pass stands for that many micro-batch losses
            self.loss_accumulator.add_(loss.float(), alpha=float(micro_batches))

    def _optimizer_step(self) -> None:
        with annotate("optimizer"):
            if isinstance(self.optimizer, FusedAdamW):
                self.optimizer.step(zero_grad=True)
            else:
                self.optimizer.step()
            self.optimizer.zero_grad()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()

    def _current_lr(self) -> float:
        if self.lr_scheduler is not None:
            return self.lr_scheduler.get_last_lr()[0]
        return self.optimizer.param_groups[0]["lr"]

    def _elapsed(self) -> float:
        return time.time() - self.start_time

    def _throughput_fields(self, pending: dict) -> dict:
        """tokens/s and MFU between the previous logging step and this one (whole job). On a GPU the
        interval is measured between device events recorded at the two logging steps, so no host
        synchronisation is needed to make it exact."""
        fields = {}
        seconds = None
        ev, prev_ev = pending.get("event"), pending.get("prev_event")
        if ev is not None and prev_ev is not None:
            ev.synchronize()
            seconds = prev_ev.elapsed_time(ev) * 1e-3
        elif pending.get("prev_time") is not None:
            if ev is not None:
                ev.synchronize()
                seconds = time.time() - pending["prev_time"]  # first interval: from the first batch to "step done"
            else:
                seconds = pending["host_time"] - pending["prev_time"]
        if seconds and seconds > 0 and pending["tokens"]:
            tps = pending["tokens"] / seconds
            fields["tok/s"] = tps
            if self._flops_per_token is None:
                self._flops_per_token = model_flops_per_token(unwrap_model(self.model), self._seq_len)
            if self._flops_per_token:
                fields["mfu"] = mfu(tps, self._flops_per_token, self.world_size)
        device = pending["device"]
        if device.type == "cuda":
            fields["mem_gb"] = torch.cuda.max_memory_allocated(device) / 2**30
        return fields

    def _queue_log(self, device) -> None:
        """A logging step has been enqueued: start the read-back of its loss and remember what its line
        needs. The line is printed by :meth:`_flush_log` once the NEXT optimizer step has been enqueued
        (or training ends), so reading the value never stalls the stream."""
        self._flush_log()
        pending = {"step": self.current_step, "lr": self._current_lr(), "elapsed": self._elapsed(),
                   "loss": self._window_loss_handle(device), "tokens": self._tokens_since_log, "device": device,
                   "host_time": time.time(), "prev_time": self._time_of_last_log,
                   "prev_event": self._event_of_last_log, "event": None}
        if device.type == "cuda":
            pending["event"] = torch.cuda.Event(enable_timing=True)
            pending["event"].record(torch.cuda.current_stream(device))
        self._event_of_last_log = pending["event"]
        self._time_of_last_log = pending["host_time"]
        self._tokens_since_log = 0
        self._pending_log = pending

    def _flush_log(self) -> None:
        pending, self._pending_log = self._pending_log, None
        if pending is None:
            return
        handle = pending["loss"]
        loss_value = handle.value() if hasattr(handle, "value") else float(handle)
        extra = self._throughput_fields(pending)
        line = (f"step={pending['step']} | loss={loss_value:.4f} | lr={pending['lr']:.2e} | "
                f"time={pending['elapsed']:.1f}s")
        if "tok/s" in extra:
            line += f" | tok/s={extra['tok/s']:.0f}"
        if "mfu" in extra:
            line += f" | mfu={100 * extra['mfu']:.1f}%"
        if "mem_gb" in extra:
            line += f" | mem={extra['mem_gb']:.1f}GB"
        if self.rank == 0:
            print(line, flush=True)
            if self.metrics_file:
                with open(self.metrics_file, "a") as f:
                    f.write(json.dumps({"step": pending["step"], "loss": loss_value, "lr": pending["lr"],
                                        "time_s": pending["elapsed"], **extra}) + "\n")

    def _aggregate_loss(self, loss: float, device) -> float:
        return loss

    def _window_loss_handle(self, device):
        """The logged loss as something with ``.value()``: mean over the accumulation window, read back
        through pinned host memory on a side stream (see ``parallel.comm.LazyScalarMean``)."""
        from ..parallel.comm import LazyScalarMean
        return LazyScalarMean(self.loss_accumulator / self.grad_accumulation_steps, reduce=False)

    def _window_loss(self, device) -> float:
        """Mean loss of the accumulation window, read now (blocks the host)."""
        return self._window_loss_handle(device).value()

    def _maybe_checkpoint(self) -> None:
        """Called after step index ``current_step - 1`` has completed. File names and the trigger
        follow the reference (``checkpoint_step_{index}.pt`` when ``index > 0 and index % n == 0``,
        trainer.py:100-106). The stored ``step`` is the number of completed optimizer steps, i.e.
        the index the resumed run continues with; the reference stores the index it has just
        finished, so a run resumed there repeats that index once."""
        done = self.current_step - 1
        if self.save_every_n_steps is not None and done > 0 and done % self.save_every_n_steps == 0:
            path = f"{self.checkpoint_dir}/checkpoint_step_{done}.pt"
            self.save_checkpoint(path)
            if self.rank == 0:
                print(f"Saved: {path}")

    # ------------------------------------------------------------------ main loop
    def _sync_flag(self, is_last_microbatch: bool):
        """Hook for the distributed trainer (gradient synchronisation only on the last micro-batch)."""
        return None

    def train(self, dataloader, profiler: Optional[Any] = None) -> None:
        self.model.train()
        self.start_time = time.time()
        self._time_of_last_log = None
        if self.rank == 0:
            print(f"Starting training for {self.max_steps} steps")
        if self.grad_accumulation_steps < 1:
            raise ValueError("gradient accumulation steps must be >= 1 "
                             "(global batch smaller than micro batch x world size?)")

        self._dataloader = dataloader
        batches = self._resume_data(dataloader)
        requested = self._requested_merge()
        self._merge = 1 if requested is None else requested
        self._merge_pending = requested is None and self.grad_accumulation_steps > 1
        tune_pending = self._merge_pending or (self.relax_recompute and self._recompute_owner() is not None)
        for inputs, targets, merged in self._passes(batches):
            device = inputs.device
            self._seq_len = inputs.shape[1]
            if self._time_of_last_log is None:
                self._time_of_last_log = time.time()
            before = self._allocated_now(device) if tune_pending and self._pass_bytes is None else None

            self._merged_now = merged
            is_last_microbatch = (self.batch_count + merged) % self.grad_accumulation_steps == 0
            loss = self._run_micro_batch(inputs, targets, is_last_microbatch)
            self._accumulate_loss(loss, merged)
            self._tokens_since_log += inputs.numel() * self.world_size
            self.batch_count += merged
            if before is not None:
                self._pass_bytes = max(1, self._allocated_peak(device) - before)

            if is_last_microbatch:
                self._optimizer_step()
                self._flush_log()  # the previous logging step's line: its values arrived a step ago
                if self.current_step % self.log_every_n_steps == 0:
                    self._queue_log(device)
                self.loss_accumulator.zero_()
                self.current_step += 1
                self._maybe_checkpoint()
                if self._shapes_changed:
                    self._shapes_changed = False
                    self._after_first_changed_step()
                if tune_pending:
                    tune_pending = False
                    self._decide_merge(device)

            if profiler is not None:
                profiler.step()

        self._flush_log()
        if self.rank == 0:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            print(f"Training completed in {self._elapsed():.1f}s")

    def _run_micro_batch(self, inputs, targets, is_last_microbatch: bool) -> torch.Tensor:
        # a model wrapped in parallel.GradAccumulator sums the window's micro-batches in its own
        # buffers and wants to know which backward closes the window
        # Nobody reads a gradient before the window's last backward (no_sync in between), so the linear
        # layers may keep the earlier micro-batches' operands and form each weight gradient once, as one
        # long product over the whole window (ops.linear.accumulation_window).
        index, total = self._window_position()
        no_sync = getattr(self.model, "no_sync", None)
        with ops.accumulation_window(index, total):
            if no_sync is not None and not is_last_microbatch:
                with no_sync():
                    return self.training_step(inputs, targets)
            return self.training_step(inputs, targets)

    def _window_position(self) -> tuple:
        """(index, count) of the running pass among the passes of its accumulation window."""
        accum, merge = self.grad_accumulation_steps, self._window_merge
        return (self.batch_count % accum) // merge, accum // merge

    # ------------------------------------------------------------------ data position
    def _data_state(self) -> dict:
        """Where the token stream stands: micro-batches consumed (and the loader's own state when it
        has one). The reference's checkpoint has no such entry, so a resumed run there re-reads the
        data from the first shard."""
        state = {"batch_count": self.batch_count}
        loader_state = getattr(self._dataloader, "state_dict", None)
        if loader_state is not None:
            state["loader"] = loader_state()
        return state

    def _resume_data(self, dataloader):
        """The iterator ``train`` consumes: positioned after the batches a loaded checkpoint had seen."""
        state, self._resume_data_state = self._resume_data_state, None
        if not state:
            return iter(dataloader)
        done = int(state.get("batch_count", 0))
        if hasattr(dataloader, "load_state_dict"):
            dataloader.load_state_dict(state.get("loader") or {"batches": done})
            return iter(dataloader)
        it = iter(dataloader)  # a loader without a position of its own: read past what was consumed
        for _ in range(done):
            next(it, None)
        return it

    # ------------------------------------------------------------------ checkpoints
    def _model_state_dict(self):
        return self.model.state_dict()

    def _optimizer_state_dict(self):
        return self.optimizer.state_dict()

    def save_checkpoint(self, path: Union[str, Path]) -> None:
        checkpoint = {
            "model_state_dict": self._model_state_dict(),
            "optimizer_state_dict": self._optimizer_state_dict(),
            "step": self.current_step,
            "data_state": self._data_state(),
        }
        if self.lr_scheduler is not None:
            checkpoint["lr_scheduler_state_dict"] = self.lr_scheduler.state_dict()
        if self.rank == 0:
            tmp = f"{path}.tmp{os.getpid()}"
            torch.save(checkpoint, tmp)
            os.replace(tmp, path)

    def load_checkpoint(self, path: Union[str, Path]) -> None:
        device = getattr(self.model, "device", None) or next(self.model.parameters()).device
        checkpoint = torch.load(path, map_location=device, weights_only=True)
        state = checkpoint["model_state_dict"]
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}
        self._load_model_state(state)
        self._load_optimizer_state(checkpoint["optimizer_state_dict"])
        self.current_step = checkpoint.get("step", 0)
        data_state = checkpoint.get("data_state")  # absent in checkpoints written by the reference trainer
        if data_state:
            self.batch_count = int(data_state.get("batch_count", 0))
            self._resume_data_state = data_state
        if self.lr_scheduler is not None and "lr_scheduler_state_dict" in checkpoint:
            self.lr_scheduler.load_state_dict(checkpoint["lr_scheduler_state_dict"])
        if self.rank == 0:
            print(f"Loaded checkpoint from step {self.current_step}")

    def _load_model_state(self, state) -> None:
        self.model.load_state_dict(state)

    def _load_optimizer_state(self, state) -> None:
        self.optimizer.load_state_dict(state)
