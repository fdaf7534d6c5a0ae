"""Step-scheduled profiling.

``StepProfiler(wait, warmup, active, repeat, trace_path)`` has the contract the reference gets
from ``torch.profiler.profile(schedule=schedule(wait=2, warmup=2, active=6, repeat=1),
on_trace_ready=export_chrome_trace)`` (assignments/assignment1/train_baseline.py:79-87): use it
as a context manager around training, the trainer calls ``.step()`` once per micro-batch,
and a Chrome-trace JSON appears at ``trace_path`` when the active window closes.

Backends:

* ``"range"`` (the default) — the framework's own tracer (``csrc/runtime/range_tracer.cpp``):
  ``ProfilerStep#N`` plus named host + GPU ranges for forward / backward / optimizer (emitted by
  the trainers through :func:`annotate`) and for all-gather wait / reduce-scatter launch /
  end-of-backward reduction wait (emitted by the C++ DDP / FSDP runtime itself), AND one row per
  HIP queue with every kernel dispatch and memory copy of the active window, collectives on a
  row of their own: ``csrc/runtime/kernel_tracer.cpp`` collects the GPU activity records in
  process, switched on and off by this schedule, so the plain training command writes the
  kernel-level trace (``kernels=False`` keeps ranges only). Every range is also mirrored to
  roctx: when the run is wrapped in ``rocprofv3 --kernel-trace --marker-trace`` (the route to
  hardware counters) the in-process collection stays off and ``utils/rocprof_trace.py``
  (``analyze_traces.py --from-rocprof``) merges rocprofv3's kernel rows with these ranges.
* ``"kineto"`` — ``torch.profiler`` (roctracer underneath on ROCm), kept for comparison.
* ``"auto"`` — the same as ``"range"``.

``profiler.range("name")`` / module-level ``annotate("name")`` annotate a phase under either backend.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from .. import _ext


_ACTIVE: Optional["StepProfiler"] = None


@contextlib.contextmanager
def annotate(name: str):
    """Named range in the active :class:`StepProfiler` (no-op when none is active): what the
    trainers and wrappers call, so their phases show up without holding a profiler object."""
    prof = _ACTIVE
    if prof is None:
        yield
        return
    with prof.range(name):
        yield


class StepProfiler:
    def __init__(self, wait: int = 2, warmup: int = 2, active: int = 6, repeat: int = 1,
                 trace_path: Optional[str] = None, backend: str = "auto", rank: int = 0,
                 record_shapes: bool = False, kernels: bool = True):
        if backend == "auto":
            backend = "range"
        if backend not in ("kineto", "range"):
            raise ValueError("backend must be 'auto', 'kineto' or 'range'")
        self.wait, self.warmup, self.active, self.repeat = wait, warmup, active, repeat
        self.trace_path = trace_path
        self.backend = backend
        self.rank = rank
        self.step_num = 0
        self.traces_written = 0
        self._kineto = None
        self._tracer = None
        self._step_open = False
        self.collects_kernels = False
        self.kernels_written = 0
        if trace_path:
            os.makedirs(os.path.dirname(os.path.abspath(trace_path)), exist_ok=True)
        if backend == "kineto":
            from torch.profiler import ProfilerActivity, profile, schedule

            acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if torch.cuda.is_available() else [])
            self._kineto = profile(activities=acts, record_shapes=record_shapes,
                                   schedule=schedule(wait=wait, warmup=warmup, active=active, repeat=repeat),
                                   on_trace_ready=self._kineto_ready)
        else:
            self._tracer = _ext.require().RangeTracer(torch.cuda.is_available(), bool(kernels))
            self._tracer.set_enabled(False)
            self.collects_kernels = bool(self._tracer.collects_kernels)
            if kernels and torch.cuda.is_available() and not self.collects_kernels and rank == 0:
                print("[profiler] kernel rows unavailable (%s): the trace will hold ranges only"
                      % _ext.require().kernel_trace_unavailable_reason(), flush=True)

    # -- schedule ------------------------------------------------------------------------
    def _phase(self, step: int) -> str:
        cycle = self.wait + self.warmup + self.active
        if self.repeat and step >= cycle * self.repeat:
            return "done"
        k = step % cycle
        if k < self.wait:
            return "wait"
        if k < self.wait + self.warmup:
            return "warmup"
        return "active"

    def _kineto_ready(self, prof) -> None:
        if self.trace_path:
            prof.export_chrome_trace(self.trace_path)
            self.traces_written += 1

    # -- context manager -----------------------------------------------------------------
    def __enter__(self):
        global _ACTIVE
        _ACTIVE = self
        if self._kineto is not None:
            self._kineto.__enter__()
        else:
            self._tracer.activate()  # the C++ runtime's own ranges land in this tracer
            self._tracer.push(f"pda.rank={self.rank}", "meta")  # lets a rocprofv3 marker trace name its rank
            self._tracer.pop()
            self._begin_step()
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        if self._kineto is not None:
            self._kineto.__exit__(*exc)
        else:
            self._end_step()
            self._flush_range_trace()
            self._tracer.deactivate()
        if _ACTIVE is self:
            _ACTIVE = None
        return False

    def _begin_step(self) -> None:
        if self._phase(self.step_num) == "active":
            self._tracer.set_enabled(True)
            self._tracer.push(f"ProfilerStep#{self.step_num}", "user_annotation")
            self._step_open = True

    def _end_step(self) -> None:
        if self._step_open:
            self._tracer.pop()
            self._step_open = False

    def _flush_range_trace(self) -> None:
        self._tracer.set_enabled(False)  # closes the kernel-collection window: drains the device, takes the records
        if self._tracer.num_ranges and self.trace_path:
            self._tracer.save(self.trace_path, self.rank)
            self.traces_written += 1
            self.kernels_written = int(self._tracer.num_kernels)
        self._tracer.clear()

    def step(self) -> None:
        """Marks the end of one profiling step (the trainer calls this once per micro-batch)."""
        if self._kineto is not None:
            self._kineto.step()
            self.step_num += 1
            return
        self._end_step()
        was_active = self._phase(self.step_num) == "active"
        self.step_num += 1
        if was_active and self._phase(self.step_num) != "active":
            self._flush_range_trace()
        self._begin_step()

    @contextlib.contextmanager
    def range(self, name: str):
        """Annotates a named phase (a record_function under kineto, a host+GPU range otherwise)."""
        if self._kineto is not None:
            with torch.profiler.record_function(name):
                yield
        else:
            self._tracer.push(name, "phase")
            try:
                yield
            finally:
                self._tracer.pop()


@contextlib.contextmanager
def null_profiler():
    yield None
