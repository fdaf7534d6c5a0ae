"""Step-scheduled profiling.

``StepProfiler(wait, warmup, active, repeat, trace_path)`` has the contract the reference gets
from ``torch.profiler.profile(schedule=schedule(wait=2, warmup=2, active=6, repeat=1),
on_trace_ready=export_chrome_trace)`` (assignments/assignment1/train_baseline.py:79-87): use it
as a context manager around training, the trainer calls ``.step()`` once per micro-batch,
and a Chrome-trace JSON appears at ``trace_path`` when the active window closes.

Backends:

* ``"kineto"`` — ``torch.profiler`` (roctracer underneath on ROCm): per-kernel GPU rows with
  stream ids, RCCL kernels, ``ProfilerStep#N`` annotations. What ``analyze_traces.py`` wants.
* ``"range"`` — the native range tracer (``csrc/runtime/range_tracer.cpp``): named host + GPU
  ranges only (forward / backward / optimizer / all-gather wait ...), negligible overhead,
  mirrored to roctx so ``rocprofv3 --marker-trace`` sees the same annotations.
* ``"auto"`` — kineto when a GPU is present, range otherwise.

``profiler.range("name")`` annotates a phase under either backend.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from .. import _ext


class StepProfiler:
    def __init__(self, wait: int = 2, warmup: int = 2, active: int = 6, repeat: int = 1,
                 trace_path: Optional[str] = None, backend: str = "auto", rank: int = 0,
                 record_shapes: bool = False):
        if backend == "auto":
            backend = "kineto" if torch.cuda.is_available() else "range"
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
        if trace_path:
            os.makedirs(os.path.dirname(os.path.abspath(trace_path)), exist_ok=True)
        if backend == "kineto":
            from torch.profiler import ProfilerActivity, profile, schedule

            acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if torch.cuda.is_available() else [])
            self._kineto = profile(activities=acts, record_shapes=record_shapes,
                                   schedule=schedule(wait=wait, warmup=warmup, active=active, repeat=repeat),
                                   on_trace_ready=self._kineto_ready)
        else:
            self._tracer = _ext.require().RangeTracer(torch.cuda.is_available())
            self._tracer.set_enabled(False)

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
        if self._kineto is not None:
            self._kineto.__enter__()
        else:
            self._begin_step()
        return self

    def __exit__(self, *exc):
        if self._kineto is not None:
            self._kineto.__exit__(*exc)
        else:
            self._end_step()
            self._flush_range_trace()
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
        if self._tracer.num_ranges and self.trace_path:
            self._tracer.save(self.trace_path, self.rank)
            self.traces_written += 1
        self._tracer.clear()
        self._tracer.set_enabled(False)

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
