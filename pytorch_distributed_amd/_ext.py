"""Access to the native extension ``_C`` (gfx950 kernels + C++ runtime).

The extension is built in-tree by :mod:`pytorch_distributed_amd._build`. On a
machine with a GPU the HIP kernels are the only compute path: a missing or
unloadable ``_C.so`` is an error there, never a silent fall back to eager
PyTorch. On a CPU-only machine the ops run their PyTorch reference
implementation and only the device-agnostic C++ runtime (reducer, sharding
units, loader, tracer) is used from ``_C``.
"""
from __future__ import annotations

import importlib
import os

_ext = None
_ext_error = None


def load():
    """Return the ``_C`` module, importing it on first use (None if unavailable)."""
    global _ext, _ext_error
    if _ext is None and _ext_error is None:
        try:
            # PyTorch first: _C must bind to the HIP runtime and hipBLASLt that libtorch_hip
            # already mapped (by SONAME), not to a second copy found through its own run path.
            import torch  # noqa: F401
            _ext = importlib.import_module("pytorch_distributed_amd._C")
        except Exception as exc:  # missing file, unresolved symbol, ...
            _ext_error = exc
    return _ext


def require():
    """Return ``_C`` or raise with build instructions."""
    ext = load()
    if ext is None:
        raise RuntimeError(
            "pytorch_distributed_amd._C is not available "
            f"({_ext_error!r}). Build it with `python -m pytorch_distributed_amd._build`."
        )
    return ext


def available() -> bool:
    return load() is not None


def force_reference() -> bool:
    """True when PDA_FORCE_REFERENCE=1 asks every op for its PyTorch reference path."""
    return os.environ.get("PDA_FORCE_REFERENCE", "0") == "1"


def use_hip(t) -> bool:
    """Should an op on tensor ``t`` run the HIP kernel? GPU tensors always do
    (and fail loudly if the extension is missing) unless the reference path is forced."""
    if not t.is_cuda or force_reference():
        return False
    require()
    return True
