"""Pre-tuned hipBLASLt / rocBLAS solution choices for the GEMM shapes of the benchmark models.

PyTorch's TunableOp can time every library solution for a GEMM shape and remember the
fastest; ``tunableop_gfx950.csv`` holds the result of that search on an MI355X for the
(forward, dgrad, wgrad) shapes of the shipped configurations. ``enable_tuned_gemms()``
switches TunableOp to lookup-only mode with that file, so training uses the tuned choices
without paying for a search; shapes that are not in the file use the library default.
``tune(path)`` re-runs the search for whatever GEMMs execute afterwards.

The file is only honoured when its validator lines (PyTorch, HIP, hipBLASLt, rocBLAS
versions, GPU architecture) match the running stack.
"""
from __future__ import annotations

import os

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_tuned_gemms(path: str = TUNED_FILE) -> bool:
    """Lookup-only TunableOp with the shipped results. Returns False when unavailable."""
    if not torch.cuda.is_available() or not os.path.exists(path) or os.environ.get("PDA_NO_TUNED_GEMMS") == "1":
        return False
    try:
        tunable = torch.cuda.tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        if hasattr(tunable, "write_file_on_exit"):
            tunable.write_file_on_exit(False)
        tunable.set_filename(path)
        return bool(tunable.read_file(path))
    except Exception:  # an incompatible file must never stop training
        try:
            torch.cuda.tunable.enable(False)
        except Exception:
            pass
        return False


def tune(path: str, max_tuning_duration_ms: int = 60) -> None:
    """Search mode: every new GEMM shape executed from now on is tuned and recorded in ``path``."""
    tunable = torch.cuda.tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(max_tuning_duration_ms)
    tunable.set_filename(path)
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(True)
