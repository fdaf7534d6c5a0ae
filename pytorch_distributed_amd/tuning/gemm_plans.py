"""Plan file of the GEMM planner: searched-once kernel choices, reused by every later process.

The planner (``csrc/runtime/gemm_planner.cpp``) can time every hipBLASLt kernel that supports a
problem (exhaustive mode). That costs seconds per shape, so the result — the library's solution
index per (shape, strides, layout, epilogue), plus the direct / transposed route chosen for each
input-gradient shape (``ops/linear.py``) — is written to a JSON file and imported before the first
GEMM of later runs: known shapes then start on their best kernel without any timing, and every rank
of a job starts from the same table. Unknown shapes are planned as before (heuristic candidates,
timed on first use).

No plan file is shipped: measured on MI355X (``profiles/r2/gemm_exhaustive_search.log``), the
kernels that win an exhaustive search in a loop of back-to-back GEMMs are, pinned into a training
step, 0.75 % slower on balance than the planner's in-place choice among the heuristic's
candidates — the ranking of kernels a few percent apart follows the chip's power state, and a
tuning loop is not a training step. A file produced with ``tools/tune_gemms.py`` is used when it
sits next to this module as ``gfx950_gemm_plans.json`` or is named by ``PDA_GEMM_PLAN_FILE``, and
only when its ``library`` tag (hipBLASLt version + architecture) equals the running library's:
solution indices do not carry across builds.

The reference has no counterpart (its GEMMs are ``nn.Linear`` on the cuBLAS heuristic,
assignments/assignment1/model/my_gpt2.py:21-23,89-91).
"""
from __future__ import annotations

import importlib
import json
import os
from typing import Optional

import torch

from .. import _ext

PLAN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gfx950_gemm_plans.json")
_loaded = False


def _linear():
    return importlib.import_module("pytorch_distributed_amd.ops.linear")


def current_table() -> dict:
    """This process's decisions in file form."""
    ext = _ext.require()
    lin = _linear()
    return {"library": ext.gemm_planner_library(),
            "planner": sorted(list(int(x) for x in r) for r in ext.gemm_planner_export()),
            "dgrad": sorted([list(k[1:]), int(v)] for k, v in lin._DGRAD_TRANSPOSED.items())}


def save(path: str = PLAN_FILE, merge: bool = True) -> dict:
    """Writes this process's decisions to ``path`` (merged over what a file of the same library
    build already holds: the newer decision wins per problem)."""
    table = current_table()
    if merge and os.path.exists(path):
        try:
            old = json.load(open(path))
        except (OSError, ValueError):
            old = None
        if old and old.get("library") == table["library"]:
            new_keys = {tuple(r[:9]) for r in table["planner"]}
            table["planner"] = sorted(table["planner"] + [r for r in old.get("planner", []) if tuple(r[:9]) not in new_keys])
            new_d = {tuple(k) for k, _ in table["dgrad"]}
            table["dgrad"] = sorted(table["dgrad"] + [[k, v] for k, v in old.get("dgrad", []) if tuple(k) not in new_d])
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(table, f, indent=0)
    os.replace(tmp, path)
    return table


def load(path: str = PLAN_FILE, device_index: Optional[int] = None) -> int:
    """Imports ``path`` into the planner. Returns the number of problems adopted (0 when the file
    is missing, unreadable or from another library build — never an error)."""
    if not torch.cuda.is_available() or not os.path.exists(path):
        return 0
    try:
        table = json.load(open(path))
        ext = _ext.require()
        if table.get("library") != ext.gemm_planner_library():
            return 0
        rows = [tuple(bool(x) if i in (6, 7) else int(x) for i, x in enumerate(r)) for r in table.get("planner", [])]
        ext.gemm_planner_import(rows)
        lin = _linear()
        if device_index is None:
            device_index = torch.cuda.current_device()
        for key, choice in table.get("dgrad", []):
            lin._DGRAD_TRANSPOSED.setdefault((device_index, *[int(x) for x in key]), int(choice))
        return len(rows)
    except Exception:  # noqa: BLE001 - a bad plan file must never stop training
        return 0


def load_default_once() -> None:
    """Called by ``ops.linear`` before its first planned GEMM."""
    global _loaded
    if _loaded:
        return
    _loaded = True
    if os.environ.get("PDA_NO_GEMM_PLAN_FILE") == "1":
        return
    load(os.environ.get("PDA_GEMM_PLAN_FILE", PLAN_FILE))
