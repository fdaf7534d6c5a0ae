"""GEMM plan files.

Every bf16 GEMM of the framework goes through the native planner (``csrc/runtime/gemm_planner.cpp``),
which times the library's candidate algorithms the first time it meets a problem and identifies
its choice by the library's solution index. :mod:`.gemm_plans` exports those choices to a JSON
file and imports them again, so a job can start from decisions measured earlier instead of
re-timing; no plan file ships with the package (timing in place measured as fast as an
exhaustive offline search), and nothing here touches PyTorch's TunableOp.
"""
from . import gemm_plans  # noqa: F401

__all__ = ["gemm_plans"]
