"""One set of GEMM decisions for all ranks, made before the timed / wrapped part of a run.

Two things in the compute path are decided by measurement the first time a shape is seen: the
hipBLASLt algorithm of every planned GEMM (``csrc/runtime/gemm_planner.cpp``) and whether an input
gradient runs on a transposed copy of the weight (``ops/linear.py``). Left alone, every rank
measures for itself — inside its first backward, with collectives in flight — and ranks can
settle on different kernels for the same problem: the slowest choice then sets the pace of every
step, and the first steps are skewed across ranks.

``plan_gemms(model, batch, seq_len)`` therefore runs one forward + backward of the BARE model
(no wrapper, no collective, gradients discarded) on every rank, then replaces each rank's
decisions with rank 0's. Call it before ``DDP(...)`` / ``FSDP(...)``.

The reference has no counterpart: its GEMMs go through ``nn.Linear`` / cuBLAS heuristics
(assignments/assignment1/model/my_gpt2.py:21-23,89-91).
"""
from __future__ import annotations

import json
from typing import Optional

import torch

import importlib

from .. import _ext
from . import comm

_linear = importlib.import_module("pytorch_distributed_amd.ops.linear")  # the module (ops.linear is also a function)


def export_decisions() -> dict:
    ext = _ext.require()
    return {"planner": [list(r) for r in ext.gemm_planner_export()],
            "dgrad": [[list(k[1:]), int(v)] for k, v in _linear._DGRAD_TRANSPOSED.items()]}


def import_decisions(decisions: dict, device_index: Optional[int] = None) -> None:
    ext = _ext.require()
    rows = [tuple(int(x) if i not in (6, 7) else bool(x) for i, x in enumerate(r)) for r in decisions["planner"]]
    ext.gemm_planner_import(rows)
    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    for key, choice in decisions["dgrad"]:
        _linear._DGRAD_TRANSPOSED[(device_index, *[int(x) for x in key])] = int(choice)


def sync_decisions(communicator=None) -> int:
    """Every rank adopts rank 0's decisions. Collective. Returns the number of planned problems."""
    c = communicator if communicator is not None else comm.get_communicator()
    mine = export_decisions()
    if c is None or c.world_size == 1:
        return len(mine["planner"])
    everyone = c.exchange(json.dumps(mine).encode())
    chosen = json.loads(everyone[0].decode())
    if c.rank != 0:
        import_decisions(chosen)
    return len(chosen["planner"])


def plan_gemms(model, micro_batch: int, seq_len: int, vocab_size: Optional[int] = None, communicator=None) -> int:
    """Planning pass on the bare ``model`` (GPU, bf16) followed by :func:`sync_decisions`."""
    p = next(model.parameters())
    if not p.is_cuda:
        return 0
    vocab = vocab_size or getattr(getattr(model, "config", None), "vocab_size", 50257)
    was_training = model.training
    model.train()
    ids = torch.randint(0, vocab, (micro_batch, seq_len + 1), device=p.device)
    loss = model(ids[:, :-1].contiguous(), ids[:, 1:].contiguous())
    loss.backward()
    for q in model.parameters():
        q.grad = None
    del loss, ids
    torch.cuda.synchronize(p.device)
    model.train(was_training)
    return sync_decisions(communicator)
