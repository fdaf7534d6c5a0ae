"""Training-memory accounting (assignment-0 ``memory_analysis`` parity).

``estimate_training_memory`` is the analytic model of the reference
(assignments/assignment0/memory_analysis.py:12-52: parameters, gradients of the same size,
two Adam moments, activations ignored) generalised to the mixed-precision layout this
framework trains with: bf16 parameters + bf16 gradients + fp32 master + two fp32 moments
= 16 bytes per parameter as well, but split differently — and divided by the world size
for whatever the chosen sharding strategy shards.
"""
from __future__ import annotations

from typing import Dict

MiB = float(2 ** 20)


def estimate_training_memory(num_params: int, dtype: str = "fp32", world_size: int = 1,
                             strategy: str = "none") -> Dict[str, float]:
    """Megabytes of parameters / gradients / optimizer state per GPU (activations excluded).

    ``strategy``: ``none`` / ``ddp`` / ``NO_SHARD`` (replicated), ``SHARD_GRAD_OP`` (gradients and
    optimizer state sharded), ``FULL_SHARD`` (everything sharded).
    """
    if dtype in ("fp32", "float32"):
        p_bytes, g_bytes, opt_bytes = 4, 4, 8
    elif dtype in ("bf16", "bfloat16"):
        p_bytes, g_bytes, opt_bytes = 2, 2, 12  # fp32 master + exp_avg + exp_avg_sq
    else:
        raise ValueError(f"unknown dtype {dtype!r}")
    w = max(1, world_size)
    shard_params = strategy == "FULL_SHARD"
    shard_rest = strategy in ("FULL_SHARD", "SHARD_GRAD_OP")
    params_mb = num_params * p_bytes / MiB / (w if shard_params else 1)
    grads_mb = num_params * g_bytes / MiB / (w if shard_rest else 1)
    opt_mb = num_params * opt_bytes / MiB / (w if shard_rest else 1)
    return {"total_params": num_params, "params_mb": params_mb, "gradients_mb": grads_mb,
            "optimizer_mb": opt_mb, "total_mb": params_mb + grads_mb + opt_mb}
