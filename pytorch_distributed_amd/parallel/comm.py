"""Process-group bootstrap and small collective helpers.

One process per GPU, launched by ``torchrun`` (``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` /
``MASTER_ADDR`` / ``MASTER_PORT`` in the environment, as the reference's scripts expect:
assignments/assignment1/train_ddp.py:23-36). On GPUs the backend is ``"nccl"``, which on ROCm
is RCCL over xGMI; without a GPU it is ``"gloo"``, so every distributed code path of this
package also runs in CPU-only tests — the reference hard-codes ``'nccl'`` and cannot.

Beyond the reference: collectives get a finite timeout (a rank that dies turns into an
error on its peers instead of a hang), and ``PDA_COMM_DEBUG=1`` cross-checks bulk
collectives between ranks before they are issued.
"""
from __future__ import annotations

import datetime
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def init_distributed(backend: Optional[str] = None, timeout_s: float = 600.0) -> Tuple[int, int, torch.device]:
    """Initialises the default process group from the torchrun environment.

    Returns ``(rank, world_size, device)``. Safe to call when already initialised.
    """
    rank, world, local_rank = env_rank_world()
    use_gpu = torch.cuda.is_available() and backend != "gloo"
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = dict(backend=backend, rank=rank, world_size=world,
                      timeout=datetime.timedelta(seconds=timeout_s))
        if use_gpu:
            kwargs["device_id"] = device  # binds the communicator to this GPU at init time
        dist.init_process_group(**kwargs)
    if rank == 0:
        print(f"Initialized process group: world_size={world}, backend={backend}")
    return rank, world, device


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


def default_group():
    """The default ProcessGroup object (what the C++ reducer / sharding units take), or None."""
    if not dist.is_initialized():
        return None
    return dist.distributed_c10d._get_default_group()


def comm_debug() -> bool:
    return os.environ.get("PDA_COMM_DEBUG", "0") == "1"


def assert_same_across_ranks(values, what: str, group=None) -> None:
    """Debug check: every rank must hold the same list of integers (shapes, counts, ...)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    mine = torch.tensor(list(values), dtype=torch.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    mine = mine.to(dev)
    gathered = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(gathered, mine, group=group)
    for r, other in enumerate(gathered):
        if not torch.equal(other, mine):
            raise RuntimeError(f"{what} differs between rank {dist.get_rank(group)} and rank {r}: "
                               f"{mine.tolist()[:8]}... vs {other.tolist()[:8]}...")


def all_reduce_mean_scalar(value: float, device) -> float:
    """Mean of a host scalar over all ranks (the logging all-reduce of the reference,
    assignments/assignment1/train/distributed_trainer.py:131-154)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()
