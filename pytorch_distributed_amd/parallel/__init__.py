from . import autotune, plan_sync
from .comm import (all_reduce, all_reduce_mean_scalar, barrier, broadcast, get_communicator, get_rank,
                   get_world_size, init_distributed, is_initialized, set_communicator, shutdown)
from .ddp import DDP
from .fsdp import FSDP, ShardingStrategy, fsdp_wrap_gpt
from .local import GradAccumulator

DistributedDataParallel = DDP
FullyShardedDataParallel = FSDP

__all__ = ["DDP", "DistributedDataParallel", "FSDP", "FullyShardedDataParallel", "ShardingStrategy",
           "fsdp_wrap_gpt", "init_distributed", "shutdown", "is_initialized", "get_communicator",
           "set_communicator", "get_rank", "get_world_size", "barrier", "all_reduce", "broadcast",
           "all_reduce_mean_scalar", "autotune", "plan_sync", "GradAccumulator"]
