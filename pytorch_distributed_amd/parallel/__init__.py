from .comm import all_reduce_mean_scalar, default_group, init_distributed, shutdown
from .ddp import DDP
from .fsdp import FSDP, ShardingStrategy, fsdp_wrap_gpt

DistributedDataParallel = DDP
FullyShardedDataParallel = FSDP

__all__ = ["DDP", "DistributedDataParallel", "FSDP", "FullyShardedDataParallel", "ShardingStrategy",
           "fsdp_wrap_gpt", "init_distributed", "shutdown", "default_group", "all_reduce_mean_scalar"]
