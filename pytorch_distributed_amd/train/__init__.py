from .distributed_trainer import DistributedTrainer
from .trainer import Trainer, unwrap_model

__all__ = ["Trainer", "DistributedTrainer", "unwrap_model"]
