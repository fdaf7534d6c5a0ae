from .adamw import FusedAdamW
from .lr_scheduler import CosineAnnealingLR

AdamW = FusedAdamW

__all__ = ["FusedAdamW", "AdamW", "CosineAnnealingLR"]
