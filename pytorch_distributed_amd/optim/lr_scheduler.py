"""Closed-form cosine learning-rate schedule.

Stands in for ``torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min)``
as the reference uses it (assignments/assignment1/train_baseline.py:62-64): same
values at every step, same ``step() / get_last_lr() / state_dict() /
load_state_dict()`` surface (``last_epoch`` and ``_last_lr`` keys), but evaluated in
closed form so resuming from a checkpoint never depends on replaying the recursion.
"""
from __future__ import annotations

import math
from typing import List


class CosineAnnealingLR:
    def __init__(self, optimizer, T_max: int, eta_min: float = 0.0, last_epoch: int = -1):
        self.optimizer = optimizer
        self.T_max = int(T_max)
        self.eta_min = float(eta_min)
        for group in optimizer.param_groups:
            group.setdefault("initial_lr", group["lr"])
        self.base_lrs = [group["initial_lr"] for group in optimizer.param_groups]
        self.last_epoch = last_epoch
        self._last_lr: List[float] = [g["lr"] for g in optimizer.param_groups]
        self.step()  # epoch 0 -> base lr, like torch schedulers do on construction

    def _lr_at(self, base_lr: float, epoch: int) -> float:
        return self.eta_min + (base_lr - self.eta_min) * (1.0 + math.cos(math.pi * epoch / self.T_max)) / 2.0

    def step(self) -> None:
        self.last_epoch += 1
        self._last_lr = [self._lr_at(b, self.last_epoch) for b in self.base_lrs]
        for group, lr in zip(self.optimizer.param_groups, self._last_lr):
            group["lr"] = lr

    def get_last_lr(self) -> List[float]:
        return list(self._last_lr)

    def state_dict(self):
        return {"T_max": self.T_max, "eta_min": self.eta_min, "base_lrs": list(self.base_lrs),
                "last_epoch": self.last_epoch, "_last_lr": list(self._last_lr)}

    def load_state_dict(self, state) -> None:
        self.T_max = int(state.get("T_max", self.T_max))
        self.eta_min = float(state.get("eta_min", self.eta_min))
        self.base_lrs = list(state.get("base_lrs", self.base_lrs))
        self.last_epoch = int(state["last_epoch"])
        self._last_lr = [self._lr_at(b, self.last_epoch) for b in self.base_lrs]
        for group, lr in zip(self.optimizer.param_groups, self._last_lr):
            group["lr"] = lr
