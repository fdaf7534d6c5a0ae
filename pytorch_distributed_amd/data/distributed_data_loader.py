"""Rank-partitioned token-shard loader.

Every rank walks the same shards in lockstep; in each step rank ``r`` of ``W`` takes
the contiguous slice ``tokens[pos + r*b*T : pos + (r+1)*b*T + 1]`` and all ranks advance
``pos`` by ``W*b*T``. Concatenating the ranks' batches in rank order therefore
reproduces exactly what the single-process loader yields for ``W*b`` consecutive
sequences — the partitioning contract of the reference
(assignments/assignment1/data/distributed_data_loader.py:13-24,69-105). A shard is left
when ``pos + W*b*T >= len(shard)``.

Unlike the reference (``except Exception: break``, lines 107-110), errors propagate: a
rank that silently stops iterating while its peers wait in a collective is a hang.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

from .data_loader import KJJ0DataLoader


class DistributedKJJ0DataLoader(KJJ0DataLoader):
    def __init__(self, file_paths: List[Union[str, Path]], local_batch_size: int, sequence_length: int,
                 device: Union[str, torch.device] = "cpu", rank: Optional[int] = None,
                 world_size: Optional[int] = None, verbose: bool = True):
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else int(world_size)
        if not 0 <= self.rank < self.world_size:
            raise ValueError(f"rank {self.rank} outside world of size {self.world_size}")
        super().__init__(file_paths, local_batch_size, sequence_length, device,
                         verbose=verbose and self.rank == 0)
        self.local_batch_size = int(local_batch_size)
        if verbose:
            print(f"Distributed dataloader: rank={self.rank}, world_size={self.world_size}")

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        self._reset()
        b, T, W = self.local_batch_size, self.sequence_length, self.world_size
        local = b * T
        while True:
            while self.current_tokens is None or self.current_position + W * local >= len(self.current_tokens):
                if not self._advance_shard():
                    return
            start = self.current_position + self.rank * local
            window = np.asarray(self.current_tokens[start: start + local + 1])
            self.current_position += W * local
            buf = self._h2d.put(window)
            yield buf[:-1].view(b, T), buf[1:].view(b, T)
