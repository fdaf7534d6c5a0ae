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
                 world_size: Optional[int] = None, verbose: bool = True, prefetch: bool = False):
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else int(world_size)
        if not 0 <= self.rank < self.world_size:
            raise ValueError(f"rank {self.rank} outside world of size {self.world_size}")
        super().__init__(file_paths, local_batch_size, sequence_length, device,
                         verbose=verbose and self.rank == 0)
        self.local_batch_size = int(local_batch_size)
        # prefetch: the C++ TokenPrefetcher (csrc/runtime/token_loader.cpp) maps the shards and a
        # worker thread widens this rank's next windows into a ring of (pinned) int64 buffers
        # while the GPU trains, so the Python thread only issues one async copy per batch.
        self.prefetch = bool(prefetch)
        if verbose:
            print(f"Distributed dataloader: rank={self.rank}, world_size={self.world_size}")

    def _iter_prefetched(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        from .. import _ext

        ext = _ext.require()
        b, T = self.local_batch_size, self.sequence_length
        on_gpu = torch.device(self.device).type == "cuda"
        slots = [torch.empty(b * T + 1, dtype=torch.int64, pin_memory=on_gpu) for _ in range(4)]
        prefetcher = ext.TokenPrefetcher(self.files, b, T, self.rank, self.world_size, slots)
        in_flight = []  # (slot index, event recorded after its host-to-device copy)
        skip, self._start_batch = self._start_batch, 0
        self._yielded = 0
        try:
            while skip > 0:  # resuming: windows already consumed are taken from the ring and dropped
                idx = prefetcher.next()
                if idx < 0:
                    return
                prefetcher.release(idx)
                self._yielded += 1
                skip -= 1
            while True:
                # hand back slots whose copy has finished; if the ring is full wait for the oldest
                while in_flight and (in_flight[0][1].query() or len(in_flight) >= len(slots) - 1):
                    idx, ev = in_flight.pop(0)
                    ev.synchronize()
                    prefetcher.release(idx)
                idx = prefetcher.next()
                if idx < 0:
                    return
                if on_gpu:
                    buf = slots[idx].to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    in_flight.append((idx, ev))
                else:
                    buf = slots[idx].clone()
                    prefetcher.release(idx)
                self._yielded += 1
                yield buf[:-1].view(b, T), buf[1:].view(b, T)
        finally:
            prefetcher.stop()

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        if self.prefetch:
            yield from self._iter_prefetched()
            return
        self._reset()
        b, T, W = self.local_batch_size, self.sequence_length, self.world_size
        local = b * T
        skip, self._start_batch = self._start_batch, 0
        self._yielded = 0
        while True:
            while self.current_tokens is None or self.current_position + W * local >= len(self.current_tokens):
                if not self._advance_shard():
                    return
            start = self.current_position + self.rank * local
            self.current_position += W * local
            self._yielded += 1
            if skip > 0:  # resuming: only the cursor moves
                skip -= 1
                continue
            window = np.asarray(self.current_tokens[start: start + local + 1])
            buf = self._h2d.put(window)
            yield buf[:-1].view(b, T), buf[1:].view(b, T)
