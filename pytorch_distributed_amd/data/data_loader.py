"""Sequential token-shard loader (single process) and dataset provisioning.

``KJJ0DataLoader(file_paths, batch_size, sequence_length, device)`` yields
``(inputs, targets)`` int64 ``[B, T]`` batches with the window arithmetic of the
reference loader (assignments/assignment1/data/data_loader.py:137-195): sequence ``i``
of the stream is ``tokens[i*T : i*T + T + 1]`` (inputs = first T, targets = last T);
a shard is left as soon as ``pos + T >= len(shard)`` and iteration ends with the last
shard.

Mechanics differ: shards are memory-mapped, a batch is one contiguous slice (plus
one per shard boundary), and tokens cross PCIe as pinned int32 with a non-blocking
copy, widened to int64 on the device.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

from . import shards

DATASET_REPO = "kjj0/fineweb10B-gpt2"
DEFAULT_DATA_DIR = ".cache/data/fineweb10B"
MAX_TRAIN_FILES = 103


def _expected_files(num_train_files: Optional[int]) -> List[str]:
    n = MAX_TRAIN_FILES if num_train_files is None else int(num_train_files)
    if not 0 <= n <= MAX_TRAIN_FILES:
        raise ValueError(f"num_train_files must be in [0, {MAX_TRAIN_FILES}]")
    return ["fineweb_val_000000.bin"] + [f"fineweb_train_{i:06d}.bin" for i in range(1, n + 1)]


def download_fineweb10B_files(local_dir: Union[str, Path] = DEFAULT_DATA_DIR,
                              num_train_files: Optional[int] = None) -> List[Path]:
    """Returns the validation shard plus ``num_train_files`` training shards of
    fineweb10B-gpt2 under ``local_dir``, fetching the missing ones from the hub.

    Only LOCAL_RANK 0 (or a non-distributed process) downloads; files appear under their
    final name atomically, and the other ranks of a node wait for them instead of racing.
    """
    local_dir = Path(local_dir)
    local_dir.mkdir(parents=True, exist_ok=True)
    names = _expected_files(num_train_files)
    missing = [n for n in names if not (local_dir / n).exists()]
    if missing and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        try:
            from huggingface_hub import hf_hub_download
        except ImportError as exc:  # pragma: no cover
            raise RuntimeError("huggingface_hub is needed to download the dataset") from exc
        for name in missing:
            print(f"  Downloading {name}...")
            try:
                hf_hub_download(repo_id=DATASET_REPO, filename=name, repo_type="dataset", local_dir=local_dir)
            except Exception as exc:
                raise RuntimeError(
                    f"could not fetch {name} from {DATASET_REPO} ({exc}); without network access use "
                    "make_synthetic_fineweb_files() or the --synthetic-data flag of the train scripts"
                ) from exc
    elif missing:
        import time

        deadline = time.time() + 3600
        while any(not (local_dir / n).exists() for n in missing):
            if time.time() > deadline:
                raise TimeoutError(f"timed out waiting for {missing} in {local_dir}")
            time.sleep(1.0)
    print(f"Using {len(names)} shard files in {local_dir}")
    return [local_dir / n for n in names]


def make_synthetic_fineweb_files(local_dir: Union[str, Path], num_train_files: int = 1,
                                 tokens_per_file: int = 1 << 20, vocab_size: int = 50257,
                                 seed: int = 0) -> List[Path]:
    """Writes random-token shards with the dataset's file names and binary format."""
    local_dir = Path(local_dir)
    out = []
    for k, name in enumerate(_expected_files(num_train_files)):
        path = local_dir / name
        if not path.exists():
            rng = np.random.default_rng(seed + k)
            shards.write_shard(path, rng.integers(0, min(vocab_size, 65536), size=tokens_per_file, dtype=np.int64))
        out.append(path)
    return out


class _HostToDevice:
    """Stages int32 token windows in pinned memory and copies them without blocking the host."""

    def __init__(self, device, n_elems: int, depth: int = 4):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.slots = []
        self.next = 0
        if self.cuda:
            for _ in range(depth):
                buf = torch.empty(n_elems, dtype=torch.int32, pin_memory=True)
                self.slots.append([buf, None])  # [pinned buffer, event of the last copy out of it]

    def put(self, window: np.ndarray) -> torch.Tensor:
        """``window`` (uint16 / integer array) -> int64 tensor on the target device."""
        if not self.cuda:
            return torch.from_numpy(window.astype(np.int64))
        buf, ev = self.slots[self.next]
        if ev is not None:
            ev.synchronize()  # the copy that last used this slot has left the host buffer
        buf[: window.size].copy_(torch.from_numpy(window.astype(np.int32)))
        dev = buf[: window.size].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.slots[self.next][1] = ev
        self.next = (self.next + 1) % len(self.slots)
        return dev.long()


class KJJ0DataLoader:
    """Sequential batches from a list of ``.bin`` shards (single process)."""

    def __init__(self, file_paths: List[Union[str, Path]], batch_size: int, sequence_length: int,
                 device: Union[str, torch.device] = "cpu", verbose: bool = True):
        self.batch_size = int(batch_size)
        self.sequence_length = int(sequence_length)
        self.device = device
        self.verbose = verbose
        self.files = sorted(str(f) for f in file_paths)
        if not self.files:
            raise ValueError("Empty file list provided")
        if verbose:
            print(f"Found {len(self.files)} shard files")
        self.current_shard_idx = 0
        self.current_tokens: Optional[np.ndarray] = None
        self.current_position = 0
        self._h2d = _HostToDevice(device, self.batch_size * self.sequence_length + 1)
        self._start_batch = 0   # batches to pass over at the start of the next iteration (resume)
        self._yielded = 0       # batches handed out so far, including the ones passed over

    # -- position (checkpoint / resume) --------------------------------------------------
    def state_dict(self) -> dict:
        """Position of the stream: the number of batches consumed. The stream is deterministic,
        so this is all a resumed run needs to continue with the batch the interrupted one
        would have seen next."""
        return {"batches": self._yielded}

    def load_state_dict(self, state: dict) -> None:
        self._start_batch = int(state.get("batches", 0))

    # -- shard handling -----------------------------------------------------------------
    def _load_shard(self, filepath: str) -> np.ndarray:
        tokens = shards.open_shard(filepath)
        if self.verbose:
            print(f"Loading shard: {filepath}\n  Tokens in shard: {len(tokens):,}")
        return tokens

    def _advance_shard(self) -> bool:
        if self.current_shard_idx >= len(self.files):
            return False
        self.current_tokens = self._load_shard(self.files[self.current_shard_idx])
        self.current_shard_idx += 1
        self.current_position = 0
        return True

    def _sequences_left_in_shard(self) -> int:
        """Number of whole windows the current shard still yields (a window needs pos + T < len)."""
        if self.current_tokens is None:
            return 0
        T = self.sequence_length
        room = len(self.current_tokens) - 1 - self.current_position  # need pos + T <= len - 1
        return max(0, room // T) if room >= T else 0

    def _reset(self) -> None:
        self.current_shard_idx = 0
        self.current_tokens = None
        self.current_position = 0

    # -- iteration ----------------------------------------------------------------------
    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        self._reset()
        B, T = self.batch_size, self.sequence_length
        skip, self._start_batch = self._start_batch, 0
        self._yielded = 0
        while True:
            pieces = []  # (window array, number of sequences) per shard touched by this batch
            need = B
            while need > 0:
                avail = self._sequences_left_in_shard()
                if avail == 0:
                    if not self._advance_shard():
                        return  # data exhausted: a partial batch is dropped, like the reference
                    continue
                take = min(avail, need)
                pos = self.current_position
                if skip == 0:
                    pieces.append((np.asarray(self.current_tokens[pos: pos + take * T + 1]), take))
                self.current_position += take * T
                need -= take
            self._yielded += 1
            if skip > 0:  # resuming: only the cursor moves (the shards are memory-mapped, nothing is read)
                skip -= 1
                continue
            if len(pieces) == 1:
                buf = self._h2d.put(pieces[0][0])
                yield buf[:-1].view(B, T), buf[1:].view(B, T)
            else:  # batch straddles a shard boundary
                ins, tgts = [], []
                for window, take in pieces:
                    buf = self._h2d.put(window)
                    ins.append(buf[:-1].view(take, T))
                    tgts.append(buf[1:].view(take, T))
                yield torch.cat(ins), torch.cat(tgts)

    # -- introspection ------------------------------------------------------------------
    def get_total_tokens(self) -> int:
        return sum(shards.read_header(f) for f in self.files)

    def get_info(self) -> dict:
        return {"num_shards": len(self.files), "batch_size": self.batch_size,
                "sequence_length": self.sequence_length, "device": self.device, "files": self.files,
                "total_tokens": self.get_total_tokens()}


class SyntheticTokenLoader:
    """Endless (or ``num_batches``-long) stream of uniform random token batches generated
    on the target device — the data source of the benchmarks (no host work, no PCIe)."""

    def __init__(self, batch_size: int, sequence_length: int, vocab_size: int = 50257,
                 device: Union[str, torch.device] = "cpu", num_batches: Optional[int] = None, seed: int = 1234):
        self.batch_size, self.sequence_length, self.vocab_size = batch_size, sequence_length, vocab_size
        self.device = torch.device(device)
        self.num_batches = num_batches
        self.seed = seed
        self._start_batch = 0
        self._yielded = 0

    def state_dict(self) -> dict:
        return {"batches": self._yielded}

    def load_state_dict(self, state: dict) -> None:
        self._start_batch = int(state.get("batches", 0))

    def __iter__(self):
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self.seed)
        skip, self._start_batch = self._start_batch, 0
        n = 0
        while self.num_batches is None or n < self.num_batches:
            buf = torch.randint(0, self.vocab_size, (self.batch_size, self.sequence_length + 1),
                                device=self.device, generator=gen)
            n += 1
            self._yielded = n
            if n <= skip:  # resuming: the generator is advanced past the batches already consumed
                continue
            yield buf[:, :-1].contiguous(), buf[:, 1:].contiguous()
