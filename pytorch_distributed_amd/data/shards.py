"""The ``.bin`` token-shard format of the ``kjj0/fineweb10B-gpt2`` dataset.

Layout (as read by the reference, assignments/assignment1/data/data_loader.py:70-75,
113-135): a header of 256 little-endian int32 values — ``[0]`` magic ``20240520``,
``[1]`` version ``1``, ``[2]`` token count — followed by the tokens as ``uint16``.

Shards are memory-mapped, never read whole: a 100 M-token shard costs page-cache
pages only for the windows actually consumed.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Union

import numpy as np

MAGIC = 20240520
VERSION = 1
HEADER_INTS = 256
HEADER_BYTES = HEADER_INTS * 4


def read_header(path: Union[str, Path]) -> int:
    """Validates the header and returns the token count."""
    with open(path, "rb") as f:
        raw = f.read(HEADER_BYTES)
    if len(raw) != HEADER_BYTES:
        raise ValueError(f"{path}: truncated header ({len(raw)} bytes)")
    header = np.frombuffer(raw, dtype="<i4")
    if int(header[0]) != MAGIC:
        raise ValueError(f"{path}: invalid magic number {int(header[0])}, expected {MAGIC}")
    if int(header[1]) != VERSION:
        raise ValueError(f"{path}: unsupported version {int(header[1])}, expected {VERSION}")
    return int(header[2])


def open_shard(path: Union[str, Path]) -> np.ndarray:
    """Memory-maps the token payload of a shard as a read-only uint16 array."""
    count = read_header(path)
    have = (os.path.getsize(path) - HEADER_BYTES) // 2
    if have < count:
        raise ValueError(f"{path}: token count mismatch: header says {count}, file holds {have}")
    return np.memmap(path, dtype="<u2", mode="r", offset=HEADER_BYTES, shape=(count,))


def write_shard(path: Union[str, Path], tokens) -> None:
    """Writes ``tokens`` (any integer sequence < 65536) as a shard file."""
    arr = np.asarray(tokens)
    if arr.size and (arr.min() < 0 or arr.max() > 0xFFFF):
        raise ValueError("tokens do not fit uint16")
    header = np.zeros(HEADER_INTS, dtype="<i4")
    header[0], header[1], header[2] = MAGIC, VERSION, arr.size
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_suffix(path.suffix + f".tmp{os.getpid()}")
    with open(tmp, "wb") as f:
        f.write(header.tobytes())
        f.write(arr.astype("<u2").tobytes())
    os.replace(tmp, path)  # atomic: concurrent ranks never observe a half-written shard
