"""Chrome-trace analytics: temporal breakdown, communication/computation overlap, op diff.

Stands in for the Holistic Trace Analysis calls of the reference's notebook
(assignments/assignment1/analyze_traces.ipynb: ``get_temporal_breakdown`` cells 5-6,
``get_comm_comp_overlap`` cell 8, ``TraceDiff.compare_traces`` / ``ops_diff`` cells 11-19) on
the per-rank ``rank{r}_trace.json`` files the train scripts write. Plain Python over the
JSON: no pandas, no HTA.

Definitions (per rank, over the GPU kernel / memcpy events of the trace):
  compute      time covered by at least one non-communication kernel
  non-compute  time covered only by communication kernels or memory copies
  idle         trace span covered by nothing
  overlap      share of communication-kernel time that runs concurrently with compute
Communication kernels are recognised by name (nccl / rccl / allreduce / allgather /
reduce_scatter / broadcast ...), as in cell 13 of the notebook.
"""
from __future__ import annotations

import glob
import gzip
import json
import os
import re
from collections import defaultdict
from typing import Dict, Iterable, List, Optional, Tuple

COMM_PATTERN = re.compile(r"nccl|rccl|all_?reduce|all_?gather|reduce_?scatter|broadcast|sendrecv|onerankreduce", re.I)
GPU_CATS = {"kernel", "gpu_memcpy", "gpu_memset"}
MEM_CATS = {"gpu_memcpy", "gpu_memset"}


def load_trace(path: str) -> List[dict]:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        data = json.load(f)
    return data["traceEvents"] if isinstance(data, dict) else data


def gpu_events(events: Iterable[dict]) -> List[dict]:
    return [e for e in events if e.get("ph") == "X" and str(e.get("cat", "")).lower() in GPU_CATS]


def is_comm(event: dict) -> bool:
    return bool(COMM_PATTERN.search(event.get("name", "")))


def _union(intervals: List[Tuple[float, float]]) -> List[Tuple[float, float]]:
    out: List[Tuple[float, float]] = []
    for s, e in sorted(intervals):
        if out and s <= out[-1][1]:
            if e > out[-1][1]:
                out[-1] = (out[-1][0], e)
        else:
            out.append((s, e))
    return out


def _length(intervals: List[Tuple[float, float]]) -> float:
    return sum(e - s for s, e in intervals)


def _intersect(a: List[Tuple[float, float]], b: List[Tuple[float, float]]) -> List[Tuple[float, float]]:
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append((s, e))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def step_window(events: Iterable[dict], step: Optional[int]) -> Optional[Tuple[float, float]]:
    """(start, end) of the ``ProfilerStep#step`` annotation, or None."""
    if step is None:
        return None
    for e in events:
        if e.get("ph") == "X" and e.get("name") == f"ProfilerStep#{step}":
            return e["ts"], e["ts"] + e.get("dur", 0.0)
    return None


def _clip(evts: List[dict], window: Optional[Tuple[float, float]]) -> List[dict]:
    if window is None:
        return evts
    s, e = window
    return [x for x in evts if x["ts"] + x.get("dur", 0) > s and x["ts"] < e]


def temporal_breakdown(events: List[dict], step: Optional[int] = None) -> Dict[str, float]:
    g = _clip(gpu_events(events), step_window(events, step))
    if not g:
        return {"compute_us": 0.0, "non_compute_us": 0.0, "idle_us": 0.0, "span_us": 0.0,
                "compute_pct": 0.0, "non_compute_pct": 0.0, "idle_pct": 0.0}
    comp = _union([(e["ts"], e["ts"] + e["dur"]) for e in g
                   if not is_comm(e) and str(e.get("cat", "")).lower() not in MEM_CATS])
    busy = _union([(e["ts"], e["ts"] + e["dur"]) for e in g])
    span = max(e["ts"] + e["dur"] for e in g) - min(e["ts"] for e in g)
    compute, busy_t = _length(comp), _length(busy)
    res = {"compute_us": compute, "non_compute_us": busy_t - compute, "idle_us": span - busy_t, "span_us": span}
    for k in ("compute", "non_compute", "idle"):
        res[f"{k}_pct"] = 100.0 * res[f"{k}_us"] / span if span else 0.0
    return res


def comm_comp_overlap(events: List[dict], step: Optional[int] = None) -> Dict[str, float]:
    g = _clip(gpu_events(events), step_window(events, step))
    comm = _union([(e["ts"], e["ts"] + e["dur"]) for e in g if is_comm(e)])
    comp = _union([(e["ts"], e["ts"] + e["dur"]) for e in g
                   if not is_comm(e) and str(e.get("cat", "")).lower() not in MEM_CATS])
    comm_t = _length(comm)
    ov = _length(_intersect(comm, comp))
    return {"comm_us": comm_t, "overlapped_us": ov, "exposed_comm_us": comm_t - ov,
            "overlap_pct": 100.0 * ov / comm_t if comm_t else 0.0}


def kernel_summary(events: List[dict], step: Optional[int] = None) -> Dict[str, Tuple[int, float]]:
    """name -> (count, total microseconds) of GPU kernels (optionally inside one profiler step)."""
    out: Dict[str, List[float]] = defaultdict(lambda: [0, 0.0])
    for e in _clip(gpu_events(events), step_window(events, step)):
        rec = out[e.get("name", "?")]
        rec[0] += 1
        rec[1] += e.get("dur", 0.0)
    return {k: (int(v[0]), v[1]) for k, v in out.items()}


def ops_diff(control: List[dict], test: List[dict], step: Optional[int] = None) -> Dict[str, list]:
    """Kernels added / deleted / changed in count between two traces (``test`` vs ``control``)."""
    a, b = kernel_summary(control, step), kernel_summary(test, step)
    return {
        "added": sorted((n, c, t) for n, (c, t) in b.items() if n not in a),
        "deleted": sorted((n, c, t) for n, (c, t) in a.items() if n not in b),
        "increased": sorted((n, a[n][0], c) for n, (c, _) in b.items() if n in a and c > a[n][0]),
        "decreased": sorted((n, a[n][0], c) for n, (c, _) in b.items() if n in a and c < a[n][0]),
    }


def comm_ops(diff: Dict[str, list]) -> List[tuple]:
    """The added communication kernels of an ``ops_diff`` (the notebook's cell-13 filter)."""
    return [x for x in diff["added"] if COMM_PATTERN.search(x[0])]


def analyze_dir(trace_dir: str, step: Optional[int] = None) -> Dict[int, dict]:
    """Per-rank breakdown + overlap for every ``*trace*.json`` in a directory."""
    results = {}
    paths = sorted(glob.glob(os.path.join(trace_dir, "*trace*.json")) +
                   glob.glob(os.path.join(trace_dir, "*trace*.json.gz")))
    for path in paths:
        m = re.search(r"rank(\d+)", os.path.basename(path))
        rank = int(m.group(1)) if m else 0
        ev = load_trace(path)
        results[rank] = {"path": path, **temporal_breakdown(ev, step), **comm_comp_overlap(ev, step)}
    return results
