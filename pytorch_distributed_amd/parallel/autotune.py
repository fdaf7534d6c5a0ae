"""Start-up communication policy: measure the fabric, then size DDP buckets and FSDP units.

Eight MI355X are fully connected by point-to-point xGMI links (7 per GPU), not by a switch:
the collective cost curve of THIS node — latency per call, bandwidth once a message is large
enough, whether reduce-scatter + all-gather beats a fused all-reduce — decides how gradients
should be bucketed and how many transformer blocks one sharding unit should hold. Defaults
copied from switch-based systems (25 MiB buckets, one block per unit) sit on the latency-bound
part of that curve. So the policy is measured, not assumed:

1. ``measure()`` times all-reduce, reduce-scatter + all-gather and all-gather on the native
   communicator for a ladder of message sizes (a few hundred milliseconds in total). The
   timings are max-reduced over ranks, so every rank derives the same policy.
2. ``fit()`` reduces each curve to a latency/bandwidth pair (least squares on t = a + S/b).
3. ``choose_policy()`` picks
   * the DDP bucket size: the smallest message that reaches ``efficiency`` of the asymptotic
     bandwidth (S >= eff/(1-eff) * a * b), capped so a backward still has >= ``min_buckets``
     buckets to overlap, and whether the two-phase reduction is faster at that size;
   * the FSDP unit size the same way on the all-gather curve, converted to whole transformer
     blocks per unit and capped so >= ``min_units`` units remain for prefetch overlap.

4. ``measure_gather_behind_reduce()`` — when the RCCL communicator owns a second library
   communicator for its all-gather queue — times the pattern of a sharded backward: a large
   reduce-scatter is issued, then the (small) all-gather prefetch of the next unit. On ONE library
   communicator the gather runs after the reduction; on two they run concurrently and share the
   links. Whichever finishes the GATHER sooner without slowing the pair down is kept
   (``policy.dual_route``), and the communicator is switched accordingly.

Explicit flags (``--bucket-mb``, ``--fsdp-unit-blocks``, ``--ddp-decompose``) always override;
the chosen values are logged by rank 0 at start-up.

RCCL settings. The library's own defaults for this topology are kept (its tuned channel
counts and protocol thresholds depend on the message size and are compiled in per gfx
target); ``rccl_environment()`` reports every ``NCCL_*`` / ``RCCL_*`` variable in effect so a run
is reproducible, and nothing is overridden silently.

Counterpart in the reference: none — torch's DDP bucket_cap_mb=25 and per-block FSDP wrapping
are used as they come (assignments/assignment1/train_ddp.py:46-49, train_fsdp.py:71-81).
"""
from __future__ import annotations

import dataclasses
import math
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch

MB = 2**20
DEFAULT_LADDER_MB = (1, 4, 16, 64, 256)


@dataclasses.dataclass
class CommPolicy:
    ddp_bucket_mb: float = 64.0
    ddp_decompose: bool = False
    fsdp_unit_mb: float = 64.0          # target unsharded bytes of one unit (bf16 parameters)
    dual_route: Optional[bool] = None   # all-gathers on a second library communicator (None: there is none)
    gather_behind_reduce_us: Dict[str, Tuple[float, float]] = dataclasses.field(default_factory=dict)
    source: str = "default"             # "default" | "measured" | "model"
    fits: Dict[str, Tuple[float, float]] = dataclasses.field(default_factory=dict)  # kind -> (latency s, bytes/s)
    measurements: Dict[str, List[Tuple[int, float]]] = dataclasses.field(default_factory=dict)

    def fsdp_unit_blocks(self, block_bytes: int, n_blocks: int, min_units: int = 6) -> int:
        """Whole transformer blocks per sharding unit for a model with ``n_blocks`` blocks of
        ``block_bytes`` each; at least ``min_units`` units are kept so prefetch has something to hide behind."""
        if block_bytes <= 0 or n_blocks <= 0:
            return 1
        k = max(1, math.ceil(self.fsdp_unit_mb * MB / block_bytes))
        k = min(k, max(1, n_blocks // min_units))
        return int(k)

    def describe(self) -> str:
        parts = [f"ddp_bucket_mb={self.ddp_bucket_mb:g}", f"ddp_decompose={self.ddp_decompose}",
                 f"fsdp_unit_mb={self.fsdp_unit_mb:g}", f"source={self.source}"]
        for kind, (a, b) in sorted(self.fits.items()):
            parts.append(f"{kind}: {a * 1e6:.0f}us + S/{b / 1e9:.0f}GB/s")
        if self.dual_route is not None:
            parts.append(f"dual_route={self.dual_route} " + " ".join(
                f"{k}: gather {g:.0f}us / pair {p:.0f}us" for k, (g, p) in sorted(self.gather_behind_reduce_us.items())))
        return ", ".join(parts)


_POLICY: Optional[CommPolicy] = None


def current_policy() -> Optional[CommPolicy]:
    return _POLICY


def set_policy(policy: Optional[CommPolicy]) -> None:
    global _POLICY
    _POLICY = policy


def fit(points: Sequence[Tuple[int, float]]) -> Tuple[float, float]:
    """Least-squares (latency seconds, bandwidth bytes/s) of t = a + S/b through ``(bytes, seconds)`` points."""
    pts = [(float(s), float(t)) for s, t in points if t > 0]
    if not pts:
        return 0.0, float("inf")
    if len(pts) == 1:
        return 0.0, pts[0][0] / pts[0][1]
    n = len(pts)
    sx = sum(s for s, _ in pts)
    sy = sum(t for _, t in pts)
    sxx = sum(s * s for s, _ in pts)
    sxy = sum(s * t for s, t in pts)
    den = n * sxx - sx * sx
    slope = (n * sxy - sx * sy) / den if den else 0.0
    if slope <= 0:  # flat or noisy curve: everything is latency; treat the largest point as the bandwidth
        s, t = max(pts)
        return min(t for _, t in pts), s / t
    a = max((sy - slope * sx) / n, 0.0)
    return a, 1.0 / slope


def _size_for_efficiency(latency: float, bandwidth: float, efficiency: float) -> float:
    """Smallest message (bytes) whose transfer time is ``efficiency`` of its total time."""
    if not math.isfinite(bandwidth):
        return 0.0
    return efficiency / (1.0 - efficiency) * latency * bandwidth


def _predict(fit_: Tuple[float, float], size: float) -> float:
    a, b = fit_
    return a + (size / b if math.isfinite(b) else 0.0)


def choose_policy(measurements: Dict[str, List[Tuple[int, float]]], total_grad_bytes: int = 0,
                  efficiency: float = 0.9, min_buckets: int = 6, bucket_range_mb=(16.0, 512.0),
                  unit_range_mb=(16.0, 1024.0), source: str = "measured") -> CommPolicy:
    """Derives the policy from ``{"all_reduce" | "rs_ag" | "all_gather": [(bytes, seconds), ...]}``."""
    fits = {k: fit(v) for k, v in measurements.items() if v}
    policy = CommPolicy(source=source, fits=fits, measurements={k: list(v) for k, v in measurements.items()})
    ar, two = fits.get("all_reduce"), fits.get("rs_ag")
    best = ar or two
    if best is not None:
        lo, hi = bucket_range_mb
        want = _size_for_efficiency(best[0], best[1], efficiency) / MB
        if total_grad_bytes > 0:
            hi = max(lo, min(hi, total_grad_bytes / MB / min_buckets))
        policy.ddp_bucket_mb = float(min(max(want, lo), hi))
        if ar is not None and two is not None:
            size = policy.ddp_bucket_mb * MB
            policy.ddp_decompose = _predict(two, size) < 0.95 * _predict(ar, size)
            if policy.ddp_decompose:
                want = _size_for_efficiency(two[0], two[1], efficiency) / MB
                policy.ddp_bucket_mb = float(min(max(want, lo), hi))
    ag = fits.get("all_gather")
    if ag is not None:
        lo, hi = unit_range_mb
        policy.fsdp_unit_mb = float(min(max(_size_for_efficiency(ag[0], ag[1], efficiency) / MB, lo), hi))
    return policy


def measure(communicator, device, ladder_mb: Sequence[float] = DEFAULT_LADDER_MB, iters: int = 4,
            dtype=torch.bfloat16) -> Dict[str, List[Tuple[int, float]]]:
    """Times the bulk collectives on ``communicator`` (every rank must call this). Returns
    ``{kind: [(message bytes, seconds per call)]}`` with the slowest rank's time per point."""
    world = communicator.world_size
    on_gpu = communicator.on_gpu
    out: Dict[str, List[Tuple[int, float]]] = {"all_reduce": [], "rs_ag": [], "all_gather": []}

    def sync():
        if on_gpu:
            torch.cuda.synchronize(device)

    def timed(fn) -> float:
        fn()  # warm-up (first call of a size sets up RCCL's channels / buffers)
        sync()
        communicator.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        sync()
        return (time.perf_counter() - t0) / iters

    es = torch.empty((), dtype=dtype).element_size()
    rows = []
    for mb in ladder_mb:
        n = int(mb * MB) // es
        n = max(world * 64, n // (world * 64) * (world * 64))
        buf = torch.zeros(n, dtype=dtype, device=device)
        chunk = n // world
        mine = buf.narrow(0, communicator.rank * chunk, chunk)

        def ar():
            communicator.all_reduce(buf, "avg", 1).wait()

        def two():
            communicator.reduce_scatter(mine, buf, "avg", 1).wait()
            communicator.all_gather(buf, mine, 1).wait()

        def ag():
            communicator.all_gather(buf, mine, 0).wait()

        rows.append((n * es, timed(ar), timed(two), timed(ag)))
        del buf
    # slowest rank per point, identical on every rank
    t = torch.tensor([[r[1], r[2], r[3]] for r in rows], dtype=torch.float64, device=device)
    if world > 1:
        if t.is_cuda:  # float64 MAX over RCCL
            communicator.all_reduce(t, "max", 1).wait()
            sync()
        else:
            communicator.all_reduce(t, "max", 1).wait()
    t = t.cpu().tolist()
    for (nbytes, *_), (a, b, c) in zip(rows, t):
        out["all_reduce"].append((nbytes, a))
        out["rs_ag"].append((nbytes, b))
        out["all_gather"].append((nbytes, c))
    return out


def measure_gather_behind_reduce(communicator, device, reduce_mb: float = 64.0, gather_mb: float = 16.0,
                                 iters: int = 4, dtype=torch.bfloat16) -> Dict[str, Tuple[float, float]]:
    """For a communicator with a second library communicator: ``{"shared" | "dual": (microseconds
    until the all-gather issued right behind a reduce-scatter has completed, microseconds until both
    have)}``, slowest rank. Collective; leaves the route as it found it."""
    world = communicator.world_size
    es = torch.empty((), dtype=dtype).element_size()

    def sized(mb):
        n = int(mb * MB) // es
        return max(world * 64, n // (world * 64) * (world * 64))

    big = torch.zeros(sized(reduce_mb), dtype=dtype, device=device)
    big_mine = big.narrow(0, communicator.rank * (big.numel() // world), big.numel() // world)
    small = torch.zeros(sized(gather_mb), dtype=dtype, device=device)
    small_mine = small.narrow(0, communicator.rank * (small.numel() // world), small.numel() // world)
    was = communicator.dual_route()
    out = {}
    try:
        for name, route in (("shared", False), ("dual", True)):
            communicator.set_dual_route(route)
            t_gather = t_pair = 0.0
            for it in range(iters + 1):  # first round warms the route up
                torch.cuda.synchronize(device)
                communicator.barrier()
                t0 = time.perf_counter()
                rs = communicator.reduce_scatter(big_mine, big, "avg", 1)
                ag = communicator.all_gather(small, small_mine, 0)
                ag.synchronize()
                t1 = time.perf_counter()
                rs.synchronize()
                t2 = time.perf_counter()
                if it:
                    t_gather += (t1 - t0) / iters
                    t_pair += (max(t1, t2) - t0) / iters
            t = torch.tensor([t_gather, t_pair], dtype=torch.float64, device=device)
            if world > 1:
                communicator.all_reduce(t, "max", 1).wait()
                torch.cuda.synchronize(device)
            g, p = t.cpu().tolist()
            out[name] = (g * 1e6, p * 1e6)
    finally:
        communicator.set_dual_route(was)
    return out


def rccl_environment() -> Dict[str, str]:
    """Every RCCL / NCCL tuning variable visible to this process (logged with the policy)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))}


def tune(communicator=None, device=None, total_grad_bytes: int = 0, verbose: bool = True,
         ladder_mb: Sequence[float] = DEFAULT_LADDER_MB) -> CommPolicy:
    """Measures the fabric and installs the resulting policy process-wide. Collective: every rank
    must call it at the same point (before the model is wrapped)."""
    from . import comm

    c = communicator if communicator is not None else comm.get_communicator()
    if c is None or c.world_size <= 1:
        policy = CommPolicy(source="default")
        set_policy(policy)
        return policy
    device = device if device is not None else comm.comm_device()
    policy = choose_policy(measure(c, device, ladder_mb), total_grad_bytes)
    if getattr(c, "has_dual", None) is not None and c.has_dual():
        try:
            policy.gather_behind_reduce_us = measure_gather_behind_reduce(c, device)
            (g_s, p_s), (g_d, p_d) = policy.gather_behind_reduce_us["shared"], policy.gather_behind_reduce_us["dual"]
            policy.dual_route = bool(g_d < 0.9 * g_s and p_d < 1.05 * p_s)
        except Exception as exc:  # noqa: BLE001 - the shared route always works
            if c.rank == 0:
                print(f"[pda.comm] gather-behind-reduce measurement failed ({exc!r}); gathers stay on the shared "
                      "communicator")
            policy.dual_route = False
        c.set_dual_route(policy.dual_route)
    set_policy(policy)
    if verbose and c.rank == 0:
        print(f"[pda.comm] policy: {policy.describe()}")
        env = rccl_environment()
        if env:
            print("[pda.comm] environment: " + " ".join(f"{k}={v}" for k, v in env.items()))
    return policy
