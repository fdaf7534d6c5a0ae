#!/usr/bin/env python3
"""Hardware-counter mode of the profiler: per-kernel matrix-pipe utilisation, stall share, LDS bank
conflicts, L2 hit rate and HBM traffic of a training step (or of any script), from rocprofv3 counter
passes.

The reference has no counterpart: its profiling is the torch.profiler timeline only
(assignments/assignment1/train_baseline.py:79-87). On MI355X the questions a kernel author asks —
is the matrix pipe busy, what are the waves waiting for, does LDS conflict, how many bytes cross
the HBM interface — are answered by the SQ / TCC performance counters, which rocprofv3 samples per
kernel dispatch.

How it runs. The hardware has a fixed number of counter slots per block (8 SQ, 4 TCC, 2 GRBM per
pass, and rocprofv3 does not split a request over passes), so the counters are collected in
several passes — the target is simply run once per pass, each time as
``rocprofv3 --pmc <group> -- python3 <script> <args>`` with the program directly after ``--``, plus
one ``--kernel-trace`` pass for durations. The per-dispatch rows are then averaged per kernel and
turned into the derived figures below. Nothing in this process touches the GPU; every pass is a
child process under its own time limit, and the first failing pass ends the run.

  python counter_profile.py                               # one optimizer step of bench.py (GPT-2 large)
  python counter_profile.py --target tools/wgrad_pmc.py   # a single kernel's driver script
  python counter_profile.py --filter attn --out outputs/counters/attn -- --model gpt2 --steps 2 --warmup 1

``--window-kernel NAME`` restricts every pass to the dispatches between the last two bursts of a
marker kernel (default for bench.py: ``adamw``, i.e. exactly the last optimizer step — start-up work
such as the GEMM planner's candidate timing is left out).

Derived figures (per kernel, mean over its dispatches):
  mfma_busy    SQ_VALU_MFMA_BUSY_CYCLES / (duration x nominal clock x SIMDs of the device)   share of SIMD time
               the matrix pipe works (clock and SIMD count from rocprofv3's agent table; a chip that
               runs below its nominal clock under load is busier than this figure says)
  wait_share   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                       share of wave time spent in s_waitcnt
  valu_per_mfma  SQ_INSTS_VALU / SQ_INSTS_MFMA
  lds_conflict SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                share of LDS cycles lost to bank conflicts
  l2_hit       TCC_HIT_sum / TCC_REQ_sum
  hbm_read / hbm_write   FETCH_SIZE / WRITE_SIZE (KiB -> MiB per dispatch) and the implied GB/s. Measured
               caveat on MI355X: WRITE_SIZE matches what a streaming kernel must write, FETCH_SIZE (rocprofv3's
               derived metric) reads about half of what it must fetch (GELU forward: 160 MiB for a 320 MiB
               input) — compare read figures between kernels, do not take them as absolute.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))

# One entry per pass; each respects the per-block slot limits.
PASSES = {
    "sq": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
           "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "GRBM_GUI_ACTIVE"],
    "hbm_read": ["FETCH_SIZE"],
    "hbm_write": ["WRITE_SIZE"],
}


def short(name: str) -> str:
    name = name.replace("pda::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:72]


def run_pass(out_dir: str, what: str, counters, target: str, target_args, limit: int) -> bool:
    d = os.path.join(out_dir, what)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    cmd = ["timeout", "-s", "KILL", str(limit), "rocprofv3"]
    cmd += ["--kernel-trace"] if counters is None else ["--pmc", *counters]
    cmd += ["--output-format", "csv", "-d", d, "--", "python3", target, *target_args]
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(out_dir, what + ".log"), "w") as log:
        rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT).returncode
    if rc != 0:
        print(f"[counters] pass '{what}' failed (exit {rc}); see {os.path.join(out_dir, what + '.log')}", file=sys.stderr)
    return rc == 0


def window(rows, marker: str):
    """Dispatch-id interval (lo, hi] between the ends of the last two bursts of kernels whose name
    contains `marker` (a burst = marker dispatches less than 50 ids apart); everything when the
    marker is empty or occurs in fewer than two bursts."""
    if not marker:
        return None
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if marker in r["Kernel_Name"]})
    ends = [a for a, b in zip(ids, ids[1:] + [None]) if b is None or b - a >= 50]
    return (ends[-2], ends[-1]) if len(ends) >= 2 else None


def in_window(r, win) -> bool:
    return win is None or win[0] < int(r["Dispatch_Id"]) <= win[1]


def collect(out_dir: str, flt: str, marker: str = ""):
    """{kernel: {counter: mean value per dispatch}} plus mean duration and dispatch count from the trace pass."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        win = window(rows, marker)
        for r in rows:
            k = short(r["Kernel_Name"])
            if flt in k and in_window(r, win):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        win = window(rows, marker)
        for r in rows:
            k = short(r["Kernel_Name"])
            if flt in k and in_window(r, win):
                agg[k]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(cs.get("duration_us", []))}
            for k, cs in agg.items()}


def device_simd_hz(out_dir: str) -> float:
    """SIMDs x nominal clock of the profiled GPU from rocprofv3's agent table (MI355X values if absent)."""
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*_agent_info.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            low = {k.lower(): v for k, v in r.items()}
            if low.get("agent_type", "").upper() != "GPU":
                continue
            try:
                cus = int(low.get("cu_count") or 0)
                simds = int(low.get("simd_per_cu") or 4)
                mhz = float(low.get("max_engine_clk_fcompute") or 0)
                if cus > 0 and mhz > 0:
                    return cus * simds * mhz * 1e6
            except ValueError:
                pass
    return 256 * 4 * 2400e6


def derive(c: dict, simd_hz: float = 256 * 4 * 2400e6) -> dict:
    def ratio(a, b, scale=1.0):
        return scale * c[a] / c[b] if a in c and b in c and c[b] > 0 else None

    busy = None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("duration_us"):
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["duration_us"] * 1e-6 * simd_hz)
    d = {"mfma_busy": busy,
         "wait_share": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
         "valu_per_mfma": ratio("SQ_INSTS_VALU", "SQ_INSTS_MFMA"),
         "lds_conflict": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
         "l2_hit": ratio("TCC_HIT_sum", "TCC_REQ_sum")}
    dur = c.get("duration_us")
    for key, ctr in (("hbm_read", "FETCH_SIZE"), ("hbm_write", "WRITE_SIZE")):
        if ctr in c:
            d[key + "_mb"] = c[ctr] / 1024.0  # the counter is in KiB
            d[key + "_gbps"] = c[ctr] * 1024.0 / (dur * 1e-6) / 1e9 if dur else None
    return d


def fmt(v, pct=False, width=8):
    if v is None:
        return " " * (width - 1) + "-"
    return f"{100 * v:{width - 1}.1f}%" if pct else f"{v:{width}.1f}"


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--target", default=os.path.join(ROOT, "bench.py"), help="python script to profile")
    ap.add_argument("--out", default=os.path.join(ROOT, "outputs", "counters", "step"))
    ap.add_argument("--filter", default="", help="only kernels whose name contains this")
    ap.add_argument("--passes", nargs="*", default=list(PASSES), choices=list(PASSES))
    ap.add_argument("--limit", type=int, default=240, help="seconds per pass")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--window-kernel", default=None,
                    help="only dispatches between the last two bursts of this kernel (default: adamw for bench.py, none otherwise)")
    ap.add_argument("--keep-raw", action="store_true", help="keep rocprofv3's per-dispatch tables")
    ap.add_argument("target_args", nargs="*", help="arguments of the target (after --)")
    args = ap.parse_args()
    target = os.path.abspath(args.target)
    targs = args.target_args
    if not targs and os.path.basename(target) == "bench.py":
        targs = ["--steps", "2", "--warmup", "2", "--no-reference-recipe"]
    marker = args.window_kernel
    if marker is None:
        marker = "adamw" if os.path.basename(target) == "bench.py" else ""
    args.out = os.path.abspath(args.out)  # the passes run with /tmp as their working directory
    os.makedirs(args.out, exist_ok=True)
    if not run_pass(args.out, "trace", None, target, targs, args.limit):
        raise SystemExit(1)
    for what in args.passes:
        if not run_pass(args.out, what, PASSES[what], target, targs, args.limit):
            raise SystemExit(1)
    rows = collect(args.out, args.filter, marker)
    simd_hz = device_simd_hz(args.out)
    order = sorted(rows, key=lambda k: -rows[k].get("duration_us", 0.0) * max(rows[k]["dispatches"], 1))[:args.top]
    lines = [f"{'kernel':72s} {'n':>5s} {'us':>8s} {'mfma':>8s} {'wait':>8s} {'valu/mf':>8s} {'ldsconf':>8s} {'l2hit':>8s} "
             f"{'rd MiB':>8s} {'rd GB/s':>8s} {'wr MiB':>8s} {'wr GB/s':>8s}"]
    for k in order:
        c, d = rows[k], derive(rows[k], simd_hz)
        lines.append(f"{k:72s} {c['dispatches']:5d} {fmt(c.get('duration_us'))} {fmt(d['mfma_busy'], True)} "
                     f"{fmt(d['wait_share'], True)} {fmt(d['valu_per_mfma'])} {fmt(d['lds_conflict'], True)} "
                     f"{fmt(d['l2_hit'], True)} {fmt(d.get('hbm_read_mb'))} {fmt(d.get('hbm_read_gbps'))} "
                     f"{fmt(d.get('hbm_write_mb'))} {fmt(d.get('hbm_write_gbps'))}")
    report = "\n".join(lines)
    print(report)
    with open(os.path.join(args.out, "counters.txt"), "w") as f:
        f.write(report + "\n")
    for what in ([] if args.keep_raw else ["trace", *args.passes]):  # the raw tables are large; the report is kept
        shutil.rmtree(os.path.join(args.out, what), ignore_errors=True)


if __name__ == "__main__":
    main()
