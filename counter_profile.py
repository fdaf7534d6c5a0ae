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

Derived figures (per kernel, mean over its dispatches):
  mfma_busy    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES)   share of SIMD time the matrix pipe works
  wait_share   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                       share of wave time spent in s_waitcnt
  valu_per_mfma  SQ_INSTS_VALU / SQ_INSTS_MFMA
  lds_conflict SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                share of LDS cycles lost to bank conflicts
  l2_hit       TCC_HIT_sum / TCC_REQ_sum
  hbm_read / hbm_write   FETCH_SIZE / WRITE_SIZE (KiB -> MB per dispatch) and the implied GB/s
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


def collect(out_dir: str, flt: str):
    """{kernel: {counter: mean value per dispatch}} plus mean duration and dispatch count from the trace pass."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt in k:
                agg[k]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(cs.get("duration_us", []))}
            for k, cs in agg.items()}


def derive(c: dict) -> dict:
    def ratio(a, b, scale=1.0):
        return scale * c[a] / c[b] if a in c and b in c and c[b] > 0 else None

    d = {"mfma_busy": ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", 0.25),
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
    ap.add_argument("target_args", nargs="*", help="arguments of the target (after --)")
    args = ap.parse_args()
    target = os.path.abspath(args.target)
    targs = args.target_args
    if not targs and os.path.basename(target) == "bench.py":
        targs = ["--steps", "1", "--warmup", "2"]
    os.makedirs(args.out, exist_ok=True)
    if not run_pass(args.out, "trace", None, target, targs, args.limit):
        raise SystemExit(1)
    for what in args.passes:
        if not run_pass(args.out, what, PASSES[what], target, targs, args.limit):
            raise SystemExit(1)
    rows = collect(args.out, args.filter)
    order = sorted(rows, key=lambda k: -rows[k].get("duration_us", 0.0) * max(rows[k]["dispatches"], 1))[:args.top]
    lines = [f"{'kernel':72s} {'n':>5s} {'us':>8s} {'mfma':>8s} {'wait':>8s} {'valu/mf':>8s} {'ldsconf':>8s} {'l2hit':>8s} "
             f"{'rd MB':>8s} {'rd GB/s':>8s} {'wr MB':>8s} {'wr GB/s':>8s}"]
    for k in order:
        c, d = rows[k], derive(rows[k])
        lines.append(f"{k:72s} {c['dispatches']:5d} {fmt(c.get('duration_us'))} {fmt(d['mfma_busy'], True)} "
                     f"{fmt(d['wait_share'], True)} {fmt(d['valu_per_mfma'])} {fmt(d['lds_conflict'], True)} "
                     f"{fmt(d['l2_hit'], True)} {fmt(d.get('hbm_read_mb'))} {fmt(d.get('hbm_read_gbps'))} "
                     f"{fmt(d.get('hbm_write_mb'))} {fmt(d.get('hbm_write_gbps'))}")
    report = "\n".join(lines)
    print(report)
    with open(os.path.join(args.out, "counters.txt"), "w") as f:
        f.write(report + "\n")
    for what in ["trace", *args.passes]:  # the raw per-dispatch tables are large; the report is what is kept
        shutil.rmtree(os.path.join(args.out, what), ignore_errors=True)


if __name__ == "__main__":
    main()
