#!/usr/bin/env python3
"""Scaling sweep: bench.py on 1/2/4/8 GPUs of one node for DDP and the three FSDP strategies.

Every point is one `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` child
(one process per GPU over RCCL); its JSON line is appended to `<out>/scaling_sweep.jsonl` together
with the weak-scaling efficiency, and a table is printed at the end.

Efficiency as reported here: (tokens/s at N GPUs / N) divided by (tokens/s / N0) of the SAME mode at
the smallest count N0 in the sweep, per-GPU batch fixed. With N0 = 1 the base point runs the wrapped
model only under `--wrap-single` (wrapper overhead and forced collectives included in the base, so
the ratio isolates the fabric); without it the base is the unwrapped single-GPU step (the ratio then
also contains the wrapper's cost). The end-of-round scaling figure is computed by the build driver
from its own 1/2/4/8 runs, not by this tool.

`--tune` additionally sweeps, at the largest GPU count, the knobs that the start-up autotuner
(`parallel/autotune.py`) otherwise sets from its latency / bandwidth fit — DDP bucket size, DDP
all-reduce vs reduce-scatter + all-gather, FSDP blocks per unit — and reports the best of each, as
a check of the autotuner's choice against measured step time.

  python tools/scaling_sweep.py                              # all visible GPUs, GPT-2 Large
  python tools/scaling_sweep.py --gpus 1 2 --modes ddp fsdp:FULL_SHARD --model gpt2-medium --tune
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_point(n, mode, strategy, args, extra=()):
    cmd = [sys.executable]
    if n > 1 or args.wrap_single:
        port = 29500 + (os.getpid() + int(time.time())) % 2000
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--model", args.model, "--seq-len", str(args.seq_len), "--batch-per-gpu", str(args.batch_per_gpu),
            "--micro-batch", str(args.micro_batch), "--mode", mode]
    if mode == "fsdp":
        cmd += ["--strategy", strategy]
    if n == 1 and args.wrap_single:
        cmd += ["--wrap-single"]
    cmd += list(extra)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              timeout=args.timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"timeout after {args.timeout}s", "cmd": " ".join(cmd)}
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        return {"error": f"exit {proc.returncode}", "stderr_tail": proc.stderr[-2000:], "cmd": " ".join(cmd)}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="*", default=None, help="GPU counts (default 1,2,4,8 up to the visible count)")
    ap.add_argument("--modes", nargs="*", default=["ddp", "fsdp:FULL_SHARD", "fsdp:SHARD_GRAD_OP", "fsdp:NO_SHARD"])
    ap.add_argument("--model", default="gpt2-large")
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--micro-batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--wrap-single", action="store_true", help="run the 1-GPU point through the wrapper with forced collectives")
    ap.add_argument("--tune", action="store_true", help="sweep DDP bucket MB and FSDP blocks per unit at the largest count")
    ap.add_argument("--out", default=os.path.join(ROOT, "outputs", "scaling"))
    args = ap.parse_args()

    import torch
    visible = torch.cuda.device_count()
    counts = args.gpus or [n for n in (1, 2, 4, 8) if n <= max(visible, 1)]
    os.makedirs(args.out, exist_ok=True)
    log = open(os.path.join(args.out, "scaling_sweep.jsonl"), "a")
    table = []
    for spec in args.modes:
        mode, _, strategy = spec.partition(":")
        strategy = strategy or "FULL_SHARD"
        base = None
        for n in counts:
            res = run_point(n, mode, strategy, args)
            res.update({"sweep_mode": spec, "sweep_gpus": n})
            if "error" not in res:
                per_gpu = res["value"] / n
                base = base or per_gpu
                res["scaling_efficiency"] = per_gpu / base
                table.append((spec, n, res["value"], res["ms_per_step"], res.get("mfu"), res["scaling_efficiency"]))
            else:
                table.append((spec, n, None, None, None, None))
                print(f"[sweep] {spec} N={n}: {res['error']}", file=sys.stderr)
            log.write(json.dumps(res) + "\n")
            log.flush()
    if args.tune:
        n = counts[-1]
        for flag, mode, strategy, values in (("--bucket-mb", "ddp", "", (16, 32, 64, 128, 256)),
                                             ("--ddp-decompose", "ddp", "", ("on", "off")),
                                             ("--fsdp-unit-blocks", "fsdp", "FULL_SHARD", (1, 2, 4, 6))):
            best = None
            for v in values:
                res = run_point(n, mode, strategy, args, extra=(flag, str(v)))
                res.update({"sweep_mode": f"tune {mode} {flag}={v}", "sweep_gpus": n})
                log.write(json.dumps(res) + "\n")
                log.flush()
                if "error" not in res:
                    print(f"[tune] N={n} {mode} {flag}={v}: {res['value']:.0f} tok/s ({res['ms_per_step']:.1f} ms)")
                    if best is None or res["value"] > best[1]:
                        best = (v, res["value"])
            if best:
                print(f"[tune] best {flag} for {mode} on {n} GPUs: {best[0]} ({best[1]:.0f} tok/s)")
    print(f"{'mode':22s} {'GPUs':>4s} {'tokens/s':>12s} {'ms/step':>9s} {'MFU':>7s} {'scaling':>8s}")
    for spec, n, v, ms, mfu, eff in table:
        if v is None:
            print(f"{spec:22s} {n:4d} {'failed':>12s}")
        else:
            print(f"{spec:22s} {n:4d} {v:12.0f} {ms:9.1f} {100 * (mfu or 0):6.1f}% {100 * eff:7.1f}%")


if __name__ == "__main__":
    main()
