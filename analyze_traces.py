"""Trace analysis of the baseline / DDP / FSDP runs (the reference's analyze_traces.ipynb as a
script): temporal breakdown and communication/computation overlap per configuration and
rank, and kernel-level diffs Baseline->DDP, DDP->FSDP_FULL, DDP->FSDP_GRAD, FULL->GRAD at
rank 0 inside one profiler step (``--step``, default 4 like the notebook's "iteration 4").

usage: python analyze_traces.py [--traces outputs/traces] [--step 4]
       python analyze_traces.py --from-rocprof <rocprofv3 output dir> --config ddp    (build the traces first)

``--from-rocprof`` converts a ``rocprofv3 --kernel-trace --marker-trace`` run of a train script into
``<traces>/<config>/rank{r}_trace.json`` (kernel rows per HIP stream, collectives on their own row,
the framework's named ranges) and then analyses as usual: no Kineto involved.
"""
import argparse
import os

from pytorch_distributed_amd.utils import trace_analysis as ta

CONFIGS = ["baseline", "ddp", "fsdp_full_shard", "fsdp_shard_grad_op", "fsdp_no_shard"]


def _rank0_trace(trace_root, cfg):
    for name in ("rank0_trace.json", "trace.json"):
        p = os.path.join(trace_root, cfg, name)
        if os.path.exists(p):
            return p
    return None


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--traces", default="outputs/traces")
    ap.add_argument("--step", type=int, default=4)
    ap.add_argument("--top", type=int, default=15)
    ap.add_argument("--from-rocprof", default=None, help="rocprofv3 output directory to convert first")
    ap.add_argument("--config", default="baseline", help="configuration name the converted traces are filed under")
    ap.add_argument("--active-only", action="store_true", help="keep only what ran inside the ProfilerStep windows")
    args = ap.parse_args()
    if args.from_rocprof:
        from pytorch_distributed_amd.utils import rocprof_trace

        written = rocprof_trace.merge(args.from_rocprof, os.path.join(args.traces, args.config), args.active_only)
        for rank, path in sorted(written.items()):
            print(f"wrote {path}")
        if args.config == "baseline" and 0 in written:  # the reference's baseline trace is a single trace.json
            os.replace(written[0], os.path.join(args.traces, args.config, "trace.json"))

    print(f"{'config':<22}{'rank':>5}{'compute%':>10}{'non-comp%':>11}{'idle%':>8}{'comm ms':>10}{'overlap%':>10}")
    loaded = {}
    for cfg in CONFIGS:
        d = os.path.join(args.traces, cfg)
        if not os.path.isdir(d):
            continue
        for rank, r in sorted(ta.analyze_dir(d).items()):
            print(f"{cfg:<22}{rank:>5}{r['compute_pct']:>10.1f}{r['non_compute_pct']:>11.1f}"
                  f"{r['idle_pct']:>8.1f}{r['comm_us'] / 1e3:>10.2f}{r['overlap_pct']:>10.1f}")
        p = _rank0_trace(args.traces, cfg)
        if p:
            loaded[cfg] = ta.load_trace(p)
    if not loaded:
        print(f"no traces under {args.traces}; run train_baseline.py / train_ddp.py / train_fsdp.py first")
        return

    pairs = [("baseline", "ddp"), ("ddp", "fsdp_full_shard"), ("ddp", "fsdp_shard_grad_op"),
             ("fsdp_full_shard", "fsdp_shard_grad_op")]
    for a, b in pairs:
        if a not in loaded or b not in loaded:
            continue
        diff = ta.ops_diff(loaded[a], loaded[b], step=args.step)
        print(f"\n=== {a} -> {b} (rank 0, ProfilerStep#{args.step}) ===")
        print(f"added kernels: {len(diff['added'])}, deleted: {len(diff['deleted'])}, "
              f"count increased: {len(diff['increased'])}, decreased: {len(diff['decreased'])}")
        comm = ta.comm_ops(diff)
        if comm:
            print("added communication kernels:")
            for name, count, total in comm[: args.top]:
                print(f"  {count:5d} x {total / max(count, 1):9.1f} us  {name[:100]}")


if __name__ == "__main__":
    main()
