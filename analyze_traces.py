"""Trace analysis of the baseline / DDP / FSDP runs (the reference's analyze_traces.ipynb as a
script): temporal breakdown and communication/computation overlap per configuration and
rank, and kernel-level diffs Baseline->DDP, DDP->FSDP_FULL, DDP->FSDP_GRAD, FULL->GRAD at
rank 0 inside one profiler step (``--step``, default 4 like the notebook's "iteration 4").

usage: python analyze_traces.py [--traces outputs/traces] [--step 4] [--plots <dir>]
       python analyze_traces.py --from-rocprof <rocprofv3 output dir> --config ddp    (build the traces first)

The traces are the ones the training scripts write by themselves (kernel rows collected in
process); ``--plots <dir>`` also draws the notebook's figures: one pie of compute / non-compute /
idle time per configuration (cells 5-6) and the communication / computation overlap per
configuration and rank (cell 8), as PNG files.

``--from-rocprof`` converts a ``rocprofv3 --kernel-trace --marker-trace`` run of a train script into
``<traces>/<config>/rank{r}_trace.json`` (kernel rows per HIP stream, collectives on their own row,
the framework's named ranges) and then analyses as usual: no Kineto involved.
"""
import argparse
import os

from pytorch_distributed_amd.utils import trace_analysis as ta

CONFIGS = ["baseline", "ddp", "fsdp_full_shard", "fsdp_shard_grad_op", "fsdp_no_shard"]


def _rank0_trace(trace_root, cfg):
    for name in ("rank0_trace.json", "trace.json", "rank0_trace.json.gz", "trace.json.gz"):
        p = os.path.join(trace_root, cfg, name)
        if os.path.exists(p):
            return p
    return None


def draw_plots(per_config, out_dir):
    """The notebook's figures from ``{config: {rank: analysis row}}``: returns the files written."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    os.makedirs(out_dir, exist_ok=True)
    written = []
    configs = [c for c in CONFIGS if per_config.get(c)]
    if not configs:
        return written
    # temporal breakdown: one pie per configuration (mean over its ranks)
    fig, axes = plt.subplots(1, len(configs), figsize=(4.2 * len(configs), 4.2), squeeze=False)
    for ax, cfg in zip(axes[0], configs):
        rows = list(per_config[cfg].values())
        parts = [sum(r[k] for r in rows) / len(rows) for k in ("compute_pct", "non_compute_pct", "idle_pct")]
        wedges, *_ = ax.pie([max(p, 0.0) for p in parts], autopct=lambda v: f"{v:.1f}%" if v >= 4 else "",
                            colors=["#4c72b0", "#dd8452", "#c7c7c7"], startangle=90)
        ax.legend(wedges, [f"{n} {v:.1f}%" for n, v in zip(("compute", "non-compute", "idle"), parts)],
                  loc="lower center", bbox_to_anchor=(0.5, -0.18), fontsize=8, ncol=1)
        ax.set_title(f"{cfg} ({len(rows)} rank{'s' if len(rows) != 1 else ''})")
    fig.suptitle("Temporal breakdown of GPU time")
    fig.tight_layout()
    path = os.path.join(out_dir, "temporal_breakdown.png")
    fig.savefig(path, dpi=110)
    plt.close(fig)
    written.append(path)
    # communication / computation overlap per configuration and rank
    with_comm = [c for c in configs if any(r["comm_us"] > 0 for r in per_config[c].values())]
    if with_comm:
        fig, ax = plt.subplots(figsize=(1.6 + 1.4 * sum(len(per_config[c]) for c in with_comm), 4.0))
        x, labels, vals = 0, [], []
        for cfg in with_comm:
            for rank, r in sorted(per_config[cfg].items()):
                ax.bar(x, r["overlap_pct"], color="#4c72b0")
                ax.text(x, r["overlap_pct"] + 1, f"{r['overlap_pct']:.0f}%", ha="center", fontsize=8)
                labels.append(f"{cfg}\nrank {rank}")
                vals.append(x)
                x += 1
        ax.set_xticks(vals)
        ax.set_xticklabels(labels, fontsize=8)
        ax.set_ylim(0, 110)
        ax.set_ylabel("communication time overlapped with computation (%)")
        ax.set_title("Communication / computation overlap")
        fig.tight_layout()
        path = os.path.join(out_dir, "comm_comp_overlap.png")
        fig.savefig(path, dpi=110)
        plt.close(fig)
        written.append(path)
    return written


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--traces", default="outputs/traces")
    ap.add_argument("--step", type=int, default=4)
    ap.add_argument("--top", type=int, default=15)
    ap.add_argument("--from-rocprof", default=None, help="rocprofv3 output directory to convert first")
    ap.add_argument("--config", default="baseline", help="configuration name the converted traces are filed under")
    ap.add_argument("--active-only", action="store_true", help="keep only what ran inside the ProfilerStep windows")
    ap.add_argument("--plots", default=None, help="directory for the temporal-breakdown pies and the overlap chart (PNG)")
    args = ap.parse_args()
    if args.from_rocprof:
        from pytorch_distributed_amd.utils import rocprof_trace

        written = rocprof_trace.merge(args.from_rocprof, os.path.join(args.traces, args.config), args.active_only)
        for rank, path in sorted(written.items()):
            print(f"wrote {path}")
        if args.config == "baseline" and 0 in written:  # the reference's baseline trace is a single trace.json
            os.replace(written[0], os.path.join(args.traces, args.config, "trace.json"))

    print(f"{'config':<22}{'rank':>5}{'compute%':>10}{'non-comp%':>11}{'idle%':>8}{'comm ms':>10}{'overlap%':>10}")
    loaded, per_config = {}, {}
    for cfg in CONFIGS:
        d = os.path.join(args.traces, cfg)
        if not os.path.isdir(d):
            continue
        per_config[cfg] = ta.analyze_dir(d)
        for rank, r in sorted(per_config[cfg].items()):
            print(f"{cfg:<22}{rank:>5}{r['compute_pct']:>10.1f}{r['non_compute_pct']:>11.1f}"
                  f"{r['idle_pct']:>8.1f}{r['comm_us'] / 1e3:>10.2f}{r['overlap_pct']:>10.1f}")
        p = _rank0_trace(args.traces, cfg)
        if p:
            loaded[cfg] = ta.load_trace(p)
    if not loaded:
        print(f"no traces under {args.traces}; run train_baseline.py / train_ddp.py / train_fsdp.py first")
        return

    if args.plots:
        for path in draw_plots(per_config, args.plots):
            print(f"wrote {path}")

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
