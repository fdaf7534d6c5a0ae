"""Per-optimizer-step kernel time by category from a rocprofv3 kernel trace of bench.py.

usage: python tools/step_breakdown.py <kernel_trace.csv> [steps] [micro_batches_per_step]
Micro-batches are delimited by the cross-entropy kernel (one launch per micro-batch; one micro-batch
per step in the default bench, four in the reference's recipe); the last `steps` optimizer steps are
averaged, so planner tuning and warm-up launches earlier in the trace do not count.
"""
import collections
import csv
import sys


def category(n: str) -> str:
    table = (("wgrad_kernel", "weight-gradient GEMM"), ("wgrad_reduce", "weight-gradient slice reduction"),
             ("attn_fwd", "attention forward"), ("attn_bwd_dq", "attention dQ"), ("attn_bwd_dkv", "attention dK/dV"),
             ("attn_bwd_delta", "attention delta"), ("gelu", "GELU"), ("norm", "LayerNorm / RMSNorm"),
             ("adamw", "AdamW"), ("::ce_kernel", "cross-entropy"), ("rope", "rotary"), ("swiglu", "SwiGLU"),
             ("embed", "embedding"), ("Cijk", "library GEMM"))
    for key, name in table:
        if key in n:
            return name
    return "other: " + n[:60]


def main():
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "::ce_kernel" in r["Kernel_Name"]]
    lo, hi = marks[-steps * per - 1], marks[-1]
    sel = rows[lo:hi]
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / steps
    tot, cnt = collections.Counter(), collections.Counter()
    for r in sel:
        c = category(r["Kernel_Name"])
        tot[c] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / steps
        cnt[c] += 1
    print(f"per step over the last {steps} steps: wall {wall:.2f} ms, kernel time {sum(tot.values()):.2f} ms")
    for k, v in tot.most_common():
        print(f"{v:8.2f} ms {cnt[k] / steps:7.1f} launches  {k}")
    # Per hardware queue: busy time and the largest idle gaps (a stalled compute stream shows here).
    queues = collections.defaultdict(list)
    for r in sel:
        queues[(r.get("Queue_Id"), r.get("Stream_Id"))].append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6 / steps
        gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"][:48], b["Kernel_Name"][:48])
                      for a, b in zip(rs, rs[1:]))
        big = [g for g in gaps if g[0] > 20]
        print(f"queue {q}: {len(rs) / steps:.0f} launches, busy {busy:.2f} ms per step, "
              f"{len(big) / steps:.0f} gaps > 20 us totalling {sum(g[0] for g in big) / 1e3 / steps:.2f} ms per step")
        for g in gaps[-5:][::-1]:
            print(f"      {g[0]:9.1f} us between {g[1]} -> {g[2]}")


if __name__ == "__main__":
    main()
