#!/usr/bin/env python3
"""Average duration and launch count per kernel over the last optimizer steps of a rocprofv3 kernel
trace of bench.py (steps delimited by the cross-entropy kernel, as in tools/step_breakdown.py).

usage: python tools/kernel_avg.py <kernel_trace.csv> [steps] [name-filter ...]
"""
import collections
import csv
import sys


def main():
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    filters = sys.argv[3:]
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "::ce_kernel" in r["Kernel_Name"]]
    sel = rows[marks[-steps - 1]:marks[-1]]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in sel:
        n = r["Kernel_Name"]
        n = n.replace("pda::(anonymous namespace)::", "").replace("void ", "")[:70]
        tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[n] += 1
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / steps
    print(f"wall {wall:.2f} ms per step")
    for n, t in tot.most_common():
        if filters and not any(f in n for f in filters):
            continue
        print(f"{t / steps / 1e3:8.3f} ms/step  {cnt[n] / steps:6.1f} x {t / cnt[n]:8.1f} us  {n}")


if __name__ == "__main__":
    main()
