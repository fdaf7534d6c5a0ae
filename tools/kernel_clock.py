#!/usr/bin/env python3
"""Effective shader clock per kernel from a rocprofv3 pass with --pmc GRBM_GUI_ACTIVE --kernel-trace:
clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration. Reads high on dispatches shorter than ~0.3 ms.

usage: python tools/kernel_clock.py <counter_collection.csv> [kernel_trace.csv]
"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    dur = {}
    if len(sys.argv) > 2:
        for r in csv.DictReader(open(sys.argv[2])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in rows:
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        d = dur.get(r["Dispatch_Id"])
        if d is None and "Start_Timestamp" in r:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if not d:
            continue
        n = r["Kernel_Name"].replace("pda::(anonymous namespace)::", "").replace("void ", "")[:64]
        a = acc[n]
        a[0] += float(r["Counter_Value"])
        a[1] += d
        a[2] += 1
    for n, (c, d, k) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if d / k < 20000:
            continue
        print(f"{c / 8 / d:6.2f} GHz  {d / k / 1e3:8.1f} us x {k:5d}  {n}")


if __name__ == "__main__":
    main()
