#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection.csv files: mean counter value per kernel. usage: pmc_summary.py dir [filter]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pda::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{root}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pda::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt in name:
            agg[name]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, ctrs in agg.items():
    print(name)
    for c, v in sorted(ctrs.items()):
        print(f"    {c:28s} {sum(v) / len(v):14.4g}  (n={len(v)})")
