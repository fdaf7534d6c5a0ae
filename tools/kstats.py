#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 kernel_stats.csv: usage tools/kstats.py file.csv [n] [per_step_divisor]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6 / div:.2f} ms")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / div:8.2f} ms {float(r['Percentage']):5.1f}% calls={int(r['Calls']) / div:7.1f} "
          f"avg={float(r['AverageNs']) / 1e3:8.1f}us  {r['Name'][:100]}")
