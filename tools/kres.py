#!/usr/bin/env python3
"""Summarise per-kernel register / LDS / occupancy figures of one HIP source.

usage: tools/kres.py file.hip [extra hipcc flags]
"""
import re
import subprocess
import sys


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        if " error" in line or "warning:" in line:
            print(line)
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    for r in rows:
        name = subprocess.run(["c++filt", "-p", r["name"]], capture_output=True, text=True).stdout.strip()
        print("%-100s vgpr=%s agpr=%s spill=%s scratch=%s occ=%s lds=%s" % (
            name[:100], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
            r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
