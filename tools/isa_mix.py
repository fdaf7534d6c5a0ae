"""Instruction mix per basic block of one kernel: compiles a .hip file to gfx950 assembly and prints, for
every block with more than `min` instructions, its MFMA / VALU / LDS / SALU counts and most frequent opcodes.
usage: python tools/isa_mix.py <file.hip> <mangled-name-substring> [min]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

src, needle = sys.argv[1], sys.argv[2]
minimum = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = os.path.join(tempfile.gettempdir(), "isa_mix.s")
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch_distributed_amd", "csrc", "kernels")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + inc, "-S", "--cuda-device-only",
                "-o", out, src], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().split(":")[0].endswith(l.split(":")[0])][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
blocks = [["entry", collections.Counter()]]
for l in lines[start + 1:end]:
    t = l.strip()
    if re.match(r"^\.LBB\S+:", t):
        blocks.append([t.split(":")[0], collections.Counter()])
        continue
    if not t or t[0] in ".;/":
        continue
    blocks[-1][1][t.split()[0]] += 1
print(lines[start].split(":")[0])
for name, c in blocks:
    n = sum(c.values())
    if n < minimum:
        continue
    cls = lambda p: sum(v for k, v in c.items() if p(k))  # noqa: E731
    print(f"{name}: {n} instructions, mfma {cls(lambda k: 'mfma' in k)}, valu {cls(lambda k: k.startswith('v_') and 'mfma' not in k)}, "
          f"lds {cls(lambda k: k.startswith('ds_'))}, salu {cls(lambda k: k.startswith('s_'))}")
    print("    " + ", ".join(f"{k} {v}" for k, v in c.most_common(16)))
