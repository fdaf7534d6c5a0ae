"""Which kernel dispatches does the in-process kernel tracer see? Variations of stream, launcher and
order of initialisation (diagnosis of kernel rows missing for the default stream in plain runs)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_amd import _ext  # noqa: E402

ext = _ext.require()
mode = sys.argv[1] if len(sys.argv) > 1 else "null"
print("mode", mode, "available", ext.kernel_trace_available(), ext.kernel_trace_unavailable_reason(),
      "WORLD_SIZE", os.environ.get("WORLD_SIZE"), flush=True)
if mode == "early":  # tracer object (and roctracer's pool) before the first GPU call
    tr = ext.RangeTracer(True, True)
    tr.set_enabled(False)
x = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
if mode != "early":
    tr = ext.RangeTracer(True, True)
    tr.set_enabled(False)
side = torch.cuda.Stream()


def work():
    y = x
    for _ in range(5):
        y = (y @ x).relu()
    return y


tr.set_enabled(True)
if mode == "pre_sync":
    torch.cuda.synchronize()
elif mode == "pre_event":
    e0 = torch.cuda.Event()
    e0.record()
elif mode == "pre_kernel_sync":
    z0 = x + 1
    torch.cuda.synchronize()
elif mode == "pre_cross":
    e0 = torch.cuda.Event()
    e0.record()
    side.wait_event(e0)
    torch.cuda.current_stream().wait_stream(side)
tr.push("window", "user_annotation")
if mode == "side":
    with torch.cuda.stream(side):
        work()
elif mode == "sync_each":
    work()
    torch.cuda.current_stream().synchronize()
elif mode == "adds":
    for _ in range(6):
        z = x + 1
elif mode == "adds_mm":
    z = x + 1
    z = z @ x
    z = z + 1
    z = z @ x
    z = z.relu()
elif mode == "torch_sync":
    work()
    torch.cuda.synchronize()
elif mode == "item":
    print(float(work().float().sum().item()))
elif mode == "cross":
    work()
    e = torch.cuda.Event()
    e.record()
    side.wait_event(e)
    with torch.cuda.stream(side):
        z = x + 1
    torch.cuda.current_stream().wait_stream(side)
    work()
else:
    work()
tr.pop()
tr.set_enabled(False)
ev = json.loads(tr.to_json(0))["traceEvents"]
kern = [e for e in ev if e.get("cat") == "kernel"]
maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "roctracer" in ln or "amdhip64" in ln or "roctx" in ln})
print("LIBS", maps)
print("COUNTS buffers/records/other", ext.kernel_trace_debug_counts())
for e in sorted(kern, key=lambda e: e["ts"]):
    print("  K", round(e["ts"], 1), e["args"]["queue"], e["name"][:70])
apis = [e for e in ev if e.get("cat") == "cuda_runtime"]
import collections
print("  API", collections.Counter(e["name"] for e in apis).most_common(8))
print("RESULT", mode, "kernels", len(kern), "queues", sorted({e["args"]["queue"] for e in kern}), flush=True)
interesting = {k: v for k, v in os.environ.items() if k.startswith(("HIP", "HSA", "ROC", "AMD", "LD_", "TORCH", "OMP", "NCCL", "RCCL", "GPU"))}
if mode == "null":
    print("ENV", json.dumps(interesting, sort_keys=True))
