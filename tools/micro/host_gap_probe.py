"""What does the host do while the GPU idles before the optimizer kernels of an accumulated step?
GPT-2 Large, 4 micro-batches of 8 x 1024 (selective recomputation), traced in process with the API
rows kept: prints the idle gaps of the compute queue over 200 us, the HIP API calls and host ranges
that fall into each, and host clock times of the phases of one step."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_amd import _ext, ops  # noqa: E402
from pytorch_distributed_amd.data import SyntheticTokenLoader  # noqa: E402
from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config, init_on_device  # noqa: E402
from pytorch_distributed_amd.optim import FusedAdamW  # noqa: E402

ext = _ext.require()
model_name = sys.argv[1] if len(sys.argv) > 1 else "gpt2-large"
micro, accum, seq = 8, 4, 1024
cfg = get_config(model_name).without_dropout()
with init_on_device("cuda"):
    model = MyGPT2LMHeadModel(cfg, activation_checkpoint="selective")
model = model.to(device="cuda", dtype=torch.bfloat16)
opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=0.1)
loader = iter(SyntheticTokenLoader(micro, seq, cfg.vocab_size, torch.device("cuda"), seed=1))
model.train()
tr = ext.RangeTracer(True, True)
tr.set_enabled(False)
marks = []


def step(record=False):
    for i in range(accum):
        t0 = time.perf_counter()
        x, y = next(loader)
        t1 = time.perf_counter()
        with ops.accumulation_window(i, accum):
            tr.push(f"fwd{i}", "user_annotation")
            loss = model(x, y) / accum
            tr.pop()
            t2 = time.perf_counter()
            tr.push(f"bwd{i}", "user_annotation")
            loss.backward()
            tr.pop()
        t3 = time.perf_counter()
        del loss
        if record:
            marks.append((f"mb{i}", t1 - t0, t2 - t1, t3 - t2))
    t4 = time.perf_counter()
    tr.push("optimizer", "user_annotation")
    opt.step(zero_grad=True)
    tr.pop()
    if record:
        marks.append(("opt", time.perf_counter() - t4, 0, 0))


for _ in range(4):
    step()
torch.cuda.synchronize()
tr.set_enabled(True)
step(True)
step(True)
tr.set_enabled(False)
for m in marks:
    print("HOST %-5s loader %.2f ms  forward %.2f ms  backward %.2f ms" % (m[0], 1e3 * m[1], 1e3 * m[2], 1e3 * m[3]))
ev = json.loads(tr.to_json(0))["traceEvents"]
kern = sorted((e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy")), key=lambda e: e["ts"])
host = [e for e in ev if e.get("cat") in ("cuda_runtime", "user_annotation") and e.get("ph") == "X"]
print("kernels", len(kern), "host rows", len(host))
for a, b in zip(kern, kern[1:]):
    gap = b["ts"] - (a["ts"] + a["dur"])
    if gap > 200:
        print("GAP %.0f us after %s -> before %s" % (gap, a["name"][:50], b["name"][:50]))
        lo, hi = a["ts"] + a["dur"] - 300, b["ts"]
        inside = sorted((e for e in host if e["ts"] + e["dur"] > lo and e["ts"] < hi), key=lambda e: e["ts"])
        big = [e for e in inside if e["dur"] > 30 or e.get("cat") == "user_annotation"]
        for e in big[:40]:
            print("    %-16s %9.0f us (+%.0f)  %s" % (e["cat"], e["dur"], e["ts"] - lo, e["name"][:60]))
        import collections
        print("    calls in the gap:", collections.Counter(e["name"] for e in inside).most_common(8))
