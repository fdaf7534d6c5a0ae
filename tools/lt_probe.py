"""GEMM planner probe: numerics of the fused epilogues against PyTorch and timing against torch.matmul."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402

ext = _ext.require()
if os.environ.get("CAND"):
    ext.gemm_planner_configure(True, int(os.environ["CAND"]), int(os.environ.get("RUNS", 3)))
dev = "cuda"
M = int(os.environ.get("M", 32768))
d = int(os.environ.get("D", 1280))
torch.manual_seed(0)


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


x = torch.randn(M, d, device=dev, dtype=torch.bfloat16)
shapes = () if os.environ.get('SKIP_PLAIN') else (("c_attn", 3 * d), ("attn.c_proj", d), ("c_fc", 4 * d), ("mlp.c_proj", -4 * d))
for name, n_out in shapes:
    k_in = d
    if n_out < 0:
        k_in, n_out = -n_out, d
    x = torch.randn(M, k_in, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n_out, k_in, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(n_out, device=dev, dtype=torch.bfloat16)
    ref = F.linear(x, w, b)
    out = ext.lt_matmul(x, False, w, True, None, "bias", b)
    t_ref = bench(lambda: F.linear(x, w, b))
    t_lt = bench(lambda: ext.lt_matmul(x, False, w, True, None, "bias", b))
    fl = 2 * M * k_in * n_out
    print(f"fwd {name:12s} rel={rel(out, ref):.2e} torch {t_ref:7.1f} us ({fl / t_ref / 1e6:6.0f} TF) planner {t_lt:7.1f} us ({fl / t_lt / 1e6:6.0f} TF)")
    dy = torch.randn(M, n_out, device=dev, dtype=torch.bfloat16)
    ref = dy @ w
    out = ext.lt_matmul(dy, False, w, False)
    t_ref = bench(lambda: torch.mm(dy, w))
    t_lt = bench(lambda: ext.lt_matmul(dy, False, w, False))
    print(f"dgr {name:12s} rel={rel(out, ref):.2e} torch {t_ref:7.1f} us ({fl / t_ref / 1e6:6.0f} TF) planner {t_lt:7.1f} us ({fl / t_lt / 1e6:6.0f} TF)")

st = ext.gemm_planner_stats()
print("plans", st["plans"], "tuned", st["tuned"])
for row in st["table"]:
    print("  m=%d n=%d k=%d ta=%d tb=%d ep=%d candidates=%d chosen=%d first=%.1fus best=%.1fus" % tuple(row))
