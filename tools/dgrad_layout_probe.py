"""Input-gradient GEMM layout probe: dx = dy . W with W stored [out, in] (the contraction index
is W's ROW index, an "NN" GEMM) against the same product from a transposed copy Wt = W^T stored
[in, out] (both operands contraction-contiguous, "NT"). Times both through the GEMM planner."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402

ext = _ext.require()
if os.environ.get("CAND"):  # widen the planner's candidate list (default 48)
    ext.gemm_planner_configure(True, int(os.environ["CAND"]), int(os.environ.get("RUNS", 4)))
M = int(os.environ.get("M", 32768))
d = int(os.environ.get("D", 1280))
V = 50304
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


for name, n_out, k_in in (("c_attn", 3 * d, d), ("attn.c_proj", d, d), ("c_fc", 4 * d, d), ("mlp.c_proj", d, 4 * d),
                          ("lm_head", V, d)):
    w = torch.randn(n_out, k_in, device="cuda", dtype=torch.bfloat16) * 0.02
    wt = w.t().contiguous()
    dy = torch.randn(M, n_out, device="cuda", dtype=torch.bfloat16)
    ref = ext.lt_matmul(dy, False, w, False)
    out = ext.lt_matmul(dy, False, wt, True)
    err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
    t_nn = bench(lambda: ext.lt_matmul(dy, False, w, False))
    t_nt = bench(lambda: ext.lt_matmul(dy, False, wt, True))
    t_tr = bench(lambda: w.t().contiguous())
    fl = 2 * M * k_in * n_out
    print(f"dgrad {name:12s} rel={err:.1e} NN {t_nn:7.1f} us ({fl / t_nn / 1e6:5.0f} TF)  NT {t_nt:7.1f} us ({fl / t_nt / 1e6:5.0f} TF)"
          f"  transpose {t_tr:6.1f} us", flush=True)
