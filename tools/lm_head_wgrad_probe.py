"""LM-head weight gradient dW = dlogits^T . hidden (V = 50257 padded to 50304): torch.mm on the transposed
view, the GEMM planner with a transposed first operand, and the native split-K kernel (all ~1.0 PFLOP/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext
ext = _ext.require()
M, V, vp, d = 32768, 50257, 50304, 1280
dl = torch.zeros(M, vp, device="cuda", dtype=torch.bfloat16); dl[:, :V] = torch.randn(M, V, device="cuda", dtype=torch.bfloat16) * 0.01
h2 = torch.randn(M, d, device="cuda", dtype=torch.bfloat16)
def bench(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = torch.mm(dl[:, :V].t(), h2)
t0 = bench(lambda: torch.mm(dl[:, :V].t(), h2))
out = ext.lt_matmul(dl, True, h2, False)
print("rel", ((out[:V].float() - ref.float()).norm() / ref.float().norm()).item())
t1 = bench(lambda: ext.lt_matmul(dl, True, h2, False))
t2 = bench(lambda: ext.linear_wgrad(dl, h2, False))
fl = 2 * M * V * d
print(f"torch.mm(TN) {t0:.0f} us ({fl/t0/1e6:.0f} TF)  planner TN padded {t1:.0f} us ({fl/t1/1e6:.0f} TF)  native wgrad {t2:.0f} us ({fl/t2/1e6:.0f} TF)")
