"""Times the attention kernels with HIP events (GPT-2 Large shape by default), without and with
attention dropout."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402

ext = _ext.require()
B, T, H, D = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 1024, 20, 64)
qkv = torch.randn(B, T, 3 * H * D, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B, T, H * D, device="cuda", dtype=torch.bfloat16)
sc = D ** -0.5


def bench(fn, n=20):
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


fl = 4 * B * H * T * T * D / 2
for p in (0.0, float(os.environ.get("DROPOUT", 0.1))):
    out, lse = ext.attention_forward(qkv, H, sc, p, 1234)
    tf = bench(lambda: ext.attention_forward(qkv, H, sc, p, 1234))
    tb = bench(lambda: ext.attention_backward(qkv, out, dout, lse, H, sc, p, 1234))
    print(f"B={B} T={T} H={H} D={D} dropout={p}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF causal)  bwd {tb:.1f} us ({2.5 * fl / tb / 1e6:.0f} TF causal, 5-GEMM count)")
