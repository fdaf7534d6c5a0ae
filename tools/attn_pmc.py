"""Runs the attention forward + backward kernels on the GPT-2 Large shape (for rocprofv3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402

ext = _ext.require()
B, T, H, D = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 1024, 20, 64)
qkv = torch.randn(B, T, 3 * H * D, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B, T, H * D, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    out, lse = ext.attention_forward(qkv, H, D ** -0.5)
    ext.attention_backward(qkv, out, dout, lse, H, D ** -0.5)
torch.cuda.synchronize()
