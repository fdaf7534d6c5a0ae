"""Runs the native weight-gradient kernel a few times on one shape (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402

ext = _ext.require()
M, I, J = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 5120, 1280)
dy = torch.randn(M, I, device="cuda", dtype=torch.bfloat16)
x = torch.randn(M, J, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ext.linear_wgrad(dy, x, True)
torch.cuda.synchronize()
