"""Times the native weight-gradient kernel against torch.mm(dy.t(), x) + dy.sum(0) on GPT-2 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402
from tools.gemm_probe import timeit  # noqa: E402

ext = _ext.require()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for name, K, N in [("c_attn", 1280, 3840), ("attn.c_proj", 1280, 1280), ("c_fc", 1280, 5120), ("mlp.c_proj", 5120, 1280),
                   ("med c_fc", 1024, 4096), ("small c_fc", 768, 3072)]:
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    t_lib = timeit(lambda: (torch.mm(dy.t(), x), dy.sum(0)))
    t_nat = timeit(lambda: ext.linear_wgrad(dy, x, True))
    print(f"{name:12s} M={M} I={N} J={K}: library {fl/t_lib/1e12:7.1f} TF ({t_lib*1e6:7.1f} us incl. bias sum) | "
          f"native {fl/t_nat/1e12:7.1f} TF ({t_nat*1e6:7.1f} us)")
