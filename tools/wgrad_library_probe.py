"""Native weight-gradient kernel against the BEST of all hipBLASLt kernels for the same product
(dY^T X, planner in exhaustive mode), per GPT-2 Large shape and token count."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402
from tools.gemm_probe import timeit  # noqa: E402

os.environ["PDA_NO_GEMM_PLAN_FILE"] = "1"
ext = _ext.require()
ext.gemm_planner_configure(True, 48, 3, True)  # exhaustive
for M in [int(a) for a in sys.argv[1:]] or [8192, 32768, 65536]:
    for name, J, I in [("c_attn", 1280, 3840), ("attn.c_proj", 1280, 1280), ("c_fc", 1280, 5120), ("mlp.c_proj", 5120, 1280)]:
        dy = torch.randn(M, I, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(M, J, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * I * J
        lib = lambda: ext.lt_matmul(dy, True, x, False, None, "none", None)  # noqa: E731
        ref = dy.float().t() @ x.float()
        err = ((lib().float() - ref).norm() / ref.norm()).item()
        t_lib = timeit(lib)
        t_nat = timeit(lambda: ext.linear_wgrad(dy, x, False))
        print(f"{name:12s} M={M:6d} I={I} J={J}: library best {fl/t_lib/1e12:7.1f} TF ({t_lib*1e6:7.1f} us, err {err:.1e}) | "
              f"native {fl/t_nat/1e12:7.1f} TF ({t_nat*1e6:7.1f} us)", flush=True)
        del dy, x, ref
