"""Fused-epilogue GEMM (csrc/kernels/gemm_fused.hip) against the planner GEMM + stand-alone GELU kernels
on the MLP shapes of the GPT-2 family: microseconds per call, interleaved rounds in one process.

usage: python tools/gemm_fused_probe.py [tokens=32768] [model widths ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import _ext  # noqa: E402
from pytorch_distributed_amd.ops.linear import planned_mm  # noqa: E402

ext = _ext.require()
dev = "cuda"


def timed(fn, rounds=5, inner=4):
    fn()
    fn()
    best = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        e.synchronize()
        best.append(s.elapsed_time(e) * 1e3 / inner)
    best.sort()
    return best[len(best) // 2], best[0]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    widths = [int(a) for a in sys.argv[2:]] or [1280, 1024, 768, 1600]
    g = torch.Generator(device="cpu").manual_seed(0)
    for d in widths:
        N, K = 4 * d, d
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
        u = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
        flops = 2.0 * M * N * K

        def lib_fwd():
            y = planned_mm(x, w, trans_b=True, bias=b)
            return ext.gelu_forward(y)

        def fused_fwd():
            return ext.gemm_fused(x, w, b, "bias_gelu", None)

        def lib_gemm_only():
            return planned_mm(x, w, trans_b=True, bias=b)

        def fused_gemm_only():
            return ext.gemm_fused(x, w, b, "bias", None)

        def lib_bwd():
            t = planned_mm(x, w, trans_b=True)
            return ext.gelu_backward(t, u)

        def fused_bwd():
            return ext.gemm_fused(x, w, None, "mul_gelu_grad", u)

        rows = [("GEMM+bias          library", lib_gemm_only), ("GEMM+bias          fused  ", fused_gemm_only),
                ("GEMM+bias+GELU     library + kernel", lib_fwd), ("GEMM+bias+GELU     fused", fused_fwd),
                ("GEMM * GELU'(u)    library + kernel", lib_bwd), ("GEMM * GELU'(u)    fused", fused_bwd)]
        print(f"--- M={M} N={N} K={K} ({flops / 1e9:.0f} GFLOP)")
        for name, fn in rows:
            med, lo = timed(fn)
            print(f"{name:<40s} median {med:8.1f} us   min {lo:8.1f} us   {flops / med / 1e9:7.1f} TFLOP/s (GEMM flops)")


if __name__ == "__main__":
    main()
