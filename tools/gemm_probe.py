"""Times the GEMM shapes of a GPT-2 training step through torch (hipBLASLt) on one GPU.

usage: python tools/gemm_probe.py [--model gpt2-large] [--batch 8] [--seq 1024]
Prints TFLOP/s per shape for forward, dgrad and wgrad, including the LM head with the
odd vocabulary (50257) and with rows padded to 50304, and the strided-output variant the
fused LM-head op uses.
"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd.models import get_config  # noqa: E402


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="gpt2-large")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1024)
    args = ap.parse_args()
    cfg = get_config(args.model)
    M, d, V = args.batch * args.seq, cfg.n_embd, cfg.vocab_size
    dev, dt = "cuda", torch.bfloat16
    shapes = [("c_attn", d, 3 * d), ("attn.c_proj", d, d), ("c_fc", d, 4 * d), ("mlp.c_proj", 4 * d, d)]
    total = 0.0
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt) * 0.02
        b = torch.zeros(N, device=dev, dtype=dt)
        dy = torch.randn(M, N, device=dev, dtype=dt)
        fl = 2.0 * M * K * N
        t_f = timeit(lambda: torch.nn.functional.linear(x, w, b))
        t_d = timeit(lambda: torch.mm(dy, w))
        t_w = timeit(lambda: torch.mm(dy.t(), x))
        total += (t_f + t_d + t_w) * cfg.n_layer
        print(f"{name:12s} M={M} K={K} N={N}: fwd {fl/t_f/1e12:7.1f} TF ({t_f*1e6:7.1f} us) | "
              f"dgrad {fl/t_d/1e12:7.1f} TF ({t_d*1e6:7.1f} us) | wgrad {fl/t_w/1e12:7.1f} TF ({t_w*1e6:7.1f} us)")
    h = torch.randn(M, d, device=dev, dtype=dt)
    for label, Vp in (("lm_head V", V), ("lm_head Vpad", (V + 127) // 128 * 128)):
        w = torch.randn(Vp, d, device=dev, dtype=dt) * 0.02
        fl = 2.0 * M * d * Vp
        dl = torch.randn(M, Vp, device=dev, dtype=dt)
        t_f = timeit(lambda: torch.mm(h, w.t()))
        t_d = timeit(lambda: torch.mm(dl, w))
        t_w = timeit(lambda: torch.mm(dl.t(), h))
        print(f"{label:12s} M={M} K={d} N={Vp}: fwd {fl/t_f/1e12:7.1f} TF ({t_f*1e6:7.1f} us) | "
              f"dgrad {fl/t_d/1e12:7.1f} TF ({t_d*1e6:7.1f} us) | wgrad {fl/t_w/1e12:7.1f} TF ({t_w*1e6:7.1f} us)")
    # strided variant: logits rows padded to Vp, real weight with V rows
    Vp = (V + 127) // 128 * 128
    w = torch.randn(V, d, device=dev, dtype=dt) * 0.02
    buf = torch.empty(M, Vp, device=dev, dtype=dt)
    fl = 2.0 * M * d * V
    t_f = timeit(lambda: torch.mm(h, w.t(), out=buf[:, :V]))
    t_d = timeit(lambda: torch.mm(buf[:, :V], w))
    t_w = timeit(lambda: torch.mm(buf[:, :V].t(), h))
    total += t_f + t_d + t_w
    print(f"{'lm_head ld':12s} M={M} K={d} N={V} ld={Vp}: fwd {fl/t_f/1e12:7.1f} TF ({t_f*1e6:7.1f} us) | "
          f"dgrad {fl/t_d/1e12:7.1f} TF ({t_d*1e6:7.1f} us) | wgrad {fl/t_w/1e12:7.1f} TF ({t_w*1e6:7.1f} us)")
    print(f"sum of GEMM time for one fwd+bwd of {args.model}: {total*1e3:.2f} ms")
    # streaming bandwidth reference point
    a = torch.empty(1 << 28, device=dev, dtype=dt)
    t = timeit(lambda: a.add_(1.0))
    print(f"elementwise add_ over {a.numel()*2/2**20:.0f} MiB: {2*a.numel()*2/t/1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
