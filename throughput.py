"""Training-throughput measurement, extrapolation and batch-size sweep.

The reference's assignment-0 tool (assignments/assignment0/throughput.py:13-220): AdamW 3e-4,
random tokens, 5 warm-up + 20 timed steps between device synchronisations -> tokens/s;
extrapolate to a 1T-parameter model and 10T tokens; sweep batch sizes {1,4,8,16,32,64}
recording tokens/s and peak memory until out of memory. Adds MFU against the MI355X dense
bf16 peak. (The reference's own arithmetic ships stubbed out and its extrapolation call has
the wrong arity; this one runs.)
"""
import argparse
import time

import torch

from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config
from pytorch_distributed_amd.optim import FusedAdamW
from pytorch_distributed_amd.utils import mfu, model_flops_per_token


def measure_tokens_per_second(model, batch_size, seq_length, num_steps=20, warmup=5, verbose=True):
    device = next(model.parameters()).device
    vocab = model.config.vocab_size
    optimizer = FusedAdamW(model.parameters(), lr=3e-4)
    ids = torch.randint(0, vocab, (batch_size, seq_length), device=device)
    targets = torch.randint(0, vocab, (batch_size, seq_length), device=device)
    tokens_per_step = batch_size * seq_length
    model.train()

    def step():
        model(ids, targets).backward()
        optimizer.step(zero_grad=True)

    for _ in range(warmup):
        step()
    if device.type == "cuda":
        torch.cuda.synchronize()
    start = time.time()
    for _ in range(num_steps):
        step()
    if device.type == "cuda":
        torch.cuda.synchronize()
    elapsed = time.time() - start
    total_tokens = num_steps * tokens_per_step
    tps = total_tokens / elapsed
    if verbose:
        print(f"  Total tokens: {total_tokens:,}\n  Elapsed: {elapsed:.3f}s\n  Tokens/s: {tps:,.0f}\n"
              f"  Steps/s: {num_steps / elapsed:.2f}")
    return tps


def extrapolate_modern_training(tokens_per_sec, num_params, target_params=1e12, target_tokens=10e12):
    """Days to train a ``target_params`` model on ``target_tokens`` at this per-parameter speed."""
    scale = target_params / num_params
    scaled_tps = tokens_per_sec / scale
    days = target_tokens / scaled_tps / 86400.0
    print(f"  Scale factor to {target_params:.0e} parameters: {scale:,.0f}x")
    print(f"  Extrapolated tokens/s: {scaled_tps:,.2f}")
    print(f"  Time for {target_tokens:.0e} tokens: {days:,.0f} days ({days / 365:,.1f} years) on one GPU")
    return days


def compare_batch_sizes(config, seq_length, device, dtype, batch_sizes=(1, 4, 8, 16, 32, 64), num_steps=10):
    results = []
    for bs in batch_sizes:
        try:
            if device == "cuda":
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats()
            model = MyGPT2LMHeadModel(config).to(device=device, dtype=dtype)
            tps = measure_tokens_per_second(model, bs, seq_length, num_steps=num_steps, warmup=3, verbose=False)
            mem = torch.cuda.max_memory_allocated() / 2**20 if device == "cuda" else float("nan")
            results.append((bs, tps, mem))
            del model
        except torch.OutOfMemoryError:
            print(f"  batch {bs}: out of memory, stopping the sweep")
            break
    fpt = model_flops_per_token(type("M", (), {"config": config})(), seq_length)
    print(f"\n{'batch':>6} {'tokens/s':>12} {'MFU':>7} {'peak MB':>10}")
    for bs, tps, mem in results:
        print(f"{bs:>6} {tps:>12,.0f} {100 * mfu(tps, fpt):>6.1f}% {mem:>10.0f}")
    return results


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--model", default="gpt2")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--seq-length", type=int, default=1024)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-sweep", action="store_true")
    args = ap.parse_args()
    device = "cuda" if torch.cuda.is_available() else "cpu"
    dtype = torch.bfloat16 if (args.dtype == "bf16" and device == "cuda") else torch.float32
    config = get_config(args.model).without_dropout()
    if device == "cpu":
        args.batch_size, args.seq_length = 2, 64
    model = MyGPT2LMHeadModel(config).to(device=device, dtype=dtype)
    n = sum(p.numel() for p in model.parameters())
    print(f"Model {args.model}: {n:,} parameters on {device} ({dtype})")
    print(f"\nMeasuring throughput (batch {args.batch_size}, T {args.seq_length}):")
    tps = measure_tokens_per_second(model, args.batch_size, args.seq_length,
                                    num_steps=20 if device == "cuda" else 3, warmup=5 if device == "cuda" else 1)
    fpt = model_flops_per_token(model, args.seq_length)
    print(f"  MFU: {100 * mfu(tps, fpt):.1f}% of MI355X dense bf16 peak")
    print("\nExtrapolation:")
    extrapolate_modern_training(tps, n)
    del model
    if not args.no_sweep and device == "cuda":
        print("\nBatch-size sweep:")
        compare_batch_sizes(config, args.seq_length, device, dtype)


if __name__ == "__main__":
    main()
