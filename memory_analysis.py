"""Training-memory analysis: analytic estimate vs what the allocator really holds.

Follows the reference tool (assignments/assignment0/memory_analysis.py:12-163): estimate
parameters / gradients / optimizer state, then run three real training steps of GPT-2
small (B = 8, T = 1024, AdamW lr 1e-4) with allocator history recording on, dump the
snapshot to ``outputs/task1_memory_snapshot.pickle`` (viewable at pytorch.org/memory_viz) and
print estimate vs measurement. Differences: the peak is the real peak
(``max_memory_allocated``, the reference prints the current value under that label), the
recording calls are skipped on a CPU-only machine, and ``--dtype bf16`` analyses the
mixed-precision layout this framework trains with.
"""
import argparse
import os

import torch

from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config
from pytorch_distributed_amd.optim import FusedAdamW
from pytorch_distributed_amd.utils import estimate_training_memory


def calculate_memory_breakdown(config, dtype="fp32"):
    """Analytic MB for parameters, gradients and optimizer state of ``config``."""
    model = MyGPT2LMHeadModel(config)
    total_params = sum(p.numel() for p in model.parameters())
    return estimate_training_memory(total_params, dtype)


def profile_actual_memory(config, batch_size, seq_length, device="cuda", dtype=torch.float32,
                          snapshot_path="outputs/task1_memory_snapshot.pickle", steps=3):
    """Runs ``steps`` training steps and returns allocator statistics in MB."""
    on_gpu = torch.device(device).type == "cuda" and torch.cuda.is_available()
    if on_gpu:
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.memory._record_memory_history(enabled="all")
    try:
        model = MyGPT2LMHeadModel(config).to(device=device, dtype=dtype)
        optimizer = FusedAdamW(model.parameters(), lr=1e-4)
        ids = torch.randint(0, config.vocab_size, (batch_size, seq_length), device=device)
        targets = torch.randint(0, config.vocab_size, (batch_size, seq_length), device=device)
        model.train()
        for _ in range(steps):
            loss = model(ids, targets)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
        stats = {"loss": loss.item()}
        if on_gpu:
            torch.cuda.synchronize()
            stats.update(allocated_mb=torch.cuda.memory_allocated() / 2**20,
                         peak_allocated_mb=torch.cuda.max_memory_allocated() / 2**20,
                         reserved_mb=torch.cuda.memory_reserved() / 2**20)
            os.makedirs(os.path.dirname(snapshot_path) or ".", exist_ok=True)
            torch.cuda.memory._dump_snapshot(snapshot_path)
            stats["snapshot"] = snapshot_path
        return stats
    finally:
        if on_gpu:
            torch.cuda.memory._record_memory_history(enabled=None)


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--model", default="gpt2")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--seq-length", type=int, default=1024)
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    args = ap.parse_args()
    device = "cuda" if torch.cuda.is_available() else "cpu"
    config = get_config(args.model).without_dropout()
    print(f"Model: {args.model}, batch {args.batch_size}, sequence length {args.seq_length}, device {device}")
    est = calculate_memory_breakdown(config, args.dtype)
    print(f"\nTotal parameters: {est['total_params']:,}")
    print("Estimated memory (activations excluded):")
    print(f"  Parameters : {est['params_mb']:10.1f} MB")
    print(f"  Gradients  : {est['gradients_mb']:10.1f} MB")
    print(f"  Optimizer  : {est['optimizer_mb']:10.1f} MB")
    print(f"  Total      : {est['total_mb']:10.1f} MB")
    dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
    if device == "cpu":
        args.batch_size, args.seq_length = min(args.batch_size, 2), min(args.seq_length, 128)
    stats = profile_actual_memory(config, args.batch_size, args.seq_length, device, dtype)
    if "allocated_mb" in stats:
        print("\nMeasured (caching allocator):")
        print(f"  Allocated after 3 steps : {stats['allocated_mb']:10.1f} MB")
        print(f"  Peak allocated          : {stats['peak_allocated_mb']:10.1f} MB")
        print(f"  Reserved                : {stats['reserved_mb']:10.1f} MB")
        print(f"  Steady state - estimate : {stats['allocated_mb'] - est['total_mb']:10.1f} MB")
        print(f"  Peak - estimate (activations, logits, workspaces): "
              f"{stats['peak_allocated_mb'] - est['total_mb']:.1f} MB")
        print(f"\nSnapshot: {stats['snapshot']}")
    else:
        print(f"\nNo GPU: ran 3 CPU steps (loss {stats['loss']:.3f}); allocator statistics need a GPU.")


if __name__ == "__main__":
    main()
