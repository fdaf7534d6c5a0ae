"""Shared command-line plumbing of the train scripts.

The reference hard-codes every hyper-parameter inside ``main()``
(assignments/assignment1/train_baseline.py:24-31, train_ddp.py:59-64, train_fsdp.py:98-103).
The flags below default to exactly those literals (GPT-2 Large, global batch 32, micro
batch 8, sequence length 1024, 20 steps, lr 3e-4, weight decay 0.1, cosine to 0.1 x lr, 10
training shards, profiler schedule 2/2/6/1), so running a script without arguments is the
reference's run; everything else is opt-in. Two defaults are decided by measured memory instead of
being fixed: how many of a window's micro-batches run as one pass (``--merge-micro-batches auto``)
and whether the selective recomputation is kept (``--activation-checkpoint auto``); neither changes
the optimizer step beyond summation order, and ``off`` / ``selective`` restore the literal recipe.
"""
from __future__ import annotations

import argparse
import contextlib
import glob
import os
from typing import List, Optional

import torch

from ..data import (DistributedKJJ0DataLoader, KJJ0DataLoader, SyntheticTokenLoader,
                    download_fineweb10B_files, make_synthetic_fineweb_files)
from ..models import MyGPT2LMHeadModel, get_config
from ..optim import CosineAnnealingLR, FusedAdamW


def add_common_args(ap: argparse.ArgumentParser) -> argparse.ArgumentParser:
    ap.add_argument("--model", default="gpt2-large", help="gpt2 | gpt2-medium | gpt2-large | gpt2-xl | llama-7b ...")
    ap.add_argument("--global-batch-size", type=int, default=32)
    ap.add_argument("--micro-batch-size", type=int, default=8)
    ap.add_argument("--sequence-length", type=int, default=1024)
    ap.add_argument("--max-steps", type=int, default=20)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--weight-decay", type=float, default=0.1)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default=None, help="default: bf16 on GPU, fp32 on CPU")
    ap.add_argument("--dropout", type=float, default=None, help="override all dropout probabilities (config: 0.1)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--init-device", choices=["cpu", "gpu"], default="cpu",
                    help="where the model is constructed: cpu = fp32 on the host then moved (the reference's order, "
                         "same weights as a CPU run); gpu = directly on the training device in the training dtype "
                         "(multi-billion-parameter models)")
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--synthetic-data", action="store_true", help="random tokens generated on the device")
    ap.add_argument("--data-dir", default=".cache/data/fineweb10B")
    ap.add_argument("--num-train-files", type=int, default=10)
    ap.add_argument("--prefetch-data", action="store_true",
                    help="distributed scripts: read shards through the native mmap + prefetch-thread loader")
    ap.add_argument("--make-synthetic-shards", action="store_true",
                    help="write random .bin shards into --data-dir instead of downloading")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profiler-backend", choices=["auto", "kineto", "range"], default="auto")
    ap.add_argument("--trace-dir", default=None)
    ap.add_argument("--save-every", type=int, default=None)
    ap.add_argument("--checkpoint-dir", default="checkpoints")
    ap.add_argument("--resume", default=None, help="checkpoint path, or 'latest' to pick the newest in --checkpoint-dir")
    ap.add_argument("--metrics-file", default=None)
    ap.add_argument("--activation-checkpoint", choices=["auto", "selective", "full", "none"], default="auto",
                    help="auto = the reference's selective recomputation, dropped after the first accumulation window "
                         "when the measured memory says everything can be kept (same gradients bit for bit)")
    ap.add_argument("--grad-accum-dtype", choices=["auto", "param", "fp32"], default="auto",
                    help="dtype in which the micro-batches of an accumulation window are "
                         "summed; auto = fp32 when bf16 gradients are accumulated over more than one micro-batch "
                         "(the reference accumulates in fp32), the parameter dtype otherwise")
    ap.add_argument("--merge-micro-batches", default="auto",
                    help="auto | off | all | N: run N of the loader's micro-batches as one forward / backward pass "
                         "(same optimizer step up to summation order, larger GEMMs, fewer collectives); auto = as "
                         "many as the memory left after the first window holds (GPU only)")
    return ap


def model_activation_checkpoint(args) -> str:
    """The policy the model is built with (``auto`` starts as the reference's ``selective``)."""
    return "selective" if args.activation_checkpoint == "auto" else args.activation_checkpoint


def resolve_merge(args):
    """``--merge-micro-batches`` as the trainers' ``merge_micro_batches`` argument."""
    m = str(getattr(args, "merge_micro_batches", "off")).lower()
    return m if m in ("auto", "off", "all") else int(m)


def resolve_grad_accum_dtype(args, dtype, world_size: int = 1) -> str:
    """'fp32' or 'param' from ``--grad-accum-dtype`` (see its help)."""
    if args.grad_accum_dtype != "auto":
        return args.grad_accum_dtype
    steps = args.global_batch_size // max(1, args.micro_batch_size * world_size)
    return "fp32" if dtype != torch.float32 and steps > 1 else "param"


def resolve_dtype(args, device) -> torch.dtype:
    if args.dtype is None:
        return torch.bfloat16 if torch.device(device).type == "cuda" else torch.float32
    return torch.bfloat16 if args.dtype == "bf16" else torch.float32


def build_model(args, device, dtype):
    cfg = get_config(args.model)
    if hasattr(cfg, "attn_pdrop") and args.dropout is not None:
        cfg.attn_pdrop = cfg.resid_pdrop = cfg.embd_pdrop = args.dropout
    torch.manual_seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(args.seed)
    from ..models import init_on_device
    on_gpu = getattr(args, "init_device", "cpu") == "gpu" and torch.device(device).type == "cuda"
    with init_on_device(device, dtype) if on_gpu else contextlib.nullcontext():
        if hasattr(cfg, "intermediate_size"):
            from ..models.llama import LlamaForCausalLM
            model = LlamaForCausalLM(cfg, activation_checkpoint=model_activation_checkpoint(args))
        else:
            model = MyGPT2LMHeadModel(cfg, activation_checkpoint=model_activation_checkpoint(args))
    return model.to(device=device, dtype=dtype), cfg


def build_optimizer(args, params):
    opt = FusedAdamW(params, lr=args.lr, weight_decay=args.weight_decay)
    sched = CosineAnnealingLR(opt, T_max=args.max_steps, eta_min=args.lr * 0.1)
    return opt, sched


def train_files(args, vocab_size: int = 50257) -> List[str]:
    if args.make_synthetic_shards:  # token ids the model's embedding can look up
        files = make_synthetic_fineweb_files(args.data_dir, args.num_train_files, vocab_size=vocab_size)
    else:
        files = download_fineweb10B_files(args.data_dir, num_train_files=args.num_train_files)
    return [str(f) for f in files if "train" in str(f)]


def build_loader(args, cfg, device, distributed: bool, rank: int = 0):
    if args.synthetic_data:
        return SyntheticTokenLoader(args.micro_batch_size, args.sequence_length, cfg.vocab_size, device,
                                    seed=1234 + rank)
    files = train_files(args, cfg.vocab_size)
    if distributed:
        return DistributedKJJ0DataLoader(files, args.micro_batch_size, args.sequence_length, device,
                                         prefetch=getattr(args, "prefetch_data", False))
    return KJJ0DataLoader(files, args.micro_batch_size, args.sequence_length, device)


def resolve_resume(args) -> Optional[str]:
    if not args.resume:
        return None
    if args.resume != "latest":
        return args.resume
    cands = glob.glob(os.path.join(args.checkpoint_dir, "checkpoint_step_*.pt"))
    if not cands:
        return None
    return max(cands, key=lambda p: int(p.rsplit("_", 1)[1].split(".")[0]))
