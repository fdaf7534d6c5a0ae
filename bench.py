"""Headline benchmark: GPT training throughput (tokens/s, MFU) on N MI355X GPUs.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N > 1 launched through
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``, one rank per
GPU over RCCL). W untimed warm-up steps, then exactly K optimizer steps timed between
barrier + device-synchronise pairs; the maximum over ranks is reported and rank 0 prints ONE
JSON line.

Workload (the reference's assignment-1 configuration, weak-scaled): GPT-2 Large
(774 M parameters), sequence length 1024, 32 sequences per GPU per optimizer step
(global batch 32 x N; the reference trains global 32 / micro 8 on its GPUs:
assignments/assignment1/train_baseline.py:24-31), AdamW lr 3e-4 wd 0.1, bf16 compute with fp32
master weights and optimizer state, synthetic uniform-random tokens, random-init weights,
dropout 0. One step = every micro-batch's forward + backward, gradient synchronisation and
the optimizer update.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="gpt2-large")
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--batch-per-gpu", type=int, default=32, help="sequences per GPU per optimizer step")
    ap.add_argument("--micro-batch", type=int, default=32, help="sequences per forward/backward")
    ap.add_argument("--mode", choices=["fsdp", "ddp"], default="fsdp",
                    help="data-parallel flavour for --gpus > 1")
    ap.add_argument("--strategy", default="FULL_SHARD", choices=["FULL_SHARD", "SHARD_GRAD_OP", "NO_SHARD"])
    ap.add_argument("--bucket-mb", type=float, default=None)
    ap.add_argument("--reduce-fp32", action="store_true", help="reduce bf16 gradients through an fp32 copy")
    ap.add_argument("--fsdp-unit-blocks", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--init-device", choices=["auto", "cpu", "gpu"], default="auto",
                    help="construct the model on the host (then move) or directly on the GPU; auto = GPU above 1e9 parameters")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="attention / residual / embedding dropout probability (GPT-2 configurations; 0 = off)")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="library-default GEMM solution selection")
    ap.add_argument("--planner-stats", action="store_true", help="print the GEMM planner's per-shape choices to stderr")
    ap.add_argument("--hip-graph", action="store_true",
                    help="single GPU, experimental: record the whole optimizer step into a HIP graph and replay it")
    ap.add_argument("--wrap-single", action="store_true",
                    help="with --gpus 1: still wrap in DDP/FSDP and force the RCCL collectives (rehearsal)")
    return ap.parse_args()


def main():
    args = parse_args()
    # The contract is ONE JSON line on stdout. Libraries print banners of their own there
    # (RCCL's version block, process-group messages), so everything written to fd 1 while the
    # benchmark runs is sent to stderr and the result line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")

    import torch.distributed as dist
    from pytorch_distributed_amd import MI355X_PEAK_BF16_FLOPS
    from pytorch_distributed_amd.data import SyntheticTokenLoader
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.utils.flops import model_flops_per_token

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    tuned = False
    if not args.no_tuned_gemms:
        from pytorch_distributed_amd.tuning import enable_tuned_gemms
        tuned = enable_tuned_gemms()
    distributed = world > 1 or args.wrap_single
    if distributed:
        from pytorch_distributed_amd.parallel import init_distributed
        init_distributed(backend="nccl")

    cfg = get_config(args.model).without_dropout()
    if args.dropout > 0 and hasattr(cfg, "attn_pdrop"):
        cfg.attn_pdrop = cfg.resid_pdrop = cfg.embd_pdrop = args.dropout
    torch.manual_seed(42)
    import contextlib
    from pytorch_distributed_amd.models import LlamaForCausalLM, init_on_device
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cls = LlamaForCausalLM if hasattr(cfg, "intermediate_size") else MyGPT2LMHeadModel
    on_gpu = args.init_device == "gpu"
    if args.init_device == "auto":
        with init_on_device("meta"):
            on_gpu = sum(p.numel() for p in cls(cfg).parameters()) > 1e9
    torch.cuda.manual_seed(42)
    with init_on_device(device, dtype) if on_gpu else contextlib.nullcontext():
        model = cls(cfg)
    model = model.to(device=device, dtype=dtype)

    parallelism = "single"
    if distributed:
        if args.mode == "ddp":
            from pytorch_distributed_amd.parallel import DDP
            kw = {} if args.bucket_mb is None else {"bucket_cap_mb": args.bucket_mb}
            model = DDP(model, device_ids=[local_rank], force_collectives=args.wrap_single,
                        reduce_fp32=args.reduce_fp32, **kw)
            parallelism = f"ddp{world}"
        else:
            from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt
            kw = {} if args.fsdp_unit_blocks is None else {"unit_blocks": args.fsdp_unit_blocks}
            model = fsdp_wrap_gpt(model, ShardingStrategy[args.strategy], force_collectives=args.wrap_single,
                                  reduce_fp32=args.reduce_fp32, **kw)
            parallelism = f"fsdp_{args.strategy.lower()}{world}"
    if args.hip_graph and distributed:
        raise SystemExit("--hip-graph is a single-process mode")
    optimizer = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=0.1, capturable=args.hip_graph)

    if args.batch_per_gpu % args.micro_batch != 0:
        raise SystemExit("--batch-per-gpu must be a multiple of --micro-batch")
    accum = args.batch_per_gpu // args.micro_batch
    loader = iter(SyntheticTokenLoader(args.micro_batch, args.seq_len, cfg.vocab_size, device, seed=1234 + rank))
    model.train()

    def one_step():
        for i in range(accum):
            inputs, targets = next(loader)
            last = i == accum - 1
            if distributed and not last:
                with model.no_sync():
                    (model(inputs, targets) / accum).backward()
            else:
                (model(inputs, targets) / accum).backward()
        optimizer.step(zero_grad=True)

    if args.hip_graph:
        from pytorch_distributed_amd.utils.graphs import GraphedTrainStep
        graphed = GraphedTrainStep(model, optimizer, args.micro_batch, args.seq_len, accum, warmup_steps=2)

        def one_step():  # noqa: F811 - replaces the eager step above
            xs, ys = zip(*(next(loader) for _ in range(accum)))
            graphed(torch.stack(xs), torch.stack(ys))

        for _ in range(3):  # 2 eager warm-up steps + the capture happen before the timed region
            one_step()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    global_batch = args.batch_per_gpu * world
    tokens_per_step = global_batch * args.seq_len
    tps = tokens_per_step * args.steps / elapsed
    fpt = model_flops_per_token(_cfg_holder(cfg), args.seq_len)
    mfu = tps * fpt / (world * MI355X_PEAK_BF16_FLOPS)
    if rank == 0:
        result_line = json.dumps({
            "metric": f"tokens/sec (whole job), {_pretty(args.model)} training, {args.dtype}", "value": tps, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "mfu": mfu, "tflops_per_gpu": tps * fpt / world / 1e12,
            "peak_mem_gb": torch.cuda.max_memory_allocated(device) / 2**30,
            "config": {"model": args.model, "global_batch": global_batch, "micro_batch": args.micro_batch,
                       "grad_accum": accum, "seq_len": args.seq_len, "parallelism": parallelism,
                       "dropout": args.dropout,
                       "optimizer": "fused AdamW lr=3e-4 wd=0.1, fp32 master + state",
                       "flops_per_token": fpt, "tuned_gemm_table": tuned, "hip_graph": bool(args.hip_graph)},
        })
        sys.stdout.flush()
        os.write(real_stdout, (result_line + "\n").encode())
    if args.planner_stats and rank == 0:
        from pytorch_distributed_amd import _ext
        st = _ext.require().gemm_planner_stats()
        for row in sorted(st["table"]):
            print("planner m=%d n=%d k=%d ta=%d tb=%d ep=%d candidates=%d chosen=%d first=%.1fus best=%.1fus"
                  % tuple(row), file=sys.stderr)
    if distributed:
        dist.destroy_process_group()


def _pretty(model_name: str) -> str:
    return {"gpt2": "GPT-2-small", "gpt2-medium": "GPT-2-medium", "gpt2-large": "GPT-2-large",
            "gpt2-xl": "GPT-2-XL"}.get(model_name, model_name)


class _cfg_holder:
    def __init__(self, cfg):
        self.config = cfg


if __name__ == "__main__":
    main()
