"""Sharded data-parallel (FSDP) training:
``torchrun --nproc_per_node=N train_fsdp.py --strategy {FULL_SHARD,SHARD_GRAD_OP,NO_SHARD}``.

The reference's ``train_fsdp.py`` run (assignments/assignment1/train_fsdp.py:86-190: every
``transformer.h[i]`` wrapped on its own, then the root; same trainer and hyper-parameters as
DDP; traces in ``outputs/traces/fsdp_{strategy}/rank{r}_trace.json``) on this framework's
flat-parameter runtime. ``--fsdp-unit-blocks k`` groups k blocks per communication unit; without the
flag the unit size follows a start-up measurement of the all-gather curve (one block per unit, the
reference's wrapping, on a single rank or with ``--no-comm-autotune``).
"""
import argparse
import os

import torch

from pytorch_distributed_amd.parallel import (FSDP, ShardingStrategy, autotune, fsdp_wrap_gpt, init_distributed,
                                              plan_sync, shutdown)
from pytorch_distributed_amd.train import DistributedTrainer
from pytorch_distributed_amd.utils import StepProfiler, cli
from pytorch_distributed_amd.utils.profiler import null_profiler


def setup_fsdp_model(model, device, sharding_strategy_name="FULL_SHARD", unit_blocks=1, reduce_fp32=False,
                     force_collectives=False, resident_gb=0.0):
    """Per-block wrapping plus the root unit (the reference's layout when ``unit_blocks == 1``)."""
    model = model.to(device)
    strategy = ShardingStrategy[sharding_strategy_name]
    kw = dict(reduce_fp32=reduce_fp32, force_collectives=force_collectives, resident_gb=resident_gb)
    if unit_blocks == 1:
        for i, block in enumerate(model.transformer.h):
            model.transformer.h[i] = FSDP(block, sharding_strategy=strategy, **kw)
        return FSDP(model, sharding_strategy=strategy, **kw)
    return fsdp_wrap_gpt(model, strategy, unit_blocks=unit_blocks, **kw)


def main():
    ap = cli.add_common_args(argparse.ArgumentParser(description=__doc__))
    ap.add_argument("--strategy", type=str, default="FULL_SHARD",
                    choices=["FULL_SHARD", "SHARD_GRAD_OP", "NO_SHARD"], help="FSDP sharding strategy")
    ap.add_argument("--fsdp-unit-blocks", type=int, default=None,
                    help="transformer blocks per sharding unit (default: from the start-up measurement)")
    ap.add_argument("--fsdp-resident-gb", default="0",
                    help="FULL_SHARD: GiB of unsharded parameters that may stay allocated from forward to backward "
                         "(the last units keep theirs and are not all-gathered again); 'auto' = a tenth of free memory")
    ap.add_argument("--no-comm-autotune", action="store_true", help="skip the start-up collective measurement")
    ap.add_argument("--force-collectives", action="store_true",
                    help="issue the collectives even at world size 1 (single-GPU rehearsal of the communication path)")
    ap.add_argument("--reduce-fp32", action="store_true", help="reduce bf16 gradients through an fp32 copy")
    ap.add_argument("--backend", default=None, choices=[None, "rccl", "socket", "c10d", "nccl", "gloo"])
    ap.add_argument("--sharded-checkpoint", default=None,
                    help="directory: also write a per-rank sharded checkpoint at the end of training")
    args = ap.parse_args()
    rank, world, device = init_distributed(args.backend)
    dtype = cli.resolve_dtype(args, device)
    if rank == 0:
        print(f"\nFSDP Configuration:\nSharding strategy: {args.strategy}\n"
              f"Global batch size: {args.global_batch_size}\nMicro batch size per rank: {args.micro_batch_size}\n"
              f"World size: {world}\nSequence length: {args.sequence_length}\nMax steps: {args.max_steps}")
    policy = autotune.tune(device=device) if world > 1 and not args.no_comm_autotune else None
    model, cfg = cli.build_model(args, device, dtype)
    if device.type == "cuda" and dtype == torch.bfloat16:
        plan_sync.plan_gemms(model, args.micro_batch_size, args.sequence_length, cfg.vocab_size)
    unit_blocks = args.fsdp_unit_blocks
    if unit_blocks is None:
        blocks = model.transformer.h
        block_bytes = sum(p.numel() * p.element_size() for p in blocks[0].parameters())
        unit_blocks = policy.fsdp_unit_blocks(block_bytes, len(blocks)) if policy is not None else 1
    resident = "auto" if args.fsdp_resident_gb == "auto" else float(args.fsdp_resident_gb)
    model = setup_fsdp_model(model, device, args.strategy, unit_blocks, args.reduce_fp32, args.force_collectives,
                             resident_gb=resident)
    if rank == 0:
        print(f"Model wrapped with FSDP ({args.strategy}, {unit_blocks} block(s) per unit)")
    loader = cli.build_loader(args, cfg, device, distributed=True, rank=rank)
    optimizer, scheduler = cli.build_optimizer(args, model.parameters())
    trainer = DistributedTrainer(model, optimizer, scheduler, max_steps=args.max_steps,
                                 global_batch_size=args.global_batch_size,
                                 micro_batch_size=args.micro_batch_size, log_every_n_steps=args.log_every,
                                 save_every_n_steps=args.save_every, checkpoint_dir=args.checkpoint_dir,
                                 ddp_enabled=True, metrics_file=args.metrics_file)
    resume = cli.resolve_resume(args)
    if resume:
        trainer.load_checkpoint(resume)
    trace_dir = args.trace_dir or f"outputs/traces/fsdp_{args.strategy.lower()}"
    prof = null_profiler() if args.no_profile else StepProfiler(
        wait=2, warmup=2, active=6, repeat=1, trace_path=os.path.join(trace_dir, f"rank{rank}_trace.json"),
        backend=args.profiler_backend, rank=rank)
    with prof as p:
        trainer.train(loader, profiler=p)
    if args.sharded_checkpoint:
        from pytorch_distributed_amd.parallel.checkpoint import save_sharded_checkpoint
        save_sharded_checkpoint(args.sharded_checkpoint, model, optimizer, step=trainer.current_step,
                                lr_scheduler=scheduler)
    if rank == 0:
        print(f"FSDP training with {args.strategy} completed!")
        if not args.no_profile:
            print(f"Traces saved to: {trace_dir}/")
    shutdown()


if __name__ == "__main__":
    main()
