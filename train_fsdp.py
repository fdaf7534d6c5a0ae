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
import time

from pytorch_distributed_amd.parallel.comm import prepare_multiprocess_env

prepare_multiprocess_env()  # before anything initialises the HIP runtime

import torch  # noqa: E402

from pytorch_distributed_amd.parallel import (FSDP, ShardingStrategy, autotune, fsdp_wrap_gpt, init_distributed,
                                              plan_sync, shutdown)
from pytorch_distributed_amd.train import DistributedTrainer
from pytorch_distributed_amd.utils import StepProfiler, cli
from pytorch_distributed_amd.utils.profiler import null_profiler


def setup_fsdp_model(model, device, sharding_strategy_name="FULL_SHARD", unit_blocks=1, reduce_fp32=False,
                     force_collectives=False, resident_gb=0.0, grad_accum_dtype=None):
    """Per-block wrapping plus the root unit (the reference's layout when ``unit_blocks == 1``)."""
    model = model.to(device)
    strategy = ShardingStrategy[sharding_strategy_name]
    kw = dict(reduce_fp32=reduce_fp32, force_collectives=force_collectives, resident_gb=resident_gb,
              grad_accum_dtype=grad_accum_dtype)
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
    ap.add_argument("--no-comm-fallback", action="store_true",
                    help="fail instead of switching to a torch.distributed group when the native communicator does "
                         "not pass its start-up self-test on every rank")
    ap.add_argument("--comm-dual", action="store_true",
                    help="RCCL: a second library communicator for the all-gather queue (gathers and reductions "
                         "run concurrently; with the start-up autotune, only if measured faster)")
    ap.add_argument("--sharded-checkpoint", default=None,
                    help="directory: also write a per-rank sharded checkpoint at the end of training")
    ap.add_argument("--resume-sharded", default=None,
                    help="directory of a sharded checkpoint (written with --sharded-checkpoint, at any world size) to "
                         "continue from: parameters, sharded optimizer state, step and schedule")
    args = ap.parse_args()
    native = args.backend in (None, "rccl", "nccl", "socket", "gloo")
    rank, world, device = init_distributed(args.backend, fallback=None if args.no_comm_fallback or not native else "c10d",
                                           dual=args.comm_dual or None)
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
                             resident_gb=resident, grad_accum_dtype=cli.resolve_grad_accum_dtype(args, dtype, world))
    if rank == 0:
        print(f"Model wrapped with FSDP ({args.strategy}, {unit_blocks} block(s) per unit)")
    loader = cli.build_loader(args, cfg, device, distributed=True, rank=rank)
    optimizer, scheduler = cli.build_optimizer(args, model.parameters())
    trainer = DistributedTrainer(model, optimizer, scheduler, max_steps=args.max_steps,
                                 global_batch_size=args.global_batch_size,
                                 micro_batch_size=args.micro_batch_size, log_every_n_steps=args.log_every,
                                 save_every_n_steps=args.save_every, checkpoint_dir=args.checkpoint_dir,
                                 ddp_enabled=True, metrics_file=args.metrics_file,
                                 merge_micro_batches=cli.resolve_merge(args),
                                 relax_recompute=args.activation_checkpoint == "auto")
    resume = cli.resolve_resume(args)
    if resume:
        trainer.load_checkpoint(resume)
    if args.resume_sharded:
        from pytorch_distributed_amd.parallel.checkpoint import load_sharded_checkpoint
        t0 = time.time()
        trainer.current_step = load_sharded_checkpoint(args.resume_sharded, model, optimizer, lr_scheduler=scheduler)
        trainer.batch_count = trainer.current_step * trainer.grad_accumulation_steps
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if rank == 0:
            print(f"Loaded sharded checkpoint {args.resume_sharded} (step {trainer.current_step}) in "
                  f"{time.time() - t0:.1f}s")
    trace_dir = args.trace_dir or f"outputs/traces/fsdp_{args.strategy.lower()}"
    prof = null_profiler() if args.no_profile else StepProfiler(
        wait=2, warmup=2, active=6, repeat=1, trace_path=os.path.join(trace_dir, f"rank{rank}_trace.json"),
        backend=args.profiler_backend, rank=rank)
    with prof as p:
        trainer.train(loader, profiler=p)
    if args.sharded_checkpoint:
        from pytorch_distributed_amd.parallel.checkpoint import save_sharded_checkpoint
        t0 = time.time()
        save_sharded_checkpoint(args.sharded_checkpoint, model, optimizer, step=trainer.current_step,
                                lr_scheduler=scheduler)
        if rank == 0:
            print(f"Saved sharded checkpoint to {args.sharded_checkpoint} in {time.time() - t0:.1f}s")
    if rank == 0:
        print(f"FSDP training with {args.strategy} completed!")
        if not args.no_profile:
            print(f"Traces saved to: {trace_dir}/")
    shutdown()


if __name__ == "__main__":
    main()
