"""Data-parallel (DDP) training: ``torchrun --nproc_per_node=N train_ddp.py``.

The reference's ``train_ddp.py`` run (assignments/assignment1/train_ddp.py:54-146: same
hyper-parameters as the baseline, global batch 32 split over the ranks, seed 42 on every
rank, per-rank traces in ``outputs/traces/ddp/rank{r}_trace.json``) on this framework's DDP:
a C++ bucket reducer launching reductions on the framework's own RCCL communicator from autograd
hooks. Runs on CPUs too (``--backend socket``). With 8 GPUs pass ``--global-batch-size 64`` (32 is
smaller than 8 x 8). Bucket size and reduction algorithm come from a start-up measurement of the
fabric unless given (``--bucket-mb``, ``--ddp-decompose``).
"""
import argparse
import os
import time

from pytorch_distributed_amd.parallel.comm import prepare_multiprocess_env

prepare_multiprocess_env()  # before anything initialises the HIP runtime

import torch  # noqa: E402

from pytorch_distributed_amd.parallel import DDP, autotune, init_distributed, plan_sync, shutdown
from pytorch_distributed_amd.train import DistributedTrainer
from pytorch_distributed_amd.utils import StepProfiler, cli
from pytorch_distributed_amd.utils.profiler import null_profiler


def setup_ddp_model(model, device, bucket_cap_mb=None, reduce_fp32=False, decompose=None, force_collectives=False,
                    grad_accum_dtype=None):
    """``model`` (already on ``device``) wrapped in DDP (None = the start-up measurement's choice)."""
    ids = [device.index] if device.type == "cuda" else None
    return DDP(model, device_ids=ids, reduce_fp32=reduce_fp32, bucket_cap_mb=bucket_cap_mb, decompose=decompose,
               force_collectives=force_collectives, grad_accum_dtype=grad_accum_dtype)


def main():
    ap = cli.add_common_args(argparse.ArgumentParser(description=__doc__))
    ap.add_argument("--bucket-mb", type=float, default=None, help="gradient bucket size (default: measured)")
    ap.add_argument("--ddp-decompose", choices=["auto", "on", "off"], default="auto",
                    help="reduce buckets as reduce-scatter + all-gather instead of all-reduce (auto: measured)")
    ap.add_argument("--no-comm-autotune", action="store_true", help="skip the start-up collective measurement")
    ap.add_argument("--force-collectives", action="store_true",
                    help="issue the collectives even at world size 1 (single-GPU rehearsal of the communication path)")
    ap.add_argument("--backend", default=None, choices=[None, "rccl", "socket", "c10d", "nccl", "gloo"])
    ap.add_argument("--no-comm-fallback", action="store_true",
                    help="fail instead of switching to a torch.distributed group when the native communicator does "
                         "not pass its start-up self-test on every rank")
    ap.add_argument("--comm-dual", action="store_true",
                    help="RCCL: a second library communicator for the all-gather queue (gathers and reductions "
                         "run concurrently; with the start-up autotune, only if measured faster)")
    ap.add_argument("--reduce-fp32", action="store_true", help="all-reduce bf16 gradients through an fp32 copy")
    args = ap.parse_args()
    native = args.backend in (None, "rccl", "nccl", "socket", "gloo")
    rank, world, device = init_distributed(args.backend, fallback=None if args.no_comm_fallback or not native else "c10d",
                                           dual=args.comm_dual or None)
    dtype = cli.resolve_dtype(args, device)
    if rank == 0:
        print(f"\nDDP Configuration:\nGlobal batch size: {args.global_batch_size}\n"
              f"Micro batch size per rank: {args.micro_batch_size}\nWorld size: {world}\n"
              f"Sequence length: {args.sequence_length}\nMax steps: {args.max_steps}")
    if world > 1 and not args.no_comm_autotune:
        autotune.tune(device=device)
    model, cfg = cli.build_model(args, device, dtype)  # same seed on every rank
    if device.type == "cuda" and dtype == torch.bfloat16:
        plan_sync.plan_gemms(model, args.micro_batch_size, args.sequence_length, cfg.vocab_size)
    model = setup_ddp_model(model, device, args.bucket_mb, args.reduce_fp32,
                            {"auto": None, "on": True, "off": False}[args.ddp_decompose], args.force_collectives,
                            grad_accum_dtype=cli.resolve_grad_accum_dtype(args, dtype, world))
    if rank == 0:
        print(f"Model wrapped with DDP ({model.reducer.num_buckets} gradient buckets of <= {model.bucket_cap_mb:g} MiB"
              f"{', reduce-scatter + all-gather' if model.decompose else ''}"
              f"{', fp32 gradient accumulation' if model.grad_accum_dtype == 'fp32' else ''})")
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
    trace_dir = args.trace_dir or "outputs/traces/ddp"
    prof = null_profiler() if args.no_profile else StepProfiler(
        wait=2, warmup=2, active=6, repeat=1, trace_path=os.path.join(trace_dir, f"rank{rank}_trace.json"),
        backend=args.profiler_backend, rank=rank)
    with prof as p:
        trainer.train(loader, profiler=p)
    if rank == 0:
        print("DDP training completed!")
        if not args.no_profile:
            print(f"Traces saved to: {trace_dir}/")
    shutdown()


if __name__ == "__main__":
    main()
