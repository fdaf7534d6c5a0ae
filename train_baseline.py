"""Single-GPU baseline training with profiling.

Same run as the reference's ``train_baseline.py`` when started without arguments
(GPT-2 Large, global 32 / micro 8, T = 1024, 20 steps, AdamW 3e-4 wd 0.1, cosine schedule,
profiler schedule wait 2 / warmup 2 / active 6, trace in ``outputs/traces/baseline/trace.json``:
assignments/assignment1/train_baseline.py:18-94), on the fused gfx950 kernels.
Without network access add ``--synthetic-data`` (or ``--make-synthetic-shards``).
"""
import argparse
import os

import torch

from pytorch_distributed_amd.train import Trainer
from pytorch_distributed_amd.utils import StepProfiler, cli
from pytorch_distributed_amd.utils.profiler import null_profiler


def main():
    args = cli.add_common_args(argparse.ArgumentParser(description=__doc__)).parse_args()
    device = "cuda" if torch.cuda.is_available() else "cpu"
    dtype = cli.resolve_dtype(args, device)
    print(f"Using device: {device}, dtype: {dtype}")
    model, cfg = cli.build_model(args, device, dtype)
    print(f"Model parameters: {sum(p.numel() for p in model.parameters()):,}")
    print(f"Global batch size: {args.global_batch_size}\nMicro batch size: {args.micro_batch_size}\n"
          f"Gradient accumulation steps: {args.global_batch_size // args.micro_batch_size}\n"
          f"Sequence length: {args.sequence_length}")
    loader = cli.build_loader(args, cfg, device, distributed=False)
    if device == "cuda" and cli.resolve_grad_accum_dtype(args, dtype) == "fp32":
        # bf16 gradients summed over several micro-batches: sum them in fp32 buffers, as the reference's
        # fp32 training loop does (assignments/assignment1/train/trainer.py:57-61)
        from pytorch_distributed_amd.parallel import GradAccumulator
        model = GradAccumulator(model, grad_accum_dtype="fp32")
        print("Gradient accumulation in fp32 buffers")
    optimizer, scheduler = cli.build_optimizer(args, model.parameters())
    trainer = Trainer(model, optimizer, scheduler, max_steps=args.max_steps,
                      global_batch_size=args.global_batch_size, micro_batch_size=args.micro_batch_size,
                      log_every_n_steps=args.log_every, save_every_n_steps=args.save_every,
                      checkpoint_dir=args.checkpoint_dir, metrics_file=args.metrics_file,
                      merge_micro_batches=cli.resolve_merge(args),
                      relax_recompute=args.activation_checkpoint == "auto")
    resume = cli.resolve_resume(args)
    if resume:
        trainer.load_checkpoint(resume)
    trace_dir = args.trace_dir or "outputs/traces/baseline"
    prof = null_profiler() if args.no_profile else StepProfiler(
        wait=2, warmup=2, active=6, repeat=1, trace_path=os.path.join(trace_dir, "trace.json"),
        backend=args.profiler_backend)
    with prof as p:
        trainer.train(loader, profiler=p)
    print("Training completed!")
    if not args.no_profile:
        print(f"Trace saved to: {trace_dir}/trace.json")


if __name__ == "__main__":
    main()
