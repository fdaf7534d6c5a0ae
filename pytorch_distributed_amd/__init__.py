"""pytorch_distributed_amd — an MI355X-native data-parallel training framework.

Single-GPU GPT training with profiling, replicated data parallelism (DDP: a
C++ gradient-bucket reducer driving RCCL all-reduce under the backward pass)
and sharded data parallelism (FSDP: flat-parameter units with all-gather /
reduce-scatter prefetch; FULL_SHARD, SHARD_GRAD_OP, NO_SHARD), with the
training-step hot ops as hand-written gfx950 HIP kernels.
"""
from . import _ext  # noqa: F401

__version__ = "0.1.0"

# Dense bf16 MFMA peak of one MI355X in FLOP/s (AMD's 5 PF figure counts 2:1
# sparsity). The single constant every MFU number in this package is priced against.
MI355X_PEAK_BF16_FLOPS = 2.5e15

__all__ = ["MI355X_PEAK_BF16_FLOPS", "__version__"]
