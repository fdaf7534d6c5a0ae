from .flops import gpt_flops_per_token, mfu, model_flops_per_token
from .memory import estimate_training_memory
from .profiler import StepProfiler

__all__ = ["gpt_flops_per_token", "mfu", "model_flops_per_token", "estimate_training_memory", "StepProfiler"]
