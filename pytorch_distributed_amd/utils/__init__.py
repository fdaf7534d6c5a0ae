from .flops import gpt_flops_per_token, mfu, model_flops_per_token

__all__ = ["gpt_flops_per_token", "mfu", "model_flops_per_token"]
