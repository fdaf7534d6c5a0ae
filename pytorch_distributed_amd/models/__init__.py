import contextlib

import torch

from .config import GPT2Config, LlamaConfig, get_config, gpt2_param_count, normalize_gpt2_config
from .llama import LlamaForCausalLM
from .gpt2 import MyGPT2Attention, MyGPT2Block, MyGPT2LMHeadModel, MyGPT2MLP, MyGPT2Model



@contextlib.contextmanager
def init_on_device(device, dtype=None):
    """Build modules with their parameters and buffers allocated (and initialised) directly on
    ``device`` in ``dtype`` instead of in fp32 host memory followed by ``.to(device, dtype)`` as the
    reference does (assignments/assignment1/train_ddp.py:39-51): a 7B-parameter model is 27 GB of
    host RAM and tens of seconds of single-threaded initialisation per rank on the CPU, seconds on
    the GPU. ``device="meta"`` builds the structure without any storage (parameter counting,
    memory planning). Random initial values differ from the CPU generator's for the same seed;
    every rank of a job still gets identical weights from identical seeds."""
    old = torch.get_default_dtype()
    if dtype is not None:
        torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            yield
    finally:
        torch.set_default_dtype(old)


__all__ = ["init_on_device", "GPT2Config", "LlamaConfig", "get_config", "gpt2_param_count", "normalize_gpt2_config",
           "LlamaForCausalLM", "MyGPT2Attention", "MyGPT2Block", "MyGPT2LMHeadModel", "MyGPT2MLP", "MyGPT2Model"]
