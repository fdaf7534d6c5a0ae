from .config import GPT2Config, LlamaConfig, get_config, gpt2_param_count, normalize_gpt2_config
from .llama import LlamaForCausalLM
from .gpt2 import MyGPT2Attention, MyGPT2Block, MyGPT2LMHeadModel, MyGPT2MLP, MyGPT2Model

__all__ = ["GPT2Config", "LlamaConfig", "get_config", "gpt2_param_count", "normalize_gpt2_config",
           "LlamaForCausalLM", "MyGPT2Attention", "MyGPT2Block", "MyGPT2LMHeadModel", "MyGPT2MLP", "MyGPT2Model"]
