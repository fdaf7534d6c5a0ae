"""Training-step hot ops: HIP kernels on the GPU, PyTorch references on the CPU."""
from .activations import gelu_new, gelu_new_reference, swiglu, swiglu_reference
from .attention import attention_reference, causal_attention
from .cross_entropy import (cross_entropy, cross_entropy_reference, lm_head_cross_entropy,
                            lm_head_cross_entropy_reference)
from .embedding import embedding, token_position_embedding
from .linear import accumulation_window, linear, pending_weight_grads
from .mlp import gelu_mlp
from .norm import (fused_add_layer_norm, fused_add_rms_norm, layer_norm, layer_norm_reference, rms_norm,
                   rms_norm_reference)

__all__ = [
    "linear", "accumulation_window", "pending_weight_grads", "gelu_mlp", "embedding", "token_position_embedding", "gelu_new", "gelu_new_reference", "swiglu", "swiglu_reference", "attention_reference",
    "causal_attention", "cross_entropy", "cross_entropy_reference", "lm_head_cross_entropy",
    "lm_head_cross_entropy_reference", "fused_add_layer_norm", "fused_add_rms_norm", "layer_norm",
    "layer_norm_reference", "rms_norm", "rms_norm_reference",
]
