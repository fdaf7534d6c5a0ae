"""Offline model configurations.

The reference obtains its configuration from the Hugging Face hub
(``AutoConfig.from_pretrained("gpt2-large")``,
assignments/assignment1/train_baseline.py:24) and reads ``n_embd, n_head, n_ctx,
n_layer, vocab_size, attn_pdrop, resid_pdrop, embd_pdrop, layer_norm_epsilon,
activation_function`` from it (assignments/assignment1/model/my_gpt2.py:16-29,
90,112,148-155). There is no network here, so the four GPT-2 family
configurations are built in with the hub's values. Any object exposing those
attributes (e.g. a ``transformers.GPT2Config``) is accepted by the models too;
``n_ctx`` falls back to ``n_positions`` because recent ``transformers``
releases no longer define ``n_ctx``.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, replace
from typing import Any, Dict


@dataclass
class GPT2Config:
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 768
    vocab_size: int = 50257
    n_ctx: int = 1024
    activation_function: str = "gelu_new"
    attn_pdrop: float = 0.1
    resid_pdrop: float = 0.1
    embd_pdrop: float = 0.1
    layer_norm_epsilon: float = 1e-5
    name: str = "gpt2"

    @property
    def n_positions(self) -> int:
        return self.n_ctx

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head

    def without_dropout(self) -> "GPT2Config":
        return replace(self, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


@dataclass
class LlamaConfig:
    """Llama-style decoder: RMSNorm, rotary positions, SwiGLU MLP, no biases, untied head."""
    n_layer: int = 32
    n_head: int = 32
    n_embd: int = 4096
    intermediate_size: int = 11008
    vocab_size: int = 32000
    n_ctx: int = 2048
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    tie_word_embeddings: bool = False
    name: str = "llama-7b"

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    def without_dropout(self) -> "LlamaConfig":
        """Llama-style models have no dropout; present so callers can treat both families alike."""
        return self


_GPT2 = {
    "gpt2": dict(n_layer=12, n_head=12, n_embd=768),
    "gpt2-medium": dict(n_layer=24, n_head=16, n_embd=1024),
    "gpt2-large": dict(n_layer=36, n_head=20, n_embd=1280),
    "gpt2-xl": dict(n_layer=48, n_head=25, n_embd=1600),
    # Tiny shapes for CPU plumbing tests (head_dim stays 64 like every GPT-2).
    "gpt2-tiny": dict(n_layer=2, n_head=2, n_embd=128, vocab_size=512, n_ctx=128),
    "gpt2-mini": dict(n_layer=4, n_head=4, n_embd=256, vocab_size=2048, n_ctx=256),
}

_LLAMA = {
    "llama-7b": dict(n_layer=32, n_head=32, n_embd=4096, intermediate_size=11008),
    "llama-1b": dict(n_layer=16, n_head=16, n_embd=2048, intermediate_size=5632),
    "llama-tiny": dict(n_layer=2, n_head=2, n_embd=256, intermediate_size=512, vocab_size=512, n_ctx=128),
}


def get_config(name: str, **overrides):
    """Built-in configuration by name (``gpt2``, ``gpt2-medium``, ``gpt2-large``, ``gpt2-xl``,
    ``llama-7b``, ... plus tiny test shapes); keyword overrides replace individual fields."""
    if name in _GPT2:
        return GPT2Config(name=name, **{**_GPT2[name], **overrides})
    if name in _LLAMA:
        return LlamaConfig(name=name, **{**_LLAMA[name], **overrides})
    raise KeyError(f"unknown model config {name!r}; known: {sorted(_GPT2) + sorted(_LLAMA)}")


def normalize_gpt2_config(config) -> GPT2Config:
    """Accepts a GPT2Config, a name, or any HF-style object carrying the GPT-2 fields."""
    if isinstance(config, GPT2Config):
        return config
    if isinstance(config, str):
        return get_config(config)
    n_ctx = getattr(config, "n_ctx", None) or getattr(config, "n_positions", None) or 1024
    return GPT2Config(
        n_layer=config.n_layer, n_head=config.n_head, n_embd=config.n_embd, vocab_size=config.vocab_size,
        n_ctx=n_ctx, activation_function=getattr(config, "activation_function", "gelu_new"),
        attn_pdrop=getattr(config, "attn_pdrop", 0.1), resid_pdrop=getattr(config, "resid_pdrop", 0.1),
        embd_pdrop=getattr(config, "embd_pdrop", 0.1),
        layer_norm_epsilon=getattr(config, "layer_norm_epsilon", 1e-5),
        name=getattr(config, "name_or_path", None) or "gpt2-custom")


def gpt2_param_count(cfg: GPT2Config) -> int:
    """Closed-form parameter count (tied LM head counted once): L(12d^2 + 13d) + (V + n_ctx + 2)d."""
    d = cfg.n_embd
    return cfg.n_layer * (12 * d * d + 13 * d) + (cfg.vocab_size + cfg.n_ctx + 2) * d
