"""FLOP accounting and MFU.

Convention: training FLOPs per token = ``6 * N_mm + 12 * L * d * T`` where ``N_mm`` is
the number of matmul weights (every Linear weight incl. the tied LM head; embeddings
tables other than the head, biases and norms excluded), the attention term is not
discounted for causality, and recomputation is not counted (model FLOPs, not hardware
FLOPs). MFU divides by the MI355X dense bf16 peak, ``MI355X_PEAK_BF16_FLOPS``.
"""
from __future__ import annotations

from .. import MI355X_PEAK_BF16_FLOPS


def gpt_flops_per_token(n_layer: int, d_model: int, vocab: int, seq_len: int,
                        ffn_mult_params: int = 8) -> float:
    """GPT-2 style block: 4 d^2 attention + 8 d^2 MLP matmul weights per layer."""
    n_mm = n_layer * (4 + ffn_mult_params) * d_model * d_model + d_model * vocab
    return 6.0 * n_mm + 12.0 * n_layer * d_model * seq_len


def model_flops_per_token(model, seq_len: int) -> float:
    """FLOPs per trained token of a model carrying a known ``config``; 0.0 if unknown."""
    cfg = getattr(model, "config", None)
    if cfg is None:
        return 0.0
    if hasattr(cfg, "intermediate_size"):  # Llama-style: 4 d^2 + 3 d f per layer, untied head
        n_mm = cfg.n_layer * (4 * cfg.n_embd ** 2 + 3 * cfg.n_embd * cfg.intermediate_size) \
            + cfg.n_embd * cfg.vocab_size
        return 6.0 * n_mm + 12.0 * cfg.n_layer * cfg.n_embd * seq_len
    return gpt_flops_per_token(cfg.n_layer, cfg.n_embd, cfg.vocab_size, seq_len)


def mfu(tokens_per_second: float, flops_per_token: float, n_gpus: int = 1,
        peak_flops: float = MI355X_PEAK_BF16_FLOPS) -> float:
    return tokens_per_second * flops_per_token / (n_gpus * peak_flops)
