"""What the reference's own recipe achieves on this GPU: GPT-2 Large built from stock PyTorch
modules and trained the way assignments/assignment1/train_baseline.py:24-31,61-64 does it
(global batch 32 as 4 micro-batches of 8, T = 1024, torch.optim.AdamW lr 3e-4 wd 0.1), with the
op sequence of model/my_gpt2.py:60-77 (scores materialised, masked_fill, softmax, matmul) and the
tanh GELU as separate element-wise ops.

The reference itself needs the Hugging Face hub for its configuration and cannot run offline, so
this script rebuilds the same computation from `torch.nn` only. Variants:

  fp32-eager   the reference's precision and attention (its actual recipe)
  bf16-eager   the same graph with bf16 parameters (same precision as bench.py)
  bf16-sdpa    bf16 with torch's fused scaled_dot_product_attention (best stock-PyTorch baseline)

usage: python tools/reference_style_bench.py [variant ...] [--steps 5] [--warmup 2] [--micro-batch 8]
Prints one JSON line per variant (tokens/s for 32 sequences of 1024 tokens per optimizer step).
"""
import argparse
import json
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

L, D, H, V, T = 36, 1280, 20, 50257, 1024


class Attention(nn.Module):
    def __init__(self, sdpa):
        super().__init__()
        self.c_attn, self.c_proj, self.sdpa = nn.Linear(D, 3 * D), nn.Linear(D, D), sdpa
        self.register_buffer("mask", torch.tril(torch.ones(T, T, dtype=torch.bool)).view(1, 1, T, T), persistent=False)

    def forward(self, x):
        B, t, _ = x.shape
        q, k, v = (z.view(B, t, H, D // H).transpose(1, 2) for z in self.c_attn(x).split(D, dim=2))
        if self.sdpa:
            y = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        else:
            att = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(D // H)
            att = att.masked_fill_(~self.mask[:, :, :t, :t], float("-inf"))
            y = torch.matmul(F.softmax(att, dim=-1), v)
        return self.c_proj(y.transpose(1, 2).reshape(B, t, D))


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class Block(nn.Module):
    def __init__(self, sdpa):
        super().__init__()
        self.ln_1, self.ln_2, self.attn = nn.LayerNorm(D), nn.LayerNorm(D), Attention(sdpa)
        self.c_fc, self.c_proj = nn.Linear(D, 4 * D), nn.Linear(4 * D, D)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.c_proj(gelu_new(self.c_fc(self.ln_2(x))))


class GPT2(nn.Module):
    def __init__(self, sdpa):
        super().__init__()
        self.wte, self.wpe = nn.Embedding(V, D), nn.Embedding(T, D)
        self.h, self.ln_f = nn.ModuleList(Block(sdpa) for _ in range(L)), nn.LayerNorm(D)
        self.lm_head = nn.Linear(D, V, bias=False)
        self.lm_head.weight = self.wte.weight
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=0.02)

    def forward(self, ids):
        x = self.wte(ids) + self.wpe(torch.arange(ids.shape[1], device=ids.device))
        for blk in self.h:
            x = blk(x)
        return self.lm_head(self.ln_f(x))


def run(variant, steps, warmup, micro):
    dtype = torch.float32 if variant.startswith("fp32") else torch.bfloat16
    torch.manual_seed(42)
    model = GPT2(variant.endswith("sdpa")).to(device="cuda", dtype=dtype)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=0.1)
    accum = 32 // micro
    g = torch.Generator(device="cuda").manual_seed(1)

    def step():
        for _ in range(accum):
            ids = torch.randint(0, V, (micro, T + 1), device="cuda", generator=g)
            logits = model(ids[:, :-1])
            loss = F.cross_entropy(logits.view(-1, V), ids[:, 1:].reshape(-1))
            (loss / accum).backward()
        opt.step()
        opt.zero_grad()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"variant": variant, "micro_batch": micro, "ms_per_step": 1e3 * dt, "tokens_per_s": 32 * T / dt,
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30}), flush=True)
    del model, opt
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", default=["fp32-eager", "bf16-eager", "bf16-sdpa"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--micro-batch", type=int, default=8)
    a = ap.parse_args()
    for v in a.variants:
        run(v, a.steps, a.warmup, a.micro_batch)


if __name__ == "__main__":
    main()
