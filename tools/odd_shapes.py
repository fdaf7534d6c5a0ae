"""Odd shapes through the whole model on one GPU: loss and gradients of the HIP path against the PyTorch reference
ops (PDA_FORCE_REFERENCE=1) of the same bf16 model, for batch sizes and sequence lengths that are not multiples of
the kernels' tiles. usage: python tools/odd_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd.models import LlamaForCausalLM, MyGPT2LMHeadModel, get_config  # noqa: E402

dev = torch.device("cuda:0")


def run(model, ids, reference):
    if reference:
        os.environ["PDA_FORCE_REFERENCE"] = "1"
    else:
        os.environ.pop("PDA_FORCE_REFERENCE", None)
    for p in model.parameters():
        p.grad = None
    loss = model(ids[:, :-1].contiguous(), ids[:, 1:].contiguous())
    loss.backward()
    os.environ.pop("PDA_FORCE_REFERENCE", None)
    return loss.item(), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}


worst = 0.0
for arch in ("gpt2", "llama"):
    for policy in ("none", "selective", "full"):
        torch.manual_seed(0)
        if arch == "gpt2":
            cfg = get_config("gpt2", n_layer=2).without_dropout()
            model = MyGPT2LMHeadModel(cfg, activation_checkpoint=policy)
        else:
            cfg = get_config("llama-tiny", n_layer=2, n_embd=256, n_head=2, intermediate_size=704, vocab_size=1000,
                             n_ctx=1100).without_dropout()
            model = LlamaForCausalLM(cfg, activation_checkpoint=policy)
        model = model.to(device=dev, dtype=torch.bfloat16).train()
        for B, T in ((1, 17), (3, 100), (2, 1000), (1, 1023), (5, 64), (7, 33)):
            if T > getattr(cfg, "n_ctx", 1024):
                continue
            ids = torch.randint(0, cfg.vocab_size, (B, T + 1), device=dev)
            l_hip, g_hip = run(model, ids, False)
            l_ref, g_ref = run(model, ids, True)
            assert g_hip.keys() == g_ref.keys()
            rel = max(((g_hip[k] - g_ref[k]).norm() / (g_ref[k].norm() + 1e-6)).item() for k in g_ref)
            worst = max(worst, rel)
            flag = "" if abs(l_hip - l_ref) < 3e-2 and rel < 6e-2 else "  <-- MISMATCH"
            print(f"{arch:6s} {policy:9s} B={B} T={T:5d}: loss {l_hip:.4f} vs {l_ref:.4f}, worst relative gradient error {rel:.4f}{flag}")
            assert not flag
print(f"odd shapes ok (worst relative gradient error {worst:.4f})")
