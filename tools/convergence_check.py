"""End-to-end convergence check: the same GPT-2 (6 layers of GPT-2 small by default) trained on
the same learnable synthetic stream three ways —

  hip-bf16   every HIP kernel of this package (the path bench.py measures)
  ref-bf16   the PyTorch reference ops in bf16 (PDA_FORCE_REFERENCE=1)
  hip-accum  the same kernels with every batch split into ``--accum`` micro-batches inside an
             accumulation window (weight gradients formed once per window from the kept operands)
  ref-fp32   the PyTorch reference ops in fp32 (the reference repo's precision)

and prints the loss curves side by side. The data is a first-order Markov chain over 128 token
ids in which every token has four equally likely successors, so the achievable loss is ln 4 = 1.386
and a model that trains correctly drops from ln V = 10.8 through the unigram plateau (ln 128 = 4.85)
towards it within a few hundred steps. The
three runs see identical batches and start from identical weights; the check is that the HIP
run leaves the plateau about when the references do and ends at the same loss.

usage: python tools/convergence_check.py [--steps 300] [--layers 6] [--batch 16] [--seq-len 512]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd.models import LlamaForCausalLM, MyGPT2LMHeadModel, get_config  # noqa: E402
from pytorch_distributed_amd.optim import CosineAnnealingLR, FusedAdamW  # noqa: E402


def markov_batches(steps, batch, seq_len, device, seed=7, states=1024, fanout=4):
    g = torch.Generator(device=device).manual_seed(seed)
    succ = torch.randint(0, states, (states, fanout), device=device, generator=g)
    out = []
    for _ in range(steps):
        x = torch.empty(batch, seq_len + 1, dtype=torch.long, device=device)
        x[:, 0] = torch.randint(0, states, (batch,), device=device, generator=g)
        choice = torch.randint(0, fanout, (batch, seq_len), device=device, generator=g)
        for t in range(seq_len):
            x[:, t + 1] = succ[x[:, t], choice[:, t]]
        out.append(x)
    return out


def train(name, cls, cfg, init, batches, lr, dtype, reference, accum=1):
    if reference:
        os.environ["PDA_FORCE_REFERENCE"] = "1"
    else:
        os.environ.pop("PDA_FORCE_REFERENCE", None)
    model = cls(cfg).to("cuda")
    model.load_state_dict(init)
    model = model.to(dtype)
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=0.1)
    sched = CosineAnnealingLR(opt, T_max=len(batches), eta_min=0.1 * lr)
    losses = []
    from pytorch_distributed_amd import ops
    for x in batches:
        if accum == 1:
            loss = model(x[:, :-1], x[:, 1:])
            loss.backward()
        else:
            loss = 0.0
            for i, part in enumerate(x.chunk(accum)):
                with ops.accumulation_window(i, accum):
                    piece = model(part[:, :-1], part[:, 1:]) / accum
                    piece.backward()
                loss = loss + piece.detach()
        opt.step(zero_grad=True)
        sched.step()
        losses.append(loss.detach() if accum == 1 else loss)
    os.environ.pop("PDA_FORCE_REFERENCE", None)
    return name, torch.stack(losses).float().cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--lr", type=float, default=6e-4)
    ap.add_argument("--accum", type=int, default=4, help="micro-batches per step of the hip-accum run")
    ap.add_argument("--states", type=int, default=128, help="token ids the Markov chain visits")
    ap.add_argument("--arch", choices=["gpt2", "llama"], default="gpt2",
                    help="llama: RMSNorm / rotary / SwiGLU blocks, d = 512, four heads of 128")
    a = ap.parse_args()
    if a.arch == "gpt2":
        cls, cfg = MyGPT2LMHeadModel, get_config("gpt2", n_layer=a.layers).without_dropout()
    else:
        cls = LlamaForCausalLM
        cfg = get_config("llama-tiny", n_layer=a.layers, n_embd=512, n_head=4, intermediate_size=1408, vocab_size=32000,
                         n_ctx=a.seq_len).without_dropout()
    torch.manual_seed(0)
    init = {k: v.clone() for k, v in cls(cfg).state_dict().items()}
    batches = markov_batches(a.steps, a.batch, a.seq_len, "cuda", states=a.states)
    runs = [train("hip-bf16", cls, cfg, init, batches, a.lr, torch.bfloat16, False),
            train("hip-accum", cls, cfg, init, batches, a.lr, torch.bfloat16, False, accum=a.accum),
            train("ref-bf16", cls, cfg, init, batches, a.lr, torch.bfloat16, True),
            train("ref-fp32", cls, cfg, init, batches, a.lr, torch.float32, True)]
    print(f"{a.arch} {a.layers} layers d={cfg.n_embd}, batch {a.batch} x {a.seq_len}, {a.steps} steps, "
          f"unigram plateau ln {a.states} = {math.log(a.states):.3f}, floor ln 4 = {math.log(4):.3f}")
    print(f"{'step':>6s} " + " ".join(f"{n:>10s}" for n, _ in runs))
    win = 10
    for s in list(range(0, a.steps, max(1, a.steps // 15))) + [a.steps - 1]:
        lo = max(0, s - win + 1)
        print(f"{s:6d} " + " ".join(f"{c[lo:s + 1].mean().item():10.4f}" for _, c in runs))
    tail = {n: c[-20:].mean().item() for n, c in runs}
    print("mean of the last 20 steps: " + ", ".join(f"{n} {v:.4f}" for n, v in tail.items()))
    # Leaving the unigram plateau is a phase transition whose timing is chaotic (the two reference
    # runs differ from each other by tens of percent mid-way), so curves are compared by when they
    # cross a threshold and by where they end, not point by point.
    def first_below(c, thr):
        idx = (c < thr).nonzero()
        return int(idx[0]) if idx.numel() else None
    cross = {n: first_below(c, 2.0) for n, c in runs}
    print("first step with loss < 2.0: " + ", ".join(f"{n} {v}" for n, v in cross.items()))
    floor = math.log(4)
    ref_tail = min(tail["ref-bf16"], tail["ref-fp32"])
    ref_cross = [v for k, v in cross.items() if k.startswith("ref-") and v is not None]
    # Pass: each HIP run ends within 2 % of the better reference and 5 % of the entropy floor, and
    # leaves the plateau no later than 1.5x the slower reference.
    ok = all(tail[n] < 1.02 * ref_tail and tail[n] < 1.05 * floor and cross[n] is not None
             and bool(ref_cross) and cross[n] <= 1.5 * max(ref_cross) for n in ("hip-bf16", "hip-accum"))
    print("CONVERGENCE", "OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
