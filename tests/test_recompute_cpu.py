"""Activation-memory policies on the CPU: "selective" (element-wise activations rebuilt in backward
by the consuming op), "full" (own block-recompute engine) and "none" give the same numbers, and
"selective" really keeps less."""
import copy

import pytest
import torch

from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config
from pytorch_distributed_amd.models.llama import LlamaForCausalLM
from pytorch_distributed_amd.models.config import LlamaConfig
from pytorch_distributed_amd.utils.recompute import recompute

from dist_utils import run_distributed


def _tiny(mode, seed=0, **overrides):
    torch.manual_seed(seed)
    cfg = get_config("gpt2-tiny", **overrides)
    return MyGPT2LMHeadModel(cfg, activation_checkpoint=mode)


def _loss_and_grads(model, ids, seed=None):
    model.train()
    model.zero_grad(set_to_none=True)
    if seed is not None:
        torch.manual_seed(seed)
    loss = model(ids[:, :-1], ids[:, 1:])
    loss.backward()
    return loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("mode", ["selective", "full"])
def test_policies_match_keep_everything(mode):
    ids = torch.randint(0, 128, (2, 17))
    base = _tiny("none", attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
    other = _tiny(mode, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
    other.load_state_dict(base.state_dict())
    l0, g0 = _loss_and_grads(base, ids)
    l1, g1 = _loss_and_grads(other, ids)
    assert torch.allclose(l0, l1, atol=1e-6)
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=1e-5, rtol=1e-4), n


def test_full_recompute_reproduces_dropout_masks():
    ids = torch.randint(0, 128, (2, 17))
    base = _tiny("none", attn_pdrop=0.1, resid_pdrop=0.1, embd_pdrop=0.1)
    full = _tiny("full", attn_pdrop=0.1, resid_pdrop=0.1, embd_pdrop=0.1)
    full.load_state_dict(base.state_dict())
    l0, g0 = _loss_and_grads(base, ids, seed=11)
    l1, g1 = _loss_and_grads(full, ids, seed=11)
    assert torch.allclose(l0, l1, atol=1e-6)
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=1e-5, rtol=1e-4), n
    # and the generator ends where a run without recomputation ends
    torch.manual_seed(11)
    base(ids[:, :-1], ids[:, 1:]).backward()
    after_plain = torch.get_rng_state()
    torch.manual_seed(11)
    full(ids[:, :-1], ids[:, 1:]).backward()
    assert torch.equal(after_plain, torch.get_rng_state())


def _kept_bytes(model, ids):
    """Bytes of distinct storages autograd keeps between forward and backward."""
    seen = {}

    def pack(t):
        if t.numel() > 0:
            seen[t.untyped_storage().data_ptr()] = t.untyped_storage().nbytes()
        return t

    model.train()
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        loss = model(ids[:, :-1], ids[:, 1:])
    params = {p.untyped_storage().data_ptr() for p in model.parameters()}
    loss.backward()
    return sum(n for ptr, n in seen.items() if ptr not in params)


def test_selective_keeps_less_than_none_and_full_least():
    ids = torch.randint(0, 128, (4, 33))
    kept = {}
    for mode in ("none", "selective", "full"):
        kept[mode] = _kept_bytes(_tiny(mode, n_layer=4, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0), ids)
    cfg = get_config("gpt2-tiny")
    tokens, d, layers, bytes_per = 4 * 32, cfg.n_embd, 4, 4
    # per block "selective" drops both LayerNorm outputs (d each) and gelu(u) (4d)
    assert kept["none"] - kept["selective"] == layers * tokens * 6 * d * bytes_per, kept
    assert kept["full"] < kept["selective"] // 2, kept


def test_llama_selective_matches_none():
    cfg = LlamaConfig(vocab_size=128, n_ctx=64, n_embd=64, n_layer=2, n_head=2, intermediate_size=128)
    ids = torch.randint(0, 128, (2, 17))
    torch.manual_seed(0)
    base = LlamaForCausalLM(cfg, activation_checkpoint="none")
    outs = {}
    for mode in ("none", "selective", "full"):
        m = LlamaForCausalLM(cfg, activation_checkpoint=mode)
        m.load_state_dict(base.state_dict())
        outs[mode] = _loss_and_grads(m, ids)
    for mode in ("selective", "full"):
        assert torch.allclose(outs["none"][0], outs[mode][0], atol=1e-6)
        for n in outs["none"][1]:
            assert torch.allclose(outs["none"][1][n], outs[mode][1][n], atol=1e-5, rtol=1e-4), (mode, n)


def test_recompute_function_contract():
    """Non-tensor and None arguments, an argument handed straight through, an argument that needs no
    gradient, and parameters reached only through the closure."""
    lin = torch.nn.Linear(8, 8)
    x = torch.randn(4, 8, requires_grad=True)
    frozen = torch.randn(4, 8)

    def fn(a, b, nothing, scale, *, shift):
        assert nothing is None
        return lin(a) * scale + b + shift, a

    y, passthrough = recompute(fn, x, frozen, None, 2.0, shift=1.0)
    (y.sum() + 3 * passthrough.sum()).backward()
    gx, gw = x.grad.clone(), lin.weight.grad.clone()
    x.grad = None
    lin.zero_grad()
    y2, p2 = fn(x, frozen, None, 2.0, shift=1.0)
    (y2.sum() + 3 * p2.sum()).backward()
    assert torch.allclose(gx, x.grad) and torch.allclose(gw, lin.weight.grad)
    # no argument requires grad: parameters inside still receive gradients
    lin.zero_grad()
    recompute(lambda a: lin(a), frozen).sum().backward()
    assert lin.weight.grad is not None and lin.weight.grad.abs().sum() > 0


def _ddp_full_worker(rank, world):
    from pytorch_distributed_amd.parallel import DDP

    torch.manual_seed(0)
    cfg = get_config("gpt2-tiny", n_layer=3, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
    ref = MyGPT2LMHeadModel(cfg, activation_checkpoint="none")
    model = MyGPT2LMHeadModel(cfg, activation_checkpoint="full")
    model.load_state_dict(ref.state_dict())
    ddp = DDP(model, bucket_cap_mb=0.05, first_bucket_mb=0.02)
    g = torch.Generator().manual_seed(5)
    all_ids = torch.randint(0, cfg.vocab_size, (2 * world, 17), generator=g)
    for step in range(2):  # the second pass checks that the first one was closed properly
        ddp.zero_grad()
        mine = all_ids[2 * rank:2 * rank + 2]
        ddp.train()
        ddp(mine[:, :-1], mine[:, 1:]).backward()
        ref.zero_grad(set_to_none=True)
        ref.train()
        for r in range(world):
            part = all_ids[2 * r:2 * r + 2]
            (ref(part[:, :-1], part[:, 1:]) / world).backward()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-5, rtol=1e-4), (step, n)
        assert ddp.reducer.num_buckets > 2
        assert ddp.reducer.num_collectives == (step + 1) * ddp.reducer.num_buckets


def test_ddp_with_full_recompute_reduces_once_per_bucket_at_the_models_end():
    run_distributed(_ddp_full_worker, 2)
