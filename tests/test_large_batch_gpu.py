"""Tensors beyond 2^31 elements / 4 GiB: 64 sequences of 1024 tokens against the GPT-2 vocabulary make a
65,536 x 50,304 bf16 logits buffer (3.3 G elements, 6.6 GB). One pass over the whole batch must agree with the
same batch taken as two halves of 32 sequences (sizes every other test and the convergence runs cover)."""
import pytest
import torch

from pytorch_distributed_amd import _ext
from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_batch_of_64_sequences_matches_two_halves():
    _ext.require()
    if torch.cuda.get_device_properties(0).total_memory < 64 * 2**30:
        pytest.skip("needs a 64 GB device")
    cfg = get_config("gpt2-large", n_layer=2).without_dropout()
    torch.manual_seed(0)
    model = MyGPT2LMHeadModel(cfg).to(device=DEV, dtype=torch.bfloat16)
    ids = torch.randint(0, cfg.vocab_size, (64, 1025), device=DEV)

    def run(chunks):
        model.zero_grad(set_to_none=True)
        total = 0.0
        for part in ids.chunk(chunks):
            loss = model(part[:, :-1].contiguous(), part[:, 1:].contiguous()) / chunks
            loss.backward()
            total += loss.item()
        return total, {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}

    loss_whole, grads_whole = run(1)
    loss_halves, grads_halves = run(2)
    assert abs(loss_whole - loss_halves) < 2e-3 * abs(loss_halves), (loss_whole, loss_halves)
    for name, g in grads_whole.items():
        ref = grads_halves[name]
        err = ((g - ref).norm() / (ref.norm() + 1e-12)).item()
        assert err < 2e-2, (name, err)
    # the rows that live past the 4 GiB mark of the logits buffer: their targets' embedding rows get gradient
    # only through the LM head and the embedding of the LAST sequences
    last = ids[-1, :-1].unique()
    g = grads_whole["transformer.wte.weight"][last]
    ref = grads_halves["transformer.wte.weight"][last]
    assert ((g - ref).norm() / ref.norm()).item() < 2e-2
