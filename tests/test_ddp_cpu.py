"""DDP on the native reducer, CPU processes on the native socket communicator."""
import copy

import pytest
import torch

from tests.dist_utils import run_distributed


def _tiny_model(seed=0):
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

    torch.manual_seed(seed)
    return MyGPT2LMHeadModel(get_config("gpt2-tiny").without_dropout())


def _batch(seed, n, T=32, vocab=512):
    g = torch.Generator().manual_seed(seed)
    buf = torch.randint(0, vocab, (n, T + 1), generator=g)
    return buf[:, :-1].contiguous(), buf[:, 1:].contiguous()


def _ddp_matches_single_process(rank, world, accum):
    from pytorch_distributed_amd.parallel import DDP

    ref = _tiny_model(0)
    model = copy.deepcopy(ref)
    if rank != 0:  # the constructor must overwrite this with rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    ddp = DDP(model, bucket_cap_mb=0.25, first_bucket_mb=0.05)
    assert ddp.reducer.num_buckets > 2
    per_rank = 2
    x, y = _batch(123, per_rank * world * accum)
    # single process: the whole global batch, mean loss
    ref.zero_grad()
    ref(x, y).backward()
    # DDP: each rank its slice of each micro-batch
    before = ddp.reducer.num_collectives
    for a in range(accum):
        lo = (a * world + rank) * per_rank
        xs, ys = x[lo:lo + per_rank], y[lo:lo + per_rank]
        if a < accum - 1:
            with ddp.no_sync():
                (ddp(xs, ys) / accum).backward()
            assert ddp.reducer.num_collectives == before, "no_sync must not communicate"
        else:
            (ddp(xs, ys) / accum).backward()
    assert ddp.reducer.num_collectives == before + ddp.reducer.num_buckets
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        torch.testing.assert_close(p.grad, q.grad, rtol=2e-4, atol=2e-6, msg=lambda m: f"{n}: {m}")
    # gradients are views of the reducer's buckets
    ptrs = {b.data_ptr(): b.numel() * b.element_size() for b in ddp.reducer.bucket_tensors()}
    for p in model.parameters():
        assert any(start <= p.grad.data_ptr() < start + size for start, size in ptrs.items())


@pytest.mark.parametrize("accum", [1, 3])
def test_ddp_grads_match_single_process(accum):
    run_distributed(_ddp_matches_single_process, world=2, args=(accum,))


def _ddp_training_matches(rank, world):
    """A few optimizer steps: DDP(world=2) == single process on the concatenated batch."""
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import DDP

    ref = _tiny_model(1)
    model = copy.deepcopy(ref)
    ddp = DDP(model, bucket_cap_mb=0.5)
    opt_ref = FusedAdamW(ref.parameters(), lr=1e-3, weight_decay=0.1)
    opt = FusedAdamW(ddp.parameters(), lr=1e-3, weight_decay=0.1)
    for step in range(4):
        x, y = _batch(1000 + step, 2 * world)
        ref(x, y).backward()
        opt_ref.step(zero_grad=True)
        ddp(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()
        opt.step(zero_grad=True)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        # Adam normalises tiny gradients to O(lr) updates, so fp32 summation-order noise shows at ~lr/50
        torch.testing.assert_close(p, q, rtol=1e-3, atol=1e-4, msg=lambda m: f"{n}: {m}")
    # state_dict keys are the bare model's
    assert list(ddp.state_dict().keys()) == list(ref.state_dict().keys())


def test_ddp_training_matches_single_process():
    run_distributed(_ddp_training_matches, world=2)


def _ddp_set_to_none(rank, world):
    """zero_grad(set_to_none=True) between steps must not break the bucket views."""
    from pytorch_distributed_amd.parallel import DDP

    model = _tiny_model(2)
    ref = copy.deepcopy(model)
    ddp = DDP(model, bucket_cap_mb=0.25)
    x, y = _batch(7, 2 * world)
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        ddp(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()
    ref(x, y).backward()
    assert ddp.reducer.num_rebound_grads > 0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=2e-4, atol=2e-6, msg=lambda m: f"{n}: {m}")


def test_ddp_survives_set_to_none():
    run_distributed(_ddp_set_to_none, world=2)


def _comm_debug_modes(rank, world):
    """PDA_COMM_DEBUG / PDA_COMM_SERIALIZE: same gradients with the cross-rank agreement check and
    with every collective waited for at once; a rank that is about to issue a different collective
    (here: different bucket boundaries) gets an error instead of a hang."""
    import os

    os.environ["PDA_COMM_DEBUG"] = "1"
    os.environ["PDA_COMM_SERIALIZE"] = "1"
    from pytorch_distributed_amd.parallel import DDP, FSDP, ShardingStrategy

    ref = _tiny_model(3)
    x, y = _batch(11, 2 * world)
    ref(x, y).backward()
    model = copy.deepcopy(ref)
    model.zero_grad()
    ddp = DDP(model, bucket_cap_mb=0.25)
    ddp(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=2e-4, atol=2e-6, msg=lambda m: f"{n}: {m}")

    sharded = FSDP(copy.deepcopy(ref).requires_grad_(True), sharding_strategy=ShardingStrategy.FULL_SHARD)
    sharded(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()

    # mismatch: rank 1 builds different buckets -> its first all-reduce has a different size
    bad = DDP(copy.deepcopy(ref), bucket_cap_mb=0.25 if rank == 0 else 0.11, first_bucket_mb=0.05 if rank == 0 else 0.02)
    with pytest.raises(RuntimeError, match="collective mismatch"):
        bad(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()


def test_comm_debug_modes():
    run_distributed(_comm_debug_modes, world=2)


def _reduce_fp32(rank, world, kind):
    """bf16 model, gradients reduced through an fp32 copy: the averaged gradient must equal the
    fp32 mean of the two ranks' bf16 gradients rounded ONCE (a bf16 reduction rounds the sum too),
    and the fp32-reduced run must agree with the default one to bf16 accuracy."""
    from pytorch_distributed_amd.parallel import DDP, FSDP, ShardingStrategy, all_reduce

    def grads(reduce_fp32):
        model = _tiny_model(0).to(torch.bfloat16)
        if kind == "ddp":
            wrapped = DDP(model, bucket_cap_mb=0.25, first_bucket_mb=0.05, reduce_fp32=reduce_fp32)
        else:
            wrapped = FSDP(model, sharding_strategy=ShardingStrategy[kind], reduce_fp32=reduce_fp32)
        x, y = _batch(321, 2 * world)
        wrapped(x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]).backward()
        if kind == "ddp":
            return torch.cat([p.grad.detach().float().reshape(-1) for p in model.parameters()])
        return torch.cat([p.grad.detach().float().reshape(-1) for p in wrapped.parameters()])

    default, wide = grads(False), grads(True)
    assert default.shape == wide.shape and torch.isfinite(wide).all()
    rel = ((default - wide).norm() / wide.norm()).item()
    assert rel < 2e-2, rel
    if kind == "ddp":
        # exact statement of the option: local bf16 gradients, averaged in fp32, rounded once
        local = _tiny_model(0).to(torch.bfloat16)
        x, y = _batch(321, 2 * world)
        local(x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]).backward()
        mine = torch.cat([p.grad.detach().float().reshape(-1) for p in local.parameters()])
        total = mine.clone()
        all_reduce(total)
        expect = (total / world).to(torch.bfloat16).float()
        assert torch.equal(wide, expect)


@pytest.mark.parametrize("kind", ["ddp", "FULL_SHARD", "NO_SHARD"])
def test_fp32_gradient_reduction(kind):
    run_distributed(_reduce_fp32, world=2, args=(kind,))
