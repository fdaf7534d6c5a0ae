"""FSDP on the native flat-parameter runtime, gloo processes on the CPU."""
import copy

import pytest
import torch

from tests.dist_utils import run_distributed
from tests.test_ddp_cpu import _batch, _tiny_model


def _wrap(model, strategy, unit_blocks=1):
    from pytorch_distributed_amd.parallel import FSDP, ShardingStrategy, fsdp_wrap_gpt

    strategy = ShardingStrategy[strategy]
    if unit_blocks == 0:  # the reference's manual wrapping, spelled out
        for i, block in enumerate(model.transformer.h):
            model.transformer.h[i] = FSDP(block, sharding_strategy=strategy)
        return FSDP(model, sharding_strategy=strategy)
    return fsdp_wrap_gpt(model, strategy, unit_blocks=unit_blocks)


def _fsdp_grads_and_training(rank, world, strategy, accum, unit_blocks):
    from pytorch_distributed_amd.optim import FusedAdamW

    ref = _tiny_model(3)
    model = _wrap(copy.deepcopy(ref), strategy, unit_blocks)
    opt_ref = FusedAdamW(ref.parameters(), lr=1e-3, weight_decay=0.1)
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
    n_flat = len(list(model.parameters()))
    assert n_flat == len([m for m in model.modules() if type(m).__name__ == "FSDP"])
    per_rank = 2
    for step in range(3):
        x, y = _batch(500 + step, per_rank * world * accum)
        ref(x, y).backward()
        opt_ref.step(zero_grad=True)
        units = [m.unit for m in model.modules() if type(m).__name__ == "FSDP"]
        red_before = sum(u.num_reductions for u in units)
        for a in range(accum):
            lo = (a * world + rank) * per_rank
            xs, ys = x[lo:lo + per_rank], y[lo:lo + per_rank]
            if a < accum - 1:
                with model.no_sync():
                    (model(xs, ys) / accum).backward()
                assert sum(u.num_reductions for u in units) == red_before, "no_sync must not reduce"
            else:
                (model(xs, ys) / accum).backward()
        assert sum(u.num_reductions for u in units) == red_before + len(units)
        if strategy != "NO_SHARD":
            assert all(not u.is_unsharded() for u in units), "units must be resharded after a synced backward"
        opt.step(zero_grad=True)
    full = model.state_dict()
    ref_sd = ref.state_dict()
    assert list(full.keys()) == list(ref_sd.keys())
    for k in ref_sd:
        torch.testing.assert_close(full[k], ref_sd[k], rtol=1e-3, atol=1e-4, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("strategy", ["FULL_SHARD", "SHARD_GRAD_OP", "NO_SHARD"])
def test_fsdp_matches_single_process(strategy):
    run_distributed(_fsdp_grads_and_training, world=2, args=(strategy, 1, 0))


def test_fsdp_full_shard_grad_accumulation_and_grouped_units():
    run_distributed(_fsdp_grads_and_training, world=2, args=("FULL_SHARD", 2, 2))


def test_fsdp_three_ranks_padding():
    """world=3 does not divide the flat sizes: exercises shard padding."""
    run_distributed(_fsdp_grads_and_training, world=3, args=("FULL_SHARD", 1, 1))


def _fsdp_resident_budget(rank, world):
    """FULL_SHARD with a resident budget at two ranks: same training as without one, fewer all-gathers,
    everything resharded at the end of every backward; 'auto' agrees on a budget across the ranks
    (zero on CPU tensors) and behaves like plain FULL_SHARD."""
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt

    base = _tiny_model(6)
    plain = fsdp_wrap_gpt(copy.deepcopy(base), ShardingStrategy.FULL_SHARD)
    kept = fsdp_wrap_gpt(copy.deepcopy(base), ShardingStrategy.FULL_SHARD, resident_gb=1.0)
    auto = fsdp_wrap_gpt(copy.deepcopy(base), ShardingStrategy.FULL_SHARD, resident_gb="auto")
    opts = [FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.1) for m in (plain, kept, auto)]

    def gathers(m):
        return sum(u.unit.num_allgathers for u in m.modules() if type(u).__name__ == "FSDP")

    for step in range(2):
        x, y = _batch(700 + step, 2 * world)
        xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
        before = [gathers(m) for m in (plain, kept, auto)]
        for m, o in zip((plain, kept, auto), opts):
            m(xs, ys).backward()
            assert all(not u.unit.is_unsharded() for u in m.modules() if type(u).__name__ == "FSDP")
            o.step(zero_grad=True)
        n_blocks = len(base.transformer.h)
        assert kept.resident_units == n_blocks and auto.resident_units == 0
        # plain / auto: root + every block in forward, every block again in backward; kept: forward only
        assert gathers(plain) - before[0] == 1 + 2 * n_blocks
        assert gathers(auto) - before[2] == 1 + 2 * n_blocks
        assert gathers(kept) - before[1] == 1 + n_blocks
    for a, b in zip(plain.parameters(), kept.parameters()):
        torch.testing.assert_close(a, b, rtol=0, atol=0)
    for a, b in zip(plain.parameters(), auto.parameters()):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_fsdp_resident_budget_two_ranks():
    run_distributed(_fsdp_resident_budget, world=2)


def _fsdp_resident_window(rank, world):
    """Resident units stay gathered across the micro-batches of a no_sync window (their parameters cannot
    change before the window's last backward): one all-gather per unit and step instead of one per
    micro-batch and pass; same training; everything resharded once the synchronising backward is done."""
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt

    base = _tiny_model(8)
    plain = fsdp_wrap_gpt(copy.deepcopy(base), ShardingStrategy.FULL_SHARD)
    kept = fsdp_wrap_gpt(copy.deepcopy(base), ShardingStrategy.FULL_SHARD, resident_gb=1.0)
    opts = [FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.1) for m in (plain, kept)]
    units = lambda m: [u.unit for u in m.modules() if type(u).__name__ == "FSDP"]  # noqa: E731
    gathers = lambda m: sum(u.num_allgathers for u in units(m))  # noqa: E731
    n_blocks, accum = len(base.transformer.h), 3
    for step in range(2):
        x, y = _batch(900 + step, 2 * world * accum)
        before = [gathers(plain), gathers(kept)]
        for m, o in zip((plain, kept), opts):
            for a in range(accum):
                lo = (a * world + rank) * 2
                if a < accum - 1:
                    with m.no_sync():
                        (m(x[lo:lo + 2], y[lo:lo + 2]) / accum).backward()
                    if m is kept:  # blocks (resident) and root are still gathered between micro-batches
                        assert all(u.is_unsharded() for u in units(m))
                else:
                    (m(x[lo:lo + 2], y[lo:lo + 2]) / accum).backward()
            assert all(not u.is_unsharded() for u in units(m))
            o.step(zero_grad=True)
        # plain: every block twice per micro-batch, the root once per window (it never reshards inside one)
        assert gathers(plain) - before[0] == 1 + 2 * n_blocks * accum
        assert gathers(kept) - before[1] == 1 + n_blocks
    for a, b in zip(plain.parameters(), kept.parameters()):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_fsdp_resident_units_stay_gathered_through_a_no_sync_window():
    run_distributed(_fsdp_resident_window, world=2)


def _fsdp_state_dict_roundtrip(rank, world):
    ref = _tiny_model(4)
    model = _wrap(copy.deepcopy(ref), "FULL_SHARD")
    other = _tiny_model(5)  # different weights
    model.load_state_dict(other.state_dict())
    got = model.state_dict()
    for k, v in other.state_dict().items():
        torch.testing.assert_close(got[k], v, rtol=0, atol=0)
    # logits of the sharded model equal the bare model's
    x, _ = _batch(9, 2)
    other.eval()
    model.eval()
    with torch.no_grad():
        torch.testing.assert_close(model(x), other(x), rtol=1e-5, atol=1e-5)


def test_fsdp_state_dict_roundtrip():
    run_distributed(_fsdp_state_dict_roundtrip, world=2)


def _fsdp_unwritten_ranges(rank, world):
    """A unit whose flat gradient buffer is only partly written by the backward: one parameter
    gets no gradient at all, one has a length that leaves an alignment gap, and the tail is
    padding. Everything unwritten must reduce as zeros, on every step (the buffer is reused)."""
    import torch.nn as nn
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import FSDP, ShardingStrategy

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.used = nn.Linear(37, 11)            # 407 + 11 elements: neither is a multiple of the alignment
            self.unused = nn.Linear(5, 3)             # never called: no gradient
            self.scale = nn.Parameter(torch.ones(7))  # odd length

        def forward(self, x):
            return (self.used(x)[:, :7] * self.scale).sum()

    torch.manual_seed(0)
    ref = Net()
    model = FSDP(copy.deepcopy(ref), sharding_strategy=ShardingStrategy.FULL_SHARD)
    opt_ref = FusedAdamW(ref.parameters(), lr=1e-2, weight_decay=0.0)
    opt = FusedAdamW(model.parameters(), lr=1e-2, weight_decay=0.0)
    for step in range(3):
        g = torch.Generator().manual_seed(100 + step)
        x = torch.randn(4 * world, 37, generator=g)
        (ref(x) / world).backward()
        for p in ref.unused.parameters():  # the optimizer below needs a gradient for every parameter
            p.grad = torch.zeros_like(p)
        opt_ref.step(zero_grad=True)
        model(x[4 * rank:4 * rank + 4]).backward()
        opt.step(zero_grad=True)
    sd, ref_sd = model.state_dict(), ref.state_dict()
    for k in ref_sd:
        torch.testing.assert_close(sd[k], ref_sd[k], rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")


def test_fsdp_unwritten_gradient_ranges_reduce_as_zero():
    run_distributed(_fsdp_unwritten_ranges, world=2)
