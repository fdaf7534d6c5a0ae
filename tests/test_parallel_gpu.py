"""DDP / FSDP on one GPU over RCCL (world_size = 1, collectives forced on) — exercises the
native reducer / sharding runtime with real HIP streams, bf16 parameters and fp32 masters."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def pg():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    from pytorch_distributed_amd.parallel import init_distributed, shutdown

    init_distributed(backend="rccl")
    yield
    shutdown()


def _model(seed=0, n_layer=3):
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

    torch.manual_seed(seed)
    cfg = get_config("gpt2", n_layer=n_layer).without_dropout()
    return MyGPT2LMHeadModel(cfg).to(device=DEV, dtype=torch.bfloat16), cfg


def _train(model, params, ids, steps=4, accum=2, no_sync=None):
    from pytorch_distributed_amd.optim import FusedAdamW

    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.1)
    losses = []
    for s in range(steps):
        for a in range(accum):
            x = ids[s * accum + a]
            if no_sync is not None and a < accum - 1:
                with no_sync():
                    loss = model(x[:, :-1], x[:, 1:]) / accum
                    loss.backward()
            else:
                loss = model(x[:, :-1], x[:, 1:]) / accum
                loss.backward()
            losses.append(loss.item() * accum)
        opt.step(zero_grad=True)
    torch.cuda.synchronize()
    return losses


def _ids(cfg, n, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return [torch.randint(0, cfg.vocab_size, (2, 257), device=DEV, generator=g) for _ in range(n)]


@pytest.mark.parametrize("reduce_fp32", [False, True])
def test_ddp_single_gpu_rccl_matches_plain(pg, reduce_fp32):
    from pytorch_distributed_amd.parallel import DDP

    base, cfg = _model(0)
    ids = _ids(cfg, 8)
    ref_losses = _train(base, base.parameters(), ids)
    model, _ = _model(0)
    ddp = DDP(model, device_ids=[0], bucket_cap_mb=8, first_bucket_mb=1, force_collectives=True,
              reduce_fp32=reduce_fp32)
    losses = _train(ddp, ddp.parameters(), ids, no_sync=ddp.no_sync)
    assert ddp.reducer.num_collectives == 4 * ddp.reducer.num_buckets
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-2, (losses, ref_losses)
    for (n, p), (_, q) in zip(model.named_parameters(), base.named_parameters()):
        assert (p.float() - q.float()).abs().max().item() < 1e-2, n


@pytest.mark.parametrize("strategy,reduce_fp32", [("FULL_SHARD", False), ("SHARD_GRAD_OP", False), ("NO_SHARD", False),
                                                  ("FULL_SHARD", True), ("NO_SHARD", True)])
def test_fsdp_single_gpu_rccl_matches_plain(pg, strategy, reduce_fp32):
    from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt

    base, cfg = _model(1)
    ids = _ids(cfg, 8, seed=1)
    ref_losses = _train(base, base.parameters(), ids)
    model, _ = _model(1)
    fsdp = fsdp_wrap_gpt(model, ShardingStrategy[strategy], force_collectives=True, reduce_fp32=reduce_fp32)
    losses = _train(fsdp, fsdp.parameters(), ids, no_sync=fsdp.no_sync)
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-2, (losses, ref_losses)
    sd, ref_sd = fsdp.state_dict(), base.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in ref_sd:
        assert (sd[k].float().cpu() - ref_sd[k].float().cpu()).abs().max().item() < 1e-2, k
    units = [m.unit for m in fsdp.modules() if type(m).__name__ == "FSDP"]
    if strategy != "NO_SHARD":
        assert sum(u.num_allgathers for u in units) > 0
    assert sum(u.num_reductions for u in units) == 4 * len(units)


def test_fabric_measurement_and_self_test_run_on_the_rccl_communicator(pg):
    """The start-up pieces a multi-GPU job runs before training — the communicator self-test and the
    message-size ladder of the autotuner — on the real RCCL communicator (one rank: the collectives
    are copies, but the streams, fences, in-place reduce-scatter / all-gather views and the float64
    MAX are the ones used at any world size)."""
    from pytorch_distributed_amd.parallel import autotune, comm

    c = comm.get_communicator()
    comm._self_test(c, torch.device("cuda", 0))
    m = autotune.measure(c, torch.device("cuda", 0), ladder_mb=(1, 16), iters=3)
    assert set(m) == {"all_reduce", "rs_ag", "all_gather"}
    for rows in m.values():
        assert len(rows) == 2 and all(t > 0 for _, t in rows)
    policy = autotune.choose_policy(m, total_grad_bytes=1 << 30)
    assert policy.ddp_bucket_mb > 0 and policy.fits
