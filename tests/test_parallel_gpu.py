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


def _model(seed=0, n_layer=3, activation_checkpoint=None):
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

    torch.manual_seed(seed)
    cfg = get_config("gpt2", n_layer=n_layer).without_dropout()
    kw = {} if activation_checkpoint is None else {"activation_checkpoint": activation_checkpoint}
    return MyGPT2LMHeadModel(cfg, **kw).to(device=DEV, dtype=torch.bfloat16), cfg


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


# ------------------------------------------------------------------ exact collective counts on the real streams
def _comm_counts():
    from pytorch_distributed_amd.parallel import comm

    k = comm.get_communicator().counters()
    return {n: int(k[n]) for n in ("all_gather", "reduce_scatter", "all_reduce", "broadcast")}


def _delta(before):
    after = _comm_counts()
    return {n: after[n] - before[n] for n in after}


# 4 optimizer steps of 2 micro-batches (the first under no_sync); U block units + the root unit.
# FULL_SHARD, per step: the root is gathered once (it stays gathered until the synchronising backward
# ends), every block unit in each forward and again in each backward: 1 + 4 U all-gathers; one
# reduce-scatter per unit.
# With every block resident the backward all-gathers disappear and the units stay gathered through the
# no_sync micro-batch: one all-gather per unit and step, 1 + U.
@pytest.mark.parametrize("n_layer,unit_blocks,resident,all_gathers_per_step", [
    (3, 1, 0.0, 1 + 4 * 3), (6, 3, 0.0, 1 + 4 * 2), (3, 1, "auto", 1 + 3)])
def test_fsdp_full_shard_exact_collective_counts(pg, n_layer, unit_blocks, resident, all_gathers_per_step):
    from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt

    base, cfg = _model(2, n_layer=n_layer)
    ids = _ids(cfg, 8, seed=2)
    ref_losses = _train(base, base.parameters(), ids)
    model, _ = _model(2, n_layer=n_layer)
    fsdp = fsdp_wrap_gpt(model, ShardingStrategy.FULL_SHARD, unit_blocks=unit_blocks, force_collectives=True,
                         resident_gb=resident)
    before = _comm_counts()
    losses = _train(fsdp, fsdp.parameters(), ids, no_sync=fsdp.no_sync)
    got = _delta(before)
    n_units = 1 + n_layer // unit_blocks
    assert got["all_gather"] == 4 * all_gathers_per_step, got
    assert got["reduce_scatter"] == 4 * n_units and got["all_reduce"] == 0, got
    if resident == "auto":
        assert fsdp.resident_units == n_layer
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-2, (losses, ref_losses)


def test_ddp_decomposed_reduction_exact_counts(pg):
    from pytorch_distributed_amd.parallel import DDP

    base, cfg = _model(3)
    ids = _ids(cfg, 8, seed=3)
    ref_losses = _train(base, base.parameters(), ids)
    model, _ = _model(3)
    ddp = DDP(model, device_ids=[0], bucket_cap_mb=8, first_bucket_mb=1, force_collectives=True, decompose=True)
    before = _comm_counts()
    losses = _train(ddp, ddp.parameters(), ids, no_sync=ddp.no_sync)
    got = _delta(before)
    nb = ddp.reducer.num_buckets
    # world size 1: a bucket's "reduce-scatter + all-gather" degenerates to the all-reduce branch
    assert got["all_reduce"] + got["reduce_scatter"] == 4 * nb and got["all_gather"] in (0, 4 * nb), (got, nb)
    assert ddp.reducer.num_collectives == 4 * nb
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-2, (losses, ref_losses)


@pytest.mark.parametrize("wrapper", ["ddp", "fsdp"])
def test_fp32_gradient_accumulation_on_the_gpu_sinks(pg, wrapper):
    """grad_accum_dtype='fp32': the weight-gradient kernels write fp32 slices of the wrapper's buffer
    (bf16 parameters), other gradients are folded by the hooks; training matches the plain model."""
    from pytorch_distributed_amd.parallel import DDP, ShardingStrategy, fsdp_wrap_gpt

    base, cfg = _model(4)
    ids = _ids(cfg, 8, seed=4)
    ref_losses = _train(base, base.parameters(), ids)
    model, _ = _model(4)
    if wrapper == "ddp":
        wrapped = DDP(model, device_ids=[0], bucket_cap_mb=8, first_bucket_mb=1, force_collectives=True,
                      grad_accum_dtype="fp32")
        assert all(t is not None and t.dtype == torch.float32 for t in wrapped.reducer.accum_tensors())
    else:
        wrapped = fsdp_wrap_gpt(model, ShardingStrategy.FULL_SHARD, force_collectives=True, grad_accum_dtype="fp32")
        assert all(m.unit.accumulates_fp32() for m in wrapped.modules() if type(m).__name__ == "FSDP")
    losses = _train(wrapped, wrapped.parameters(), ids, no_sync=wrapped.no_sync)
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-2, (losses, ref_losses)
    sd, ref_sd = wrapped.state_dict(), base.state_dict()
    for k in ref_sd:
        assert (sd[k].float().cpu() - ref_sd[k].float().cpu()).abs().max().item() < 1e-2, k


def test_second_library_communicator_for_the_gather_queue():
    """An RCCL communicator with its own library communicator for all-gathers: both routes give the
    same results, and the start-up measurement that chooses between them runs."""
    from pytorch_distributed_amd import _ext
    from pytorch_distributed_amd.parallel import autotune

    ext = _ext.require()
    torch.cuda.set_device(0)
    c = ext.create_communicator("rccl", 0, 1, "127.0.0.1", 29733, 0, 60.0, "dual-test", True)
    try:
        assert c.has_dual() and c.dual_route()
        for route in (True, False):
            c.set_dual_route(route)
            src = torch.arange(4096, device=DEV, dtype=torch.bfloat16)
            out = torch.zeros(4096, device=DEV, dtype=torch.bfloat16)
            rs_in = torch.full((1 << 20,), 2.0, device=DEV)
            rs_out = torch.empty(1 << 20, device=DEV)
            w1 = c.reduce_scatter(rs_out, rs_in, "avg", 1)
            w2 = c.all_gather(out, src, 0)
            w2.wait()
            w1.wait()
            torch.cuda.synchronize()
            assert torch.equal(out, src) and bool((rs_out == 2.0).all())
        got = autotune.measure_gather_behind_reduce(c, torch.device("cuda", 0), reduce_mb=16, gather_mb=2, iters=2)
        assert set(got) == {"shared", "dual"} and all(g > 0 and p >= g * 0.5 for g, p in got.values())
        assert not c.dual_route()  # the measurement leaves the route as it found it
    finally:
        c.close()


def test_in_process_kernel_trace(tmp_path):
    """StepProfiler alone (no external profiler): the active window's kernel dispatches land in the
    Chrome trace next to the ranges, and the analyzer reads it."""
    from pytorch_distributed_amd import _ext
    from pytorch_distributed_amd.utils import StepProfiler
    from pytorch_distributed_amd.utils import trace_analysis as ta

    if not _ext.require().kernel_trace_available():
        pytest.fail("in-process kernel tracing unavailable: " + _ext.require().kernel_trace_unavailable_reason())
    model, cfg = _model(5, n_layer=2)
    ids = _ids(cfg, 6, seed=5)
    path = str(tmp_path / "rank0_trace.json")
    with StepProfiler(wait=1, warmup=1, active=3, repeat=1, trace_path=path) as prof:
        assert prof.collects_kernels
        for x in ids:
            with prof.range("forward"):
                loss = model(x[:, :-1], x[:, 1:])
            with prof.range("backward"):
                loss.backward()
            prof.step()
    assert prof.traces_written == 1 and prof.kernels_written > 50
    ev = ta.load_trace(path)
    steps = sorted(e["name"] for e in ev if e.get("name", "").startswith("ProfilerStep#") and e.get("tid") == 0)
    assert steps == ["ProfilerStep#2", "ProfilerStep#3", "ProfilerStep#4"]
    kernels = [e for e in ev if e.get("cat") == "kernel"]
    names = " ".join(e["name"] for e in kernels)
    assert "attn_fwd_kernel" in names and "wgrad" in names, names[:2000]
    win = ta.step_window(ev, 3)
    inside = [k for k in kernels if win[0] <= k["ts"] <= win[1]]
    assert len(inside) > 10                                  # kernel rows sit on the ranges' clock
    tb = ta.temporal_breakdown(ev, step=3)
    assert tb["compute_us"] > 0 and tb["compute_pct"] > 20, tb
    # nothing from the wait / warm-up steps was collected
    first = min(k["ts"] for k in kernels)
    assert first >= ta.step_window(ev, 2)[0] - 2000, (first, ta.step_window(ev, 2))


@pytest.mark.parametrize("wrapper,accum", [("plain", 4), ("plain", 3), ("plain", 6), ("plain_full", 4), ("plain_none", 2),
                                           ("ddp", 4), ("ddp_fp32", 4), ("fsdp", 4), ("fsdp_grad_op", 2), ("fsdp_full", 2)])
def test_weight_gradients_deferred_over_the_accumulation_window(pg, wrapper, accum, monkeypatch):
    """Inside ops.linear.accumulation_window the (dY, X) pairs of the earlier micro-batches are kept and
    the last backward forms each weight gradient as one product over the whole window: same training
    as per-micro-batch products (PDA_DEFER_WGRAD=0), nothing left pending, fewer kernel launches."""
    from pytorch_distributed_amd import ops as linear_ops
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import DDP, ShardingStrategy, fsdp_wrap_gpt

    def build():
        policy = "full" if wrapper.endswith("_full") else "none" if wrapper.endswith("_none") else None
        model, cfg = _model(7, activation_checkpoint=policy)  # full: the linear backward runs inside a block's recomputation
        if wrapper.startswith("ddp"):
            model = DDP(model, device_ids=[0], bucket_cap_mb=8, first_bucket_mb=1, force_collectives=True,
                        grad_accum_dtype="fp32" if wrapper == "ddp_fp32" else "param")
        elif wrapper.startswith("fsdp"):
            strategy = ShardingStrategy.SHARD_GRAD_OP if wrapper == "fsdp_grad_op" else ShardingStrategy.FULL_SHARD
            model = fsdp_wrap_gpt(model, strategy, force_collectives=True)
        return model, cfg

    def run(defer: bool):
        monkeypatch.setenv("PDA_DEFER_WGRAD", "1" if defer else "0")
        model, cfg = build()
        ids = _ids(cfg, 2 * accum, seed=7)
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
        no_sync = getattr(model, "no_sync", None)
        losses, pending = [], []
        for step in range(2):
            for a in range(accum):
                x = ids[step * accum + a]
                with linear_ops.accumulation_window(a, accum):
                    if no_sync is not None and a < accum - 1:
                        with no_sync():
                            loss = model(x[:, :-1], x[:, 1:]) / accum
                            loss.backward()
                    else:
                        loss = model(x[:, :-1], x[:, 1:]) / accum
                        loss.backward()
                pending.append(linear_ops.pending_weight_grads())
                losses.append(loss.item() * accum)
            opt.step(zero_grad=True)
        torch.cuda.synchronize()
        sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
        return losses, pending, sd

    losses_d, pending_d, sd_d = run(True)
    losses_p, pending_p, sd_p = run(False)
    n_linear = 3 * 4  # c_attn, attn c_proj, c_fc, mlp c_proj in each of 3 blocks
    assert pending_p == [0] * (2 * accum)
    assert pending_d == ([n_linear] * (accum - 1) + [0]) * 2, pending_d
    assert max(abs(a - b) for a, b in zip(losses_d, losses_p)) < 2e-2, (losses_d, losses_p)
    for k in sd_p:
        assert (sd_d[k] - sd_p[k]).abs().max().item() < 1e-2, k


def test_deferred_weight_gradients_are_closer_to_fp32_than_bf16_accumulation(monkeypatch):
    """The window's single product accumulates in fp32: its bf16 result is nearer to the exact sum than
    four bf16 read-modify-writes of .grad; a window abandoned half way leaves nothing behind."""
    from pytorch_distributed_amd import ops
    linear_ops = ops

    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 384, device=DEV, dtype=torch.bfloat16)
    xs = [torch.randn(4, 256, 512, device=DEV, dtype=torch.bfloat16) for _ in range(4)]
    gs = [torch.randn(4, 256, 384, device=DEV, dtype=torch.bfloat16) for _ in range(4)]
    exact_w = sum(g.flatten(0, 1).float().t() @ x.flatten(0, 1).float() for g, x in zip(gs, xs))
    exact_b = sum(g.flatten(0, 1).float().sum(0) for g in gs)

    def window(defer):
        monkeypatch.setenv("PDA_DEFER_WGRAD", "1" if defer else "0")
        lin.weight.grad = lin.bias.grad = None
        for i, (x, g) in enumerate(zip(xs, gs)):
            with linear_ops.accumulation_window(i, 4):
                ops.linear(x, lin.weight, lin.bias).backward(g)
            if defer and i < 3:
                assert lin.weight.grad is None and lin.bias.grad is None  # nothing formed yet
        ew = ((lin.weight.grad.float() - exact_w).norm() / exact_w.norm()).item()
        eb = ((lin.bias.grad.float() - exact_b).norm() / exact_b.norm()).item()
        return ew, eb

    ew_d, eb_d = window(True)
    ew_p, eb_p = window(False)
    assert ew_d < 3e-3 and eb_d < 3e-3, (ew_d, eb_d)
    assert ew_d < 0.8 * ew_p, (ew_d, ew_p)
    # abandoned window: the next window starting at 0 drops the kept operands
    monkeypatch.setenv("PDA_DEFER_WGRAD", "1")
    with linear_ops.accumulation_window(0, 4):
        ops.linear(xs[0], lin.weight, lin.bias).backward(gs[0])
    assert linear_ops.pending_weight_grads() == 1
    with linear_ops.accumulation_window(0, 2):
        pass
    assert linear_ops.pending_weight_grads() == 0


def test_fp32_storage_accumulates_through_a_window():
    """``--dtype fp32`` on the GPU: every product runs on a bf16 copy of the fp32 weight that is made anew in each
    forward. Such copies are never filed in the window (they could not be found again); the window closes
    without complaint and the fp32 gradients are the sum over the micro-batches."""
    from pytorch_distributed_amd import ops

    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 128, device=DEV, dtype=torch.float32)
    xs = [torch.randn(2, 64, 256, device=DEV) for _ in range(3)]
    gs = [torch.randn(2, 64, 128, device=DEV) for _ in range(3)]
    for i, (x, g) in enumerate(zip(xs, gs)):
        with ops.accumulation_window(i, 3):
            ops.linear(x, lin.weight, lin.bias).backward(g)
        assert ops.pending_weight_grads() == 0
    exact_w = sum(g.flatten(0, 1).t() @ x.flatten(0, 1) for g, x in zip(gs, xs))
    exact_b = sum(g.flatten(0, 1).sum(0) for g in gs)
    assert lin.weight.grad.dtype == torch.float32
    assert ((lin.weight.grad - exact_w).norm() / exact_w.norm()).item() < 1e-2   # bf16 operands, fp32 sums
    assert ((lin.bias.grad - exact_b).norm() / exact_b.norm()).item() < 1e-2
