"""The native communication layer on CPUs: rendezvous store, socket-mesh collectives at world sizes
2 / 4 / 8, the recording decorator (exact FSDP / DDP collective sequences), the torch.distributed
adapter, the start-up policy on a synthetic latency / bandwidth model, and the autograd contract the
gradient sinks rely on."""
import copy

import pytest
import torch

from tests import dist_utils
from tests.dist_utils import run_distributed
from tests.test_ddp_cpu import _batch, _tiny_model


# ------------------------------------------------------------------ collectives
def _collectives(rank, world):
    from pytorch_distributed_amd.parallel import get_communicator

    c = get_communicator()
    assert c.rank == rank and c.world_size == world and c.backend == "socket" and not c.on_gpu
    for dtype in (torch.float32, torch.bfloat16, torch.float64, torch.int64):
        n = 1000 + 37  # not a multiple of the world size: all_reduce pads internally
        mine = (torch.arange(n) % 13 + rank).to(dtype)
        total = sum((torch.arange(n) % 13 + r).to(torch.float64) for r in range(world))
        t = mine.clone()
        c.all_reduce(t, "sum").wait()
        torch.testing.assert_close(t.to(torch.float64), total.to(dtype).to(torch.float64), rtol=0, atol=0)
        if dtype.is_floating_point:
            t = mine.clone()
            c.all_reduce(t, "avg").wait()
            torch.testing.assert_close(t.to(torch.float64), (total / world).to(dtype).to(torch.float64), rtol=0, atol=0)
        t = mine.clone()
        c.all_reduce(t, "max").wait()
        assert torch.equal(t, (torch.arange(n) % 13 + world - 1).to(dtype))
    # all_gather: slot r holds rank r's tensor
    chunk = 257
    out = torch.zeros(world * chunk)
    c.all_gather(out, torch.full((chunk,), float(rank))).wait()
    assert torch.equal(out.view(world, chunk), torch.arange(world, dtype=torch.float32)[:, None].expand(world, chunk))
    # in-place form: the input is the rank's own slot of the output
    out = torch.zeros(world * chunk)
    out.view(world, chunk)[rank] = rank + 1.0
    c.all_gather(out, out.view(world, chunk)[rank].view(-1)).wait()
    assert torch.equal(out.view(world, chunk)[:, 0], torch.arange(1, world + 1, dtype=torch.float32))
    # reduce_scatter: rank r receives the sum of everyone's chunk r
    inp = (torch.arange(world * chunk, dtype=torch.float32) + 1000 * rank)
    out = torch.empty(chunk)
    c.reduce_scatter(out, inp, "sum").wait()
    expect = sum(torch.arange(world * chunk, dtype=torch.float32) + 1000 * r for r in range(world))
    assert torch.equal(out, expect.view(world, chunk)[rank])
    # broadcast from a non-zero root
    t = torch.full((33,), float(rank))
    c.broadcast(t, world - 1).wait()
    assert torch.equal(t, torch.full((33,), float(world - 1)))
    # host-side exchange + barrier
    got = c.exchange(f"rank{rank}".encode())
    assert got == [f"rank{r}".encode() for r in range(world)]
    c.barrier()
    k = c.counters()
    assert k["all_gather"] == 2 and k["reduce_scatter"] == 1 and k["broadcast"] == 1 and k["all_reduce"] >= 8


@pytest.mark.parametrize("world", [2, 4, 8])
def test_socket_collectives(world):
    run_distributed(_collectives, world=world)


def test_socket_mesh_turns_away_foreign_clients(monkeypatch):
    """A rendezvous-store client that scans a port range for its server can land on a rank's mesh listener (an
    ephemeral port). With the injection every higher rank first opens such a connection — the store's greeting,
    then close — before its real one: the accept loop must turn it away and still build the mesh."""
    monkeypatch.setenv("PDA_SOCKET_MESH_STRANGER", "1")
    run_distributed(_collectives, world=3)


def _c10d_adapter(rank, world):
    """The same wrappers over a torch.distributed gloo group behind the Communicator interface."""
    from pytorch_distributed_amd.parallel import DDP, get_communicator

    assert get_communicator().backend == "c10d:gloo"
    ref = _tiny_model(0)
    model = copy.deepcopy(ref)
    ddp = DDP(model, bucket_cap_mb=0.25, first_bucket_mb=0.05)
    x, y = _batch(5, 2 * world)
    ref(x, y).backward()
    ddp(x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]).backward()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=2e-4, atol=2e-6, msg=lambda m: f"{n}: {m}")


def test_c10d_adapter_runs_the_same_reducer():
    run_distributed(_c10d_adapter, world=2, backend="c10d")


# ------------------------------------------------------------------ exact collective sequences
def _three_blocks(seed=0):
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

    torch.manual_seed(seed)
    return MyGPT2LMHeadModel(get_config("gpt2-tiny", n_layer=3).without_dropout())


def _recording(world_comm=None):
    from pytorch_distributed_amd import _ext
    from pytorch_distributed_amd.parallel import comm

    ext = _ext.require()
    inner = world_comm if world_comm is not None else ext.create_communicator("socket", 0, 1, "127.0.0.1", 1)
    rec = ext.recording_communicator(inner)
    comm.set_communicator(rec, "cpu")
    return rec


def test_fsdp_full_shard_collective_sequence_single_process():
    """FULL_SHARD, 3 blocks + root, prefetch one unit ahead, world size 1 with forced collectives:
    forward gathers root, b1, b2, b3 (each one unit ahead of its use); backward re-gathers b3, b2,
    b1 in that order, each before the reduce-scatter of the unit in front of it; one
    reduce-scatter per unit; nothing but all-gathers under no_sync."""
    from pytorch_distributed_amd.parallel import ShardingStrategy, comm, fsdp_wrap_gpt

    rec = _recording()
    try:
        model = fsdp_wrap_gpt(_three_blocks(), ShardingStrategy.FULL_SHARD, force_collectives=True)
        units = [m.unit for m in model.modules() if type(m).__name__ == "FSDP"]
        assert len(units) == 4  # root + 3 blocks
        size = {u.padded_numel: i for i, u in enumerate(units)}
        assert len(size) >= 2  # the root differs from the blocks; blocks share one size
        root_n, block_n = units[0].padded_numel, units[1].padded_numel
        x, y = _batch(1, 2)
        rec.log(clear=True)
        model(x, y).backward()
        log = rec.log(clear=True)
        kinds = [k for k, _, _ in log]
        assert kinds == ["all_gather"] * 4 + ["all_gather", "all_gather", "reduce_scatter", "all_gather",
                                              "reduce_scatter", "reduce_scatter", "reduce_scatter"], kinds
        # forward: root first, then the blocks; all-gathers on queue 0, reductions on queue 1
        assert [n for _, n, _ in log[:4]] == [root_n, block_n, block_n, block_n]
        assert all(q == 0 for k, _, q in log if k == "all_gather")
        assert all(q == 1 for k, _, q in log if k == "reduce_scatter")
        assert [n for k, n, _ in log if k == "reduce_scatter"] == [block_n, block_n, block_n, root_n]
        assert sum(u.num_allgathers for u in units) == 7 and sum(u.num_reductions for u in units) == 4
        # no_sync: parameters are still gathered, gradients are not reduced
        with model.no_sync():
            model(x, y).backward()
        assert [k for k, _, _ in rec.log(clear=True)] == ["all_gather"] * 7
        # the next synchronised pass reduces the accumulated (unsharded) gradients once per unit
        model(x, y).backward()
        assert [k for k, _, _ in rec.log(clear=True)].count("reduce_scatter") == 4
    finally:
        comm.shutdown()


def test_fsdp_resident_budget_skips_the_backward_all_gathers_of_the_last_units():
    """A budget that fits two blocks: b3 and b2 keep their parameters from forward to backward and are
    not gathered again; b1 still is. Gradients equal the plain FULL_SHARD run's."""
    from pytorch_distributed_amd.parallel import ShardingStrategy, comm, fsdp_wrap_gpt

    rec = _recording()
    try:
        ref = fsdp_wrap_gpt(_three_blocks(), ShardingStrategy.FULL_SHARD, force_collectives=True)
        block_bytes = [m.unit for m in ref.modules() if type(m).__name__ == "FSDP"][1].padded_numel * 4
        model = fsdp_wrap_gpt(_three_blocks(), ShardingStrategy.FULL_SHARD, force_collectives=True,
                              resident_gb=2.5 * block_bytes / 2**30)
        x, y = _batch(1, 2)
        ref(x, y).backward()
        rec.log(clear=True)
        model(x, y).backward()
        assert model.resident_units == 2
        log = rec.log(clear=True)
        kinds = [k for k, _, _ in log]
        assert kinds.count("all_gather") == 5 and kinds.count("reduce_scatter") == 4, kinds
        for a, b in zip(model.parameters(), ref.parameters()):
            torch.testing.assert_close(a.grad, b.grad)
        units = [m.unit for m in model.modules() if type(m).__name__ == "FSDP"]
        assert not any(u.is_unsharded() for u in units)  # everything is released by the end of backward
        # the second pass behaves the same (the kept units were resharded after their backward)
        model(x, y).backward()
        assert [k for k, _, _ in rec.log(clear=True)].count("all_gather") == 5
    finally:
        comm.shutdown()


def test_shard_grad_op_and_no_shard_sequences_single_process():
    from pytorch_distributed_amd.parallel import ShardingStrategy, comm, fsdp_wrap_gpt

    rec = _recording()
    try:
        x, y = _batch(2, 2)
        model = fsdp_wrap_gpt(_three_blocks(), ShardingStrategy.SHARD_GRAD_OP, force_collectives=True)
        model(x, y).backward()
        kinds = [k for k, _, _ in rec.log(clear=True)]
        assert kinds.count("all_gather") == 4 and kinds.count("reduce_scatter") == 4  # no re-gather in backward
        model = fsdp_wrap_gpt(_three_blocks(), ShardingStrategy.NO_SHARD, force_collectives=True)
        model(x, y).backward()
        assert [k for k, _, _ in rec.log(clear=True)] == ["all_reduce"] * 4
    finally:
        comm.shutdown()


@pytest.mark.parametrize("decompose", [False, True])
def test_ddp_collective_sequence_single_process(decompose):
    from pytorch_distributed_amd.parallel import DDP, comm

    rec = _recording()
    try:
        ddp = DDP(_tiny_model(0), bucket_cap_mb=0.25, first_bucket_mb=0.05, force_collectives=True, decompose=decompose)
        x, y = _batch(3, 2)
        with ddp.no_sync():
            ddp(x, y).backward()
        assert rec.log(clear=True) == []
        ddp(x, y).backward()
        log = rec.log(clear=True)
        sizes = [b.numel() for b in ddp.reducer.bucket_tensors()]
        # world size 1: the two-phase form degenerates to one all-reduce per bucket as well
        assert [k for k, _, _ in log] == ["all_reduce"] * len(sizes)
        assert [n for _, n, _ in log] == sizes and all(q == 1 for _, _, q in log)
    finally:
        comm.shutdown()


def _ddp_decomposed(rank, world):
    """Reduce-scatter + all-gather per bucket gives the same gradients as one all-reduce."""
    from pytorch_distributed_amd import _ext
    from pytorch_distributed_amd.parallel import DDP, comm, get_communicator

    rec = _ext.require().recording_communicator(get_communicator())
    ref = _tiny_model(0)
    x, y = _batch(77, 2 * world)
    ref(x, y).backward()
    model = copy.deepcopy(ref)
    model.zero_grad()
    ddp = DDP(model, bucket_cap_mb=0.25, first_bucket_mb=0.05, decompose=True, communicator=rec)
    ddp(x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]).backward()
    kinds = [k for k, _, _ in rec.log()]
    assert kinds[0] == "broadcast"  # the constructor's coalesced parameter broadcast from rank 0
    kinds = [k for k in kinds if k != "broadcast"]
    assert kinds == ["reduce_scatter", "all_gather"] * ddp.reducer.num_buckets
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=2e-4, atol=2e-6, msg=lambda m: f"{n}: {m}")
    comm.barrier()


def test_ddp_decomposed_reduction_matches():
    run_distributed(_ddp_decomposed, world=3)


# ------------------------------------------------------------------ start-up policy
def test_policy_from_a_synthetic_fabric_model():
    from pytorch_distributed_amd.parallel import autotune

    MB = 2**20

    def curve(latency, bandwidth):
        return [(int(s * MB), latency + s * MB / bandwidth) for s in (1, 4, 16, 64, 256)]

    # a fabric where the fused all-reduce is ring-bound (one link) and the two-phase form uses all links
    m = {"all_reduce": curve(60e-6, 120e9), "rs_ag": curve(90e-6, 500e9), "all_gather": curve(40e-6, 600e9)}
    a, b = autotune.fit(m["all_reduce"])
    assert abs(a - 60e-6) < 1e-6 and abs(b - 120e9) / 120e9 < 1e-6
    pol = autotune.choose_policy(m, total_grad_bytes=1548 * MB)
    assert pol.ddp_decompose is True
    # 90 % efficiency needs S >= 9 * latency * bandwidth = 9 * 90us * 500 GB/s = 386 MiB; six buckets per
    # backward cap it at 1548 / 6 = 258 MiB
    assert abs(pol.ddp_bucket_mb - 258.0) < 1.0
    # all-gather: 9 * 40us * 600 GB/s = 206 MiB per unit -> gpt2-large blocks of 39.4 MB bf16: 6 per unit, 36 blocks
    assert abs(pol.fsdp_unit_mb - 206.0) < 1.0
    assert pol.fsdp_unit_blocks(block_bytes=19_677_440 * 2, n_blocks=36) == 6
    assert pol.fsdp_unit_blocks(block_bytes=19_677_440 * 2, n_blocks=12) == 2  # at least six units stay
    # a switch-like fabric (low latency, all-reduce as fast as the two-phase form): small buckets, no decomposition
    m2 = {"all_reduce": curve(10e-6, 300e9), "rs_ag": curve(20e-6, 300e9), "all_gather": curve(8e-6, 300e9)}
    pol2 = autotune.choose_policy(m2, total_grad_bytes=1548 * MB)
    assert pol2.ddp_decompose is False and 16.0 <= pol2.ddp_bucket_mb <= 32.0
    assert pol2.fsdp_unit_blocks(block_bytes=19_677_440 * 2, n_blocks=36) == 1
    assert "ddp_bucket_mb" in pol.describe()


def _tune_live(rank, world):
    from pytorch_distributed_amd.parallel import DDP, autotune

    pol = autotune.tune(ladder_mb=(0.25, 1.0), verbose=False)
    assert pol.source == "measured" and set(pol.fits) == {"all_reduce", "rs_ag", "all_gather"}
    assert autotune.current_policy() is pol
    ddp = DDP(_tiny_model(0))  # no explicit bucket size: the measured policy applies
    assert ddp.bucket_cap_mb == pol.ddp_bucket_mb and ddp.decompose == pol.ddp_decompose
    autotune.set_policy(None)


def test_autotune_measures_and_wrappers_follow_it():
    run_distributed(_tune_live, world=2)


# ------------------------------------------------------------------ autograd contract of the gradient sinks
def test_accumulator_post_hook_fires_when_an_op_returns_no_gradient():
    """ops.linear / ops.norm write a parameter's gradient into the owner's flat buffer and hand
    autograd ``None`` for it; the reducer / sharding unit count the parameter as ready from a
    post-hook on its AccumulateGrad node. That hook must still fire exactly once per backward."""
    from torch.autograd.graph import Node  # noqa: F401

    w = torch.nn.Parameter(torch.randn(4, 3))
    sink = torch.zeros(4, 3)
    fired = []

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, weight):
            ctx.save_for_backward(x, weight)
            return x @ weight.t()

        @staticmethod
        def backward(ctx, dy):
            x, weight = ctx.saved_tensors
            sink.add_(dy.t() @ x)  # "gradient sink"
            return dy @ weight, None

    x = torch.randn(5, 3, requires_grad=True)
    acc = w.view_as(w).grad_fn.next_functions[0][0]  # the AccumulateGrad node of w
    acc.register_hook(lambda *a: fired.append(1))
    for k in range(2):
        Fn.apply(x, w).sum().backward()
        assert len(fired) == k + 1, "AccumulateGrad post-hook did not fire for a None gradient"
    assert w.grad is None and sink.abs().sum() > 0


def _fallback_worker(rank, world, expect_prefix):
    from pytorch_distributed_amd import parallel

    comm = parallel.get_communicator()
    assert comm.backend.startswith(expect_prefix), comm.backend
    t = torch.full((1000,), float(rank + 1))
    parallel.all_reduce(t)
    assert bool((t == world * (world + 1) / 2).all())


def _entry_with_fallback(rank, world, port, fail_rank, expect_prefix, err_dir):
    import os
    import sys
    import traceback

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if fail_rank is not None:
        os.environ["PDA_COMM_SELFTEST_FAIL_RANK"] = str(fail_rank)
    sys.path.insert(0, dist_utils.ROOT)
    try:
        from pytorch_distributed_amd.parallel import init_distributed, shutdown

        init_distributed(backend="socket", timeout_s=60, fallback="c10d", probe_timeout_s=30)
        _fallback_worker(rank, world, expect_prefix)
        shutdown()
    except BaseException:  # noqa: BLE001
        with open(os.path.join(err_dir, f"rank{rank}.err"), "w") as f:
            f.write(traceback.format_exc()[-4000:])
        os._exit(1)


@pytest.mark.parametrize("fail_rank,expect", [(None, "socket"), (1, "c10d:gloo")])
def test_verified_init_keeps_native_or_falls_back_on_every_rank(fail_rank, expect, tmp_path):
    """init_distributed(fallback="c10d"): all ranks pass the self-test -> native communicator; one
    rank fails it -> every rank switches to the torch.distributed group behind the same interface."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    port = dist_utils.free_port()
    procs = [ctx.Process(target=_entry_with_fallback, args=(r, 3, port, fail_rank, expect, str(tmp_path)))
             for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    errs = [f.read_text() for f in tmp_path.glob("rank*.err")]
    assert not errs and all(p.exitcode == 0 for p in procs), "\n".join(errs)
