"""Failure paths and hygiene of the distributed layer, on CPUs: the collective watchdog (poll ->
time-out -> abort -> every later collective raises -> the benchmark records the error and stops
without hanging), the rendezvous store's key collection, the environment a multi-rank GPU job needs,
stale unsharded parameters after an evaluation forward, the deferred logging of the trainers, and
fp32 gradient accumulation against bf16 accumulation at the reference's four micro-batches."""
import copy
import importlib.util
import os
import time

import pytest
import torch
from torch import nn

from pytorch_distributed_amd import _ext
from pytorch_distributed_amd.parallel import comm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single_socket():
    return _ext.require().create_communicator("socket", 0, 1, "127.0.0.1", 1)


# ------------------------------------------------------------------ watchdog -> abort
def test_watchdog_aborts_a_collective_that_never_completes():
    ext = _ext.require()
    c = ext.hanging_communicator(_single_socket(), healthy_calls=2, timeout_s=0.4)
    t = torch.ones(8)
    c.all_reduce(t, "sum").wait()                       # healthy
    c.all_gather(torch.zeros(8), torch.ones(8)).wait()  # healthy
    assert not c.aborted
    t0 = time.time()
    work = c.all_reduce(t, "sum")                       # never completes
    assert not work.completed()
    with pytest.raises(RuntimeError, match="aborted"):
        work.synchronize()                              # released by the watchdog's abort, not by completion
    waited = time.time() - t0
    assert 0.3 < waited < 5.0, waited
    assert c.aborted
    for call in (lambda: c.all_reduce(t, "sum"), lambda: c.reduce_scatter(torch.zeros(8), torch.ones(8), "sum"),
                 lambda: c.broadcast(t, 0), lambda: c.barrier()):
        with pytest.raises(RuntimeError, match="aborted"):
            call()


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_records_an_aborted_mode_and_stops():
    """bench.py's per-mode loop: a mode that dies in a timed-out collective is recorded as an error,
    nothing further is attempted on the aborted communicator, and with no good mode left the caller
    exits non-zero."""
    bench = _load_bench()
    ext = _ext.require()
    hanging = ext.hanging_communicator(_single_socket(), healthy_calls=0, timeout_s=0.2)
    calls = []

    def measure(kind):
        calls.append(kind)
        hanging.all_reduce(torch.ones(4), "sum").wait()  # blocks until the watchdog aborts, then raises
        return {"parallelism": kind, "value": 1.0}

    def state():
        return "aborted" if hanging.aborted else "ok"

    t0 = time.time()
    results, good = bench.run_modes(["ddp", "fsdp"], measure, state)
    assert time.time() - t0 < 5.0
    assert calls == ["ddp"] and good == []              # fsdp never started on the dead communicator
    assert "aborted" in results["ddp"]["error"]
    # one healthy mode survives another's ordinary failure
    def flaky(kind):
        if kind == "ddp":
            raise ValueError("out of memory")
        return {"parallelism": kind, "value": 2.0}
    results, good = bench.run_modes(["ddp", "fsdp"], flaky, lambda: "ok")
    assert good == ["fsdp"] and "error" in results["ddp"] and results["fsdp"]["value"] == 2.0
    # a single mode propagates its exception (nothing to protect)
    with pytest.raises(ValueError):
        bench.run_modes(["ddp"], flaky, lambda: "ok")


def test_bench_second_try_on_the_fallback_backend_only_when_it_can_help():
    """No mode measured on the native communicator of a multi-rank `--comm auto` job: each rank runs the
    benchmark again as a child on torch.distributed — once, never from such a child, never for one rank."""
    bench = _load_bench()

    class C:
        def __init__(self, backend):
            self.backend = backend

    assert bench.should_relaunch_on_fallback("auto", 8, C("rccl"), {})
    assert bench.should_relaunch_on_fallback("auto", 2, None, {})                 # communicator already shut down
    assert not bench.should_relaunch_on_fallback("auto", 8, C("c10d"), {})        # already on the fallback
    assert not bench.should_relaunch_on_fallback("native", 8, C("rccl"), {})      # the caller pinned the backend
    assert not bench.should_relaunch_on_fallback("auto", 1, C("rccl"), {})
    assert not bench.should_relaunch_on_fallback("auto", 8, C("rccl"), {"PDA_COMM_RELAUNCHED": "1"})


def test_bench_launch_plan_flips_backend_then_ipc_mode(monkeypatch):
    """The parent's retries: native, then torch.distributed, then torch.distributed with the IPC
    variable as the caller had it; every attempt is recorded in the result line."""
    bench = _load_bench()
    seen = []

    class Proc:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen.append((("--comm" in cmd and cmd[cmd.index("--comm") + 1]) or "auto", env.get("HSA_ENABLE_IPC_MODE_LEGACY")))
        if len(seen) < 3:
            return Proc(1, "RCCL banner\n")
        return Proc(0, 'noise\n{"metric": "m", "value": 1.0}\n')

    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "2"])
    printed = []
    monkeypatch.setattr("builtins.print", lambda *a, **k: printed.append(a[0]) if k.get("file") is None else None)
    code = bench.launch_ranks(bench.parse_args(["--gpus", "2"]))
    assert code == 0
    assert seen == [("auto", "0"), ("c10d", "0"), ("c10d", None)]
    import json
    rec = json.loads(printed[-1])
    assert [a["result"] for a in rec["launch_attempts"]] == [False, False, True]


# ------------------------------------------------------------------ store hygiene
def test_store_keys_do_not_grow_with_rounds():
    ext = _ext.require()
    port = 23000 + os.getpid() % 2000
    few = ext.debug_store_key_growth(3, 5, port)
    many = ext.debug_store_key_growth(3, 200, port + 40)
    assert many <= few + 2 and many <= 16, (few, many)


def test_multiprocess_env_only_for_multi_rank_jobs():
    env = {"WORLD_SIZE": "8"}
    comm.prepare_multiprocess_env(env)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    env = {"WORLD_SIZE": "8", "HSA_ENABLE_IPC_MODE_LEGACY": "1"}
    comm.prepare_multiprocess_env(env)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"  # an explicit setting wins
    env = {"WORLD_SIZE": "1"}
    comm.prepare_multiprocess_env(env)
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in env
    import pytorch_distributed_amd  # noqa: F401 - importing the package sets nothing
    assert "pytorch_distributed_amd" in __import__("sys").modules


# ------------------------------------------------------------------ stale unsharded parameters
def _tiny_gpt(seed):
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

    torch.manual_seed(seed)
    return MyGPT2LMHeadModel(get_config("gpt2-tiny", n_layer=2).without_dropout())


@pytest.mark.parametrize("strategy", ["FULL_SHARD", "SHARD_GRAD_OP"])
def test_checkpoint_loaded_after_an_eval_forward_is_seen(strategy):
    from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt
    from pytorch_distributed_amd.parallel.checkpoint import load_full_model_state_dict

    comm.set_communicator(_single_socket(), "cpu")
    try:
        model = fsdp_wrap_gpt(_tiny_gpt(0), ShardingStrategy[strategy], force_collectives=True)
        other = _tiny_gpt(1)
        ids = torch.randint(0, 100, (2, 16))
        model.eval()
        with torch.no_grad():
            before = model(ids)
        # every unit went back to its shard at the end of the grad-less forward
        assert not any(u.unit.is_unsharded() for u in model._order), "a unit stayed gathered after evaluation"
        load_full_model_state_dict(model, other.state_dict())
        with torch.no_grad():
            after = model(ids)
            expect = other.eval()(ids)
        assert not torch.allclose(before, after)
        torch.testing.assert_close(after, expect, rtol=1e-5, atol=1e-6)
    finally:
        comm.set_communicator(None)


# ------------------------------------------------------------------ deferred logging
def test_log_lines_are_deferred_by_one_step_and_all_printed(capsys):
    from pytorch_distributed_amd.data import SyntheticTokenLoader
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.train import Trainer

    model = _tiny_gpt(0)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    trainer = Trainer(model, opt, None, max_steps=5, global_batch_size=4, micro_batch_size=2, log_every_n_steps=2)
    loader = SyntheticTokenLoader(2, 16, 100, torch.device("cpu"), seed=3)
    seen_pending = []
    original = trainer._optimizer_step

    def spy():
        seen_pending.append(None if trainer._pending_log is None else trainer._pending_log["step"])
        original()

    trainer._optimizer_step = spy
    trainer.train(loader)
    out = capsys.readouterr().out
    steps = [int(line.split("|")[0].split("=")[1]) for line in out.splitlines() if line.startswith("step=")]
    assert steps == [0, 2, 4]
    # when step k + 1's optimizer step is enqueued, step k's line is still pending (not yet read back)
    assert seen_pending == [None, 0, None, 2, None]
    losses = [float(line.split("loss=")[1].split(" ")[0]) for line in out.splitlines() if line.startswith("step=")]
    assert all(0 < v < 20 for v in losses)


def test_lazy_scalar_mean_without_communicator():
    v = comm.LazyScalarMean(torch.tensor(3.5), reduce=False)
    assert v.ready() and v.value() == 3.5


# ------------------------------------------------------------------ fp32 gradient accumulation
class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(64, 96)
        self.b = nn.Linear(96, 32)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _micro_batches(n=4):
    g = torch.Generator().manual_seed(11)
    return [(torch.randn(16, 64, generator=g).bfloat16(), torch.randn(16, 32, generator=g).bfloat16()) for _ in range(n)]


def _exact_window_grads(net, batches):
    """fp64 sum of the bf16 micro-batch gradients: what an exact accumulator would hold."""
    total = [torch.zeros_like(p, dtype=torch.float64) for p in net.parameters()]
    for x, y in batches:
        net.zero_grad(set_to_none=True)
        ((net(x) - y).float().pow(2).mean() / len(batches)).backward()
        for t, p in zip(total, net.parameters()):
            t += p.grad.double()
    net.zero_grad(set_to_none=True)
    return total


def _window(wrapper, batches):
    for i, (x, y) in enumerate(batches):
        loss = (wrapper(x) - y).float().pow(2).mean() / len(batches)
        if i < len(batches) - 1:
            with wrapper.no_sync():
                loss.backward()
        else:
            loss.backward()


def _rel_err(got, exact):
    num = sum(float((g.double() - e).abs().sum()) for g, e in zip(got, exact))
    den = sum(float(e.abs().sum()) for e in exact)
    return num / den


@pytest.mark.parametrize("reduce_fp32", [False, True])
def test_ddp_fp32_accumulation_beats_bf16_accumulation(reduce_fp32):
    from pytorch_distributed_amd.parallel import DDP

    comm.set_communicator(_single_socket(), "cpu")
    try:
        torch.manual_seed(0)
        base = _Net().bfloat16()
        batches = _micro_batches(4)  # the reference's accumulation: global 32 = 4 x micro 8
        exact = _exact_window_grads(copy.deepcopy(base), batches)
        errs = {}
        for mode in ("param", "fp32"):
            net = copy.deepcopy(base)
            ddp = DDP(net, bucket_cap_mb=0.01, first_bucket_mb=0.004, force_collectives=True,
                      grad_accum_dtype=mode, reduce_fp32=reduce_fp32)
            _window(ddp, batches)
            grads = [p.grad for p in net.parameters()]
            assert all(g is not None and g.dtype == torch.bfloat16 for g in grads)
            errs[mode] = _rel_err(grads, exact)
            if mode == "fp32":
                # a second window starts from zero, not from the first window's sum
                _window(ddp, batches)
                assert abs(_rel_err([p.grad for p in net.parameters()], exact) - errs[mode]) < 1e-6
        # one rounding of the fp32 sum to bf16: relative error <= 2^-9 per element; the bf16 accumulator
        # rounds after every one of the four additions
        assert errs["fp32"] < 2.0 ** -9, errs
        assert errs["fp32"] < 0.75 * errs["param"], errs
    finally:
        comm.set_communicator(None)


@pytest.mark.parametrize("strategy", ["FULL_SHARD", "NO_SHARD"])
def test_fsdp_fp32_accumulation_beats_bf16_accumulation(strategy):
    from pytorch_distributed_amd.parallel import FSDP, ShardingStrategy

    comm.set_communicator(_single_socket(), "cpu")
    try:
        torch.manual_seed(0)
        base = _Net().bfloat16()
        batches = _micro_batches(4)
        exact = _exact_window_grads(copy.deepcopy(base), batches)
        errs = {}
        for mode in ("param", "fp32"):
            fsdp = FSDP(copy.deepcopy(base), ShardingStrategy[strategy], force_collectives=True, grad_accum_dtype=mode)
            _window(fsdp, batches)
            flat = fsdp.flat_param.grad
            assert flat is not None and flat.dtype == torch.float32  # the master shard's gradient
            # unit layout: parameters in registration order at the unit's offsets
            got = []
            for p, off in zip(fsdp._unit_params, fsdp.unit.offsets()):
                got.append(flat[off:off + p.numel()].view(p.shape))
            errs[mode] = _rel_err(got, exact)
        assert errs["fp32"] < 2.0 ** -9, errs
        assert errs["fp32"] < 0.75 * errs["param"], errs
    finally:
        comm.set_communicator(None)
