"""Full / sharded checkpoints of sharded models and the distributed trainer, CPU processes."""
import copy
import os
import subprocess
import sys

import pytest
import torch

from tests.dist_utils import ROOT, free_port, run_distributed
from tests.test_ddp_cpu import _batch, _tiny_model


def _train_some(model, opt, rank, world, steps=3, seed=700):
    for s in range(steps):
        x, y = _batch(seed + s, 2 * world)
        model(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()
        opt.step(zero_grad=True)


def _full_and_sharded(rank, world, tmp):
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import ShardingStrategy, fsdp_wrap_gpt
    from pytorch_distributed_amd.parallel import checkpoint as ck

    ref = _tiny_model(11)
    ref_opt = FusedAdamW(ref.parameters(), lr=1e-3, weight_decay=0.1)
    model = fsdp_wrap_gpt(copy.deepcopy(ref), ShardingStrategy.FULL_SHARD)
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
    for s in range(3):
        x, y = _batch(700 + s, 2 * world)
        ref(x, y).backward()
        ref_opt.step(zero_grad=True)
    _train_some(model, opt, rank, world)

    # full optimizer state == the single-process optimizer's, parameter by parameter
    full = model.full_optim_state_dict(opt)
    ref_state = ref_opt.state_dict()
    assert full["state"].keys() == ref_state["state"].keys()
    for i, st in ref_state["state"].items():
        for k in ("exp_avg", "exp_avg_sq"):
            torch.testing.assert_close(full["state"][i][k], st[k], rtol=1e-3, atol=1e-6)
        assert float(full["state"][i]["step"]) == float(st["step"])

    # sharded save -> load into a fresh model: identical continuation
    ck.save_sharded_checkpoint(tmp, model, opt, step=3)
    fresh = fsdp_wrap_gpt(_tiny_model(99), ShardingStrategy.FULL_SHARD)
    fresh_opt = FusedAdamW(fresh.parameters(), lr=1e-3, weight_decay=0.1)
    assert ck.load_sharded_checkpoint(tmp, fresh, fresh_opt) == 3
    _train_some(model, opt, rank, world, steps=2, seed=900)
    _train_some(fresh, fresh_opt, rank, world, steps=2, seed=900)
    a, b = model.state_dict(), fresh.state_dict()
    for k in a:
        torch.testing.assert_close(a[k], b[k], rtol=0, atol=0)

    # full checkpoint -> a sharded model: load_full_optim_state_dict + load_state_dict
    again = fsdp_wrap_gpt(_tiny_model(98), ShardingStrategy.SHARD_GRAD_OP)
    again_opt = FusedAdamW(again.parameters(), lr=1e-3, weight_decay=0.1)
    again.load_state_dict(ref.state_dict())
    again.load_full_optim_state_dict(again_opt, ref_opt.state_dict())
    x, y = _batch(1234, 2 * world)
    ref(x, y).backward()
    ref_opt.step(zero_grad=True)
    again(x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]).backward()
    again_opt.step(zero_grad=True)
    sd = again.state_dict()
    for k, v in ref.state_dict().items():
        torch.testing.assert_close(sd[k], v, rtol=1e-3, atol=1e-4)


def test_full_and_sharded_checkpoints(tmp_path):
    run_distributed(_full_and_sharded, world=2, args=(str(tmp_path / "sharded"),))
    # offline consolidation of the 2-rank sharded checkpoint, loaded by a bare model (world 1)
    from pytorch_distributed_amd.parallel import checkpoint as ck

    cons = ck.consolidate_sharded_checkpoint(str(tmp_path / "sharded"))
    bare = _tiny_model(5)
    cons = ck.order_like_model(cons, bare)
    bare.load_state_dict(cons["model_state_dict"])
    assert cons["step"] == 3
    n_params = len(list(bare.parameters()))
    assert sorted(cons["optimizer_state_dict"]["state"].keys()) == list(range(n_params))
    from pytorch_distributed_amd.optim import FusedAdamW

    opt = FusedAdamW(bare.parameters(), lr=1e-3, weight_decay=0.1)
    opt.load_state_dict(cons["optimizer_state_dict"])
    for i, p in enumerate(bare.parameters()):
        assert opt.state[p]["exp_avg"].shape == p.shape


def _torchrun(script, extra, nproc=2, timeout=300):
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, script), *extra]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


COMMON = ["--model", "gpt2-tiny", "--global-batch-size", "8", "--micro-batch-size", "2", "--sequence-length", "32",
          "--max-steps", "4", "--log-every", "2", "--synthetic-data", "--dropout", "0", "--backend", "gloo",
          "--profiler-backend", "range"]


def test_train_ddp_script_on_cpu(tmp_path):
    r = _torchrun("train_ddp.py", COMMON + ["--trace-dir", str(tmp_path / "tr"), "--save-every", "2",
                                            "--checkpoint-dir", str(tmp_path / "ck")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "step=0 | loss=" in r.stdout and "step=2 | loss=" in r.stdout
    assert "grad_acc_steps=2" in r.stdout
    assert (tmp_path / "tr" / "rank0_trace.json").exists() and (tmp_path / "tr" / "rank1_trace.json").exists()
    ck = torch.load(tmp_path / "ck" / "checkpoint_step_2.pt", weights_only=True)
    assert not any(k.startswith("module.") for k in ck["model_state_dict"])


@pytest.mark.parametrize("strategy", ["FULL_SHARD", "SHARD_GRAD_OP"])
def test_train_fsdp_script_on_cpu_with_resume(tmp_path, strategy):
    args = COMMON + ["--strategy", strategy, "--no-profile", "--checkpoint-dir", str(tmp_path / "ck")]
    r = _torchrun("train_fsdp.py", args + ["--save-every", "2", "--sharded-checkpoint", str(tmp_path / "sh")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / "sh" / "rank1_of_2.pt").exists() and (tmp_path / "sh" / "meta.json").exists()
    ck = torch.load(tmp_path / "ck" / "checkpoint_step_2.pt", weights_only=True)
    assert "transformer.h.1.mlp.c_fc.weight" in ck["model_state_dict"]
    # one optimizer entry per distinct parameter (the tied lm_head / wte pair counts once)
    assert len(ck["optimizer_state_dict"]["state"]) == len({v.data_ptr() for v in ck["model_state_dict"].values()})
    assert ck["step"] == 3 and ck["data_state"]["batch_count"] == 3 * 2  # 3 optimizer steps of 2 micro-batches done
    r2 = _torchrun("train_fsdp.py", args + ["--resume", "latest", "--max-steps", "6"])
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    assert "Loaded checkpoint from step 3" in r2.stdout and "step=4 | loss=" in r2.stdout


def test_train_baseline_script_on_cpu(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    args = [a for a in COMMON if a not in ("--backend", "gloo")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_baseline.py"), *args,
                        "--trace-dir", str(tmp_path / "tr")], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Gradient accumulation steps: 4" in r.stdout and (tmp_path / "tr" / "trace.json").exists()
