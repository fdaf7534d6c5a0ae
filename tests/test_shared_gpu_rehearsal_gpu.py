"""Multi-rank DDP / FSDP WITH the GPU kernels on a one-GPU box: 2-4 ranks share the card
(PDA_SHARED_GPU=1) and talk over torch.distributed's gloo backend through the adapter, because the
GPU collective library refuses two ranks on one device. What this covers that nothing else does:
the C++ reducer / sharding runtime at world size > 1 together with the HIP kernels — weight-gradient
kernels writing into bucket views and flat-gradient slices, AdamW on shard slices that do not start
at offset 0, padding for a world size that does not divide the unit, gradient accumulation with the
weight gradients formed once per window — against a single-process run on the whole batch."""
import copy
import os

import pytest
import torch

from tests.dist_utils import run_distributed

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(seed, n_layer=2):
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config

    torch.manual_seed(seed)
    cfg = get_config("gpt2", n_layer=n_layer).without_dropout()
    return MyGPT2LMHeadModel(cfg).to(device=DEV, dtype=torch.bfloat16), cfg


def _tokens(cfg, seed, n, T=256):
    g = torch.Generator(device="cpu").manual_seed(seed)
    buf = torch.randint(0, cfg.vocab_size, (n, T + 1), generator=g).to(DEV)
    return buf[:, :-1].contiguous(), buf[:, 1:].contiguous()


def _train(rank, world, kind, accum):
    from pytorch_distributed_amd import ops
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import DDP, ShardingStrategy, fsdp_wrap_gpt, get_communicator

    assert get_communicator().backend == "c10d:gloo" and get_communicator().world_size == world
    ref, cfg = _model(11)
    model = copy.deepcopy(ref)
    if kind.startswith("ddp"):
        wrapped = DDP(model, bucket_cap_mb=8, first_bucket_mb=1, decompose=kind == "ddp_decompose",
                      reduce_fp32=kind == "ddp_reduce_fp32", grad_accum_dtype="fp32" if kind == "ddp_fp32" else None)
    elif kind == "FULL_SHARD_resident":  # the last units keep their gathered parameters from forward to backward
        wrapped = fsdp_wrap_gpt(model, ShardingStrategy.FULL_SHARD, resident_gb="auto")
    elif kind == "FULL_SHARD_units2":    # two blocks per flat-parameter unit, fp32 reduce-scatter and accumulation
        wrapped = fsdp_wrap_gpt(model, ShardingStrategy.FULL_SHARD, unit_blocks=2, reduce_fp32=True,
                                grad_accum_dtype="fp32")
    else:
        wrapped = fsdp_wrap_gpt(model, ShardingStrategy[kind])
    opt_ref = FusedAdamW(ref.parameters(), lr=1e-3, weight_decay=0.1)
    opt = FusedAdamW(wrapped.parameters(), lr=1e-3, weight_decay=0.1)
    per_rank = 2
    losses, ref_losses = [], []
    for step in range(3):
        x, y = _tokens(cfg, 100 + step, per_rank * world * accum)
        # one process, the whole global batch at once
        loss_ref = ref(x, y)
        loss_ref.backward()
        opt_ref.step(zero_grad=True)
        ref_losses.append(loss_ref.item())
        # this rank's slice of every micro-batch
        window = 0.0
        for a in range(accum):
            lo = (a * world + rank) * per_rank
            with ops.accumulation_window(a, accum):
                if a < accum - 1:
                    with wrapped.no_sync():
                        loss = wrapped(x[lo:lo + per_rank], y[lo:lo + per_rank]) / accum
                        loss.backward()
                else:
                    loss = wrapped(x[lo:lo + per_rank], y[lo:lo + per_rank]) / accum
                    loss.backward()
            window += loss.item()
        opt.step(zero_grad=True)
        t = torch.tensor([window], device=DEV)
        get_communicator().all_reduce(t, "avg").wait()
        losses.append(t.item())
    torch.cuda.synchronize()
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 3e-2, (losses, ref_losses)
    sd, ref_sd = wrapped.state_dict(), ref.state_dict()  # collective for sharded models
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in ref_sd:
        assert (sd[k].float().cpu() - ref_sd[k].float().cpu()).abs().max().item() < 1.5e-2, k


@pytest.mark.parametrize("kind,world,accum", [("ddp", 2, 1), ("ddp", 2, 2), ("ddp_fp32", 2, 2), ("ddp_decompose", 3, 2),
                                              ("ddp_reduce_fp32", 2, 1), ("FULL_SHARD", 2, 1), ("FULL_SHARD", 3, 2),
                                              ("SHARD_GRAD_OP", 2, 2), ("NO_SHARD", 2, 2), ("FULL_SHARD", 4, 1),
                                              ("FULL_SHARD_resident", 2, 2), ("FULL_SHARD_units2", 2, 2)])
def test_multi_rank_training_on_one_shared_gpu_matches_single_process(kind, world, accum, monkeypatch):
    monkeypatch.setenv("PDA_SHARED_GPU", "1")
    run_distributed(_train, world=world, args=(kind, accum), timeout=200.0, backend="rccl")


def _sharded_checkpoint(rank, world, tmp):
    """Sharded save at world 2 on the GPU, load into a fresh wrapped model: same parameters, same next step."""
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.parallel import ShardingStrategy, checkpoint, fsdp_wrap_gpt

    model, cfg = _model(5)
    wrapped = fsdp_wrap_gpt(model, ShardingStrategy.FULL_SHARD)
    opt = FusedAdamW(wrapped.parameters(), lr=1e-3, weight_decay=0.1)
    x, y = _tokens(cfg, 7, 2 * world)
    for _ in range(2):
        wrapped(x[rank * 2:rank * 2 + 2], y[rank * 2:rank * 2 + 2]).backward()
        opt.step(zero_grad=True)
    checkpoint.save_sharded_checkpoint(tmp, wrapped, opt, step=2)
    fresh, _ = _model(6)
    wrapped2 = fsdp_wrap_gpt(fresh, ShardingStrategy.FULL_SHARD)
    opt2 = FusedAdamW(wrapped2.parameters(), lr=1e-3, weight_decay=0.1)
    assert checkpoint.load_sharded_checkpoint(tmp, wrapped2, opt2) == 2
    a, b = wrapped.state_dict(), wrapped2.state_dict()
    for k in a:  # the restored model IS the saved one, bit for bit
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    for m, o in ((wrapped, opt), (wrapped2, opt2)):
        m(x[rank * 2:rank * 2 + 2], y[rank * 2:rank * 2 + 2]).backward()
        o.step(zero_grad=True)
    a, b = wrapped.state_dict(), wrapped2.state_dict()
    diffs = [(a[k].float() - b[k].float()).abs() for k in a]
    worst = max(d.max().item() for d in diffs)
    mean = sum(d.sum().item() for d in diffs) / sum(d.numel() for d in diffs)
    # The optimizer state came along: a step from the restored state lands where the original lands. Not
    # bit for bit on the GPU — the library's stream-K GEMMs do not sum in a fixed order, and AdamW turns a
    # gradient at rounding level into a step of either sign, so single elements may differ by up to 2 x lr.
    # A lost moment or step count moves EVERY element by a sizeable part of lr (1e-3): the mean tells.
    assert worst <= 2.0e-3 and mean <= 2.0e-5, (worst, mean)


def test_sharded_checkpoint_round_trip_at_two_ranks_on_one_gpu(tmp_path, monkeypatch):
    monkeypatch.setenv("PDA_SHARED_GPU", "1")
    run_distributed(_sharded_checkpoint, world=2, args=(str(tmp_path),), timeout=200.0, backend="rccl")


def test_bench_and_train_scripts_with_two_ranks_on_one_shared_gpu(tmp_path):
    """The benchmark exactly as a multi-GPU run starts it (`--gpus 2`: own child ranks, start-up collective
    measurement, agreed GEMM plans, DDP then FSDP, the reference recipe, one JSON line) and the two
    distributed training scripts under torchrun, with both ranks on the one GPU of this box."""
    import json
    import subprocess
    import sys

    from tests.dist_utils import ROOT, free_port

    # (a launcher's variables left in this process by another test module must not reach the children)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_COMM_BACKEND")}
    env.update(PDA_SHARED_GPU="1", OMP_NUM_THREADS="4")
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--model", "gpt2", "--batch-per-gpu", "4",
                           "--micro-batch", "4", "--steps", "2", "--warmup", "1", "--reference-steps", "1"],
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 8
    assert res["config"]["comm_backend"] == "c10d:gloo"
    assert set(res["modes"]) == {"ddp", "fsdp", "fsdp_resident"}, res["modes"]
    assert all("error" not in m for m in res["modes"].values()), res["modes"]
    assert res["config"]["parallelism"] == "fsdp_full_shard2"        # the headline is the strict form
    assert res["modes"]["fsdp"]["resident_units"] == 0 and res["modes"]["fsdp_resident"]["resident_units"] > 0
    assert res["config"]["comm_policy"]["ddp_bucket_mb"] > 0          # the start-up measurement ran at world size 2
    assert "error" not in res["reference_recipe"] and res["reference_recipe"]["global_batch"] == 64
    assert res["launch_attempts"][0]["result"] is True
    tokens = res["config"]["global_batch"] * res["config"]["seq_len"]
    assert abs(res["value"] - tokens / (res["ms_per_step"] / 1e3)) < 1e-6 * res["value"]

    common = ["--synthetic-data", "--model", "gpt2", "--max-steps", "3", "--log-every", "1", "--no-profile",
              "--global-batch-size", "16", "--micro-batch-size", "4", "--sequence-length", "256", "--dropout", "0"]
    for script, extra in (("train_ddp.py", []), ("train_fsdp.py", ["--strategy", "FULL_SHARD"])):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), script, *common, *extra,
               "--checkpoint-dir", str(tmp_path / script)]
        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
        assert p.returncode == 0, p.stdout[-3000:]
        losses = [float(ln.split("loss=")[1].split()[0]) for ln in p.stdout.splitlines() if "loss=" in ln]
        assert len(losses) == 3 and all(10.0 < v < 11.5 for v in losses), p.stdout[-1500:]
        assert "world_size=2" in p.stdout and "c10d:gloo" in p.stdout
        # the two ranks agreed to run the second window's 2 micro-batches of 4 as one pass
        assert "runs as 1 pass(es) of 8 sequences" in p.stdout, p.stdout[-1500:]
