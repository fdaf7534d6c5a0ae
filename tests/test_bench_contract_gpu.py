"""The benchmark entry point's contract, checked end to end on a GPU: `bench.py` prints exactly one
JSON line on stdout with the agreed fields, directly and under `torch.distributed.run` (the way a
multi-GPU run is launched), and `__graft_entry__.smoke()` runs."""
import json
import os
import subprocess
import sys

import pytest

from tests.dist_utils import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold the JSON line only, got {len(lines)} lines: {lines[:3]}"
    return json.loads(lines[0])


def _check(res, n_gpus, steps, warmup, parallelism_prefix):
    assert REQUIRED <= set(res), REQUIRED - set(res)
    assert res["n_gpus"] == n_gpus and res["steps"] == steps and res["warmup"] == warmup
    assert res["unit"] == "tokens/s" and res["higher_is_better"] is True and res["scaling"] == "weak"
    assert res["dtype"] == "bf16" and res["data"] == "synthetic" and res["vs_baseline"] is None
    cfg = res["config"]
    assert cfg["model"] == "gpt2" and cfg["seq_len"] == 1024 and cfg["global_batch"] == 8 * n_gpus
    assert cfg["parallelism"].startswith(parallelism_prefix)
    tokens_per_step = cfg["global_batch"] * cfg["seq_len"]
    assert abs(res["value"] - tokens_per_step / (res["ms_per_step"] / 1e3)) < 1e-6 * res["value"]


def test_bench_prints_one_json_line():
    res = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--model", "gpt2",
                "--batch-per-gpu", "8", "--micro-batch", "8"])
    _check(res, 1, 3, 1, "single")


def test_bench_under_torchrun_uses_the_wrapper():
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1",
                "--model", "gpt2", "--batch-per-gpu", "8", "--micro-batch", "8", "--wrap-single"])
    _check(res, 1, 3, 1, "fsdp_full_shard")


def test_smoke_entry():
    proc = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert proc.returncode == 0, proc.stderr[-2000:]


def test_bench_reports_the_reference_recipe_next_to_the_headline():
    """Without --no-reference-recipe the same JSON line carries a second, short measurement of the
    reference's recipe (32 sequences as 4 micro-batches of 8, selective recomputation)."""
    res = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--model", "gpt2",
                "--batch-per-gpu", "8", "--micro-batch", "8", "--reference-steps", "2"])
    _check(res, 1, 2, 1, "single")
    ref = res["reference_recipe"]
    assert "error" not in ref, ref
    assert ref["global_batch"] == 32 and ref["micro_batch"] == 8 and ref["grad_accum"] == 4
    assert ref["activation_checkpoint"] == "selective" and ref["steps"] == 2
    assert abs(ref["value"] - 32 * 1024 / (ref["ms_per_step"] / 1e3)) < 1e-6 * ref["value"]
    merged = ref["merged_window"]  # the same 32 sequences as one pass: what the trainers do when memory allows
    assert abs(merged["value"] - 32 * 1024 / (merged["ms_per_step"] / 1e3)) < 1e-6 * merged["value"]


def test_train_baseline_script_accumulates_in_fp32_buffers_and_defers_weight_gradients():
    """The single-GPU script at its own recipe shape (4 micro-batches per step) on a small model: with the
    fp32 accumulation buffers and one weight-gradient product per window it logs the same losses as with
    bf16 accumulation and a product per micro-batch (same seed, same synthetic tokens)."""
    def run(extra, env_extra, merge="off"):
        proc = subprocess.run([sys.executable, "train_baseline.py", "--synthetic-data", "--model", "gpt2", "--max-steps", "6",
                               "--log-every", "1", "--no-profile", "--global-batch-size", "16", "--micro-batch-size", "4",
                               "--sequence-length", "256", "--dropout", "0", "--merge-micro-batches", merge, *extra],
                              cwd=ROOT, env=dict(os.environ, **env_extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, timeout=170)
        assert proc.returncode == 0, proc.stderr[-2000:]
        losses = [float(ln.split("loss=")[1].split()[0]) for ln in proc.stdout.splitlines() if "loss=" in ln]
        assert len(losses) == 6, proc.stdout[-1500:]
        return losses, proc.stdout

    new, out = run([], {})
    assert "Gradient accumulation in fp32 buffers" in out
    old, out_old = run(["--grad-accum-dtype", "param"], {"PDA_DEFER_WGRAD": "0"})
    assert "Gradient accumulation in fp32 buffers" not in out_old
    assert max(abs(a - b) for a, b in zip(new, old)) < 3e-2, (new, old)
    assert all(10.0 < v < 11.5 for v in new), new  # random tokens: the loss stays near ln(vocabulary)
    # the scripts' default: from the second window on the 4 micro-batches run as one pass of 16 sequences
    auto, out_auto = run([], {}, merge="auto")
    assert "runs as 1 pass(es) of 16 sequences" in out_auto, out_auto[-1500:]
    assert max(abs(a - b) for a, b in zip(new, auto)) < 3e-2, (new, auto)
