"""CPU unit tests: shard format, loaders, configs, schedule, trainer, tools."""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

from pytorch_distributed_amd import _ext
from pytorch_distributed_amd.data import (DistributedKJJ0DataLoader, KJJ0DataLoader, SyntheticTokenLoader,
                                          make_synthetic_fineweb_files, open_shard, read_header, write_shard)
from pytorch_distributed_amd.models import (MyGPT2LMHeadModel, get_config, gpt2_param_count)
from pytorch_distributed_amd.optim import CosineAnnealingLR, FusedAdamW
from pytorch_distributed_amd.train import Trainer


# ------------------------------------------------------------------ shard format / loaders
def _shards(tmp_path, sizes, vocab=1000):
    paths = []
    rng = np.random.default_rng(0)
    for i, n in enumerate(sizes):
        p = tmp_path / f"fineweb_train_{i + 1:06d}.bin"
        write_shard(p, rng.integers(0, vocab, size=n))
        paths.append(p)
    return paths


def test_shard_roundtrip_and_header_validation(tmp_path):
    toks = np.arange(5000) % 65536
    p = tmp_path / "a.bin"
    write_shard(p, toks)
    assert read_header(p) == 5000
    assert np.array_equal(np.asarray(open_shard(p)), toks.astype(np.uint16))
    raw = bytearray(p.read_bytes())
    raw[0] ^= 0xFF
    bad = tmp_path / "bad.bin"
    bad.write_bytes(raw)
    with pytest.raises(ValueError, match="magic"):
        read_header(bad)
    trunc = tmp_path / "trunc.bin"
    trunc.write_bytes(p.read_bytes()[:-10])
    with pytest.raises(ValueError, match="mismatch"):
        open_shard(trunc)


def _reference_stream(paths, T):
    """The reference loader's sequence stream, written from its specification: per shard,
    windows [pos, pos+T+1) with pos += T while pos + T < len."""
    for p in sorted(map(str, paths)):
        toks = np.asarray(open_shard(p)).astype(np.int64)
        pos = 0
        while pos + T < len(toks):
            w = toks[pos: pos + T + 1]
            yield w[:-1], w[1:]
            pos += T


def test_single_loader_matches_window_arithmetic(tmp_path):
    paths = _shards(tmp_path, [1000, 777, 1531])
    B, T = 4, 32
    ref = list(_reference_stream(paths, T))
    got = list(KJJ0DataLoader(paths, B, T, "cpu", verbose=False))
    assert len(got) == len(ref) // B
    for i, (x, y) in enumerate(got):
        assert x.dtype == torch.int64 and x.shape == (B, T)
        for j in range(B):
            assert np.array_equal(x[j].numpy(), ref[i * B + j][0])
            assert np.array_equal(y[j].numpy(), ref[i * B + j][1])


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_distributed_loader_partitions_the_single_stream(tmp_path, world):
    """Concatenating the ranks' batches reproduces the single-process loader (within a shard)."""
    paths = _shards(tmp_path, [40000])
    b, T = 2, 16
    single = list(KJJ0DataLoader(paths, b * world, T, "cpu", verbose=False))
    ranks = [list(DistributedKJJ0DataLoader(paths, b, T, "cpu", rank=r, world_size=world, verbose=False))
             for r in range(world)]
    n = min(len(r) for r in ranks)
    assert n >= len(single) - 1 and n > 10
    for i in range(n):
        x = torch.cat([ranks[r][i][0] for r in range(world)])
        y = torch.cat([ranks[r][i][1] for r in range(world)])
        assert torch.equal(x, single[i][0]) and torch.equal(y, single[i][1])


def test_distributed_loader_env_defaults(tmp_path, monkeypatch):
    paths = _shards(tmp_path, [5000])
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    dl = DistributedKJJ0DataLoader(paths, 2, 8, "cpu", verbose=False)
    assert (dl.rank, dl.world_size) == (1, 2)


def test_native_prefetcher_matches_python_loader(tmp_path):
    ext = _ext.require()
    paths = _shards(tmp_path, [3000, 2500])
    b, T, world = 2, 16, 2
    for rank in range(world):
        want = list(DistributedKJJ0DataLoader(paths, b, T, "cpu", rank=rank, world_size=world, verbose=False))
        slots = [torch.empty(b * T + 1, dtype=torch.int64) for _ in range(3)]
        pf = ext.TokenPrefetcher(sorted(map(str, paths)), b, T, rank, world, slots)
        got = []
        while True:
            i = pf.next()
            if i < 0:
                break
            buf = slots[i].clone()
            pf.release(i)
            got.append((buf[:-1].view(b, T), buf[1:].view(b, T)))
        pf.stop()
        assert len(got) == len(want) and pf.shards_opened == 2
        for (x, y), (wx, wy) in zip(got, want):
            assert torch.equal(x, wx) and torch.equal(y, wy)


def test_loader_prefetch_option_matches_python_path(tmp_path):
    paths = _shards(tmp_path, [3000, 2500])
    for rank in range(2):
        want = list(DistributedKJJ0DataLoader(paths, 2, 16, "cpu", rank=rank, world_size=2, verbose=False))
        got = list(DistributedKJJ0DataLoader(paths, 2, 16, "cpu", rank=rank, world_size=2, verbose=False,
                                             prefetch=True))
        assert len(got) == len(want) > 0
        for (x, y), (wx, wy) in zip(got, want):
            assert torch.equal(x, wx) and torch.equal(y, wy)


def test_synthetic_files_and_loader(tmp_path):
    files = make_synthetic_fineweb_files(tmp_path, num_train_files=2, tokens_per_file=4096)
    assert [f.name for f in files] == ["fineweb_val_000000.bin", "fineweb_train_000001.bin", "fineweb_train_000002.bin"]
    x, y = next(iter(SyntheticTokenLoader(3, 8, 100)))
    assert x.shape == (3, 8) and torch.equal(x[:, 1:], y[:, :-1])


# ------------------------------------------------------------------ configs / model
@pytest.mark.parametrize("name,total", [("gpt2", 124_439_808), ("gpt2-medium", 354_823_168),
                                        ("gpt2-large", 774_030_080), ("gpt2-xl", 1_557_611_200)])
def test_param_count_formula(name, total):
    assert gpt2_param_count(get_config(name)) == total


def test_model_matches_formula_and_schema():
    cfg = get_config("gpt2-tiny")
    m = MyGPT2LMHeadModel(cfg)
    assert m.num_parameters() == gpt2_param_count(cfg)
    keys = list(m.state_dict().keys())
    assert keys[0] == "transformer.wte.weight" and keys[1] == "transformer.wpe.weight"
    assert "transformer.h.0.attn.c_attn.weight" in keys and "transformer.h.1.mlp.c_proj.bias" in keys
    assert keys[-1] == "lm_head.weight" and not any("mask" in k for k in keys)
    assert m.lm_head.weight is m.transformer.wte.weight
    assert m.state_dict()["transformer.h.0.attn.c_attn.weight"].shape == (3 * cfg.n_embd, cfg.n_embd)


def test_conv1d_conversion_transposes_only_conv1d_weights():
    sd = {"transformer.h.0.attn.c_attn.weight": torch.randn(8, 24), "transformer.h.0.attn.c_attn.bias": torch.randn(24),
          "transformer.h.0.mlp.c_fc.weight": torch.randn(8, 32), "transformer.wte.weight": torch.randn(10, 8),
          "transformer.h.0.ln_1.weight": torch.randn(8)}
    out = MyGPT2LMHeadModel._convert_conv1d_to_linear_state_dict(sd)
    assert out["transformer.h.0.attn.c_attn.weight"].shape == (24, 8)
    assert out["transformer.h.0.mlp.c_fc.weight"].shape == (32, 8)
    assert out["transformer.wte.weight"].shape == (10, 8)
    assert torch.equal(out["transformer.h.0.attn.c_attn.bias"], sd["transformer.h.0.attn.c_attn.bias"])


def test_block_single_tensor_api_equals_fused_pair():
    cfg = get_config("gpt2-tiny").without_dropout()
    m = MyGPT2LMHeadModel(cfg)
    blk = m.transformer.h[0]
    x = torch.randn(2, 16, cfg.n_embd)
    out, res = blk(x, None, fused=True)
    torch.testing.assert_close(blk(x), out + res)


def test_hf_style_config_object_is_accepted():
    class HF:  # what transformers.GPT2Config exposes (n_ctx absent / None on recent releases)
        n_layer, n_head, n_embd, vocab_size, n_positions, n_ctx = 2, 2, 128, 300, 64, None
        activation_function, attn_pdrop, resid_pdrop, embd_pdrop, layer_norm_epsilon = "gelu_new", 0.0, 0.0, 0.0, 1e-5

    m = MyGPT2LMHeadModel(HF())
    assert m.transformer.wpe.weight.shape == (64, 128)


# ------------------------------------------------------------------ schedule / optimizer / trainer
def test_cosine_schedule_matches_torch():
    p1, p2 = [torch.nn.Parameter(torch.zeros(1))], [torch.nn.Parameter(torch.zeros(1))]
    o1, o2 = torch.optim.SGD(p1, lr=3e-4), torch.optim.SGD(p2, lr=3e-4)
    ours = CosineAnnealingLR(o1, T_max=20, eta_min=3e-5)
    ref = torch.optim.lr_scheduler.CosineAnnealingLR(o2, T_max=20, eta_min=3e-5)
    for _ in range(20):
        assert math.isclose(ours.get_last_lr()[0], ref.get_last_lr()[0], rel_tol=1e-9)
        o1.step(); o2.step(); ours.step(); ref.step()
    state = ours.state_dict()
    again = CosineAnnealingLR(torch.optim.SGD(p1, lr=3e-4), T_max=20, eta_min=3e-5)
    again.load_state_dict(state)
    assert math.isclose(again.get_last_lr()[0], ours.get_last_lr()[0])


def test_fused_adamw_cpu_path_matches_torch_and_state_dict_layout():
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    o1, o2 = FusedAdamW(a, lr=1e-2, weight_decay=0.1), torch.optim.AdamW(b, lr=1e-2, weight_decay=0.1)
    for _ in range(5):
        for p, q in zip(a, b):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        o1.step(); o2.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-6)
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert s1["state"].keys() == s2["state"].keys()
    assert set(s1["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    o3 = FusedAdamW([torch.nn.Parameter(p.detach().clone()) for p in a], lr=1e-2, weight_decay=0.1)
    o3.load_state_dict(s2)  # a stock AdamW state loads into ours ...
    o2.load_state_dict(s1)  # ... and ours into stock AdamW


def _trainer(tmp_path, steps=4, save_every=None, seed=0):
    torch.manual_seed(seed)
    cfg = get_config("gpt2-tiny").without_dropout()
    model = MyGPT2LMHeadModel(cfg)
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
    sched = CosineAnnealingLR(opt, T_max=10, eta_min=1e-4)
    tr = Trainer(model, opt, sched, max_steps=steps, global_batch_size=4, micro_batch_size=2,
                 log_every_n_steps=2, save_every_n_steps=save_every, checkpoint_dir=str(tmp_path / "ckpt"))
    return tr, cfg


def test_trainer_log_format_accumulation_and_checkpoint(tmp_path, capsys):
    tr, cfg = _trainer(tmp_path, steps=4, save_every=2)
    assert tr.grad_accumulation_steps == 2
    tr.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=50))
    out = capsys.readouterr().out
    import re
    lines = [l for l in out.splitlines() if l.startswith("step=")]
    assert len(lines) == 2
    assert re.match(r"step=0 \| loss=\d+\.\d{4} \| lr=\d\.\d{2}e-\d{2} \| time=\d+\.\ds", lines[0])
    assert tr.current_step == 4 and tr.batch_count == 8
    path = tmp_path / "ckpt" / "checkpoint_step_2.pt"
    ck = torch.load(path, weights_only=True)
    # the reference's four keys plus the position of the token stream
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "step", "lr_scheduler_state_dict", "data_state"}
    assert ck["step"] == 3  # optimizer steps completed when step index 2 had finished
    assert ck["data_state"]["batch_count"] == 6 and ck["data_state"]["loader"] == {"batches": 6}
    tr2, _ = _trainer(tmp_path, seed=1)
    tr2.load_checkpoint(path)
    assert tr2.current_step == 3 and tr2.batch_count == 6
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v, ck["model_state_dict"][k])
    # checkpoints written by a DDP-wrapped torch model carry a "module." prefix: accepted
    ck["model_state_dict"] = {"module." + k: v for k, v in ck["model_state_dict"].items()}
    torch.save(ck, tmp_path / "prefixed.pt")
    tr2.load_checkpoint(tmp_path / "prefixed.pt")


def test_trainer_rejects_indivisible_batch():
    with pytest.raises(ValueError):
        Trainer(torch.nn.Linear(2, 2), None, global_batch_size=10, micro_batch_size=4)


@pytest.mark.skipif(not os.path.isdir("/root/reference/assignments/assignment1"), reason="reference not mounted")
def test_checkpoint_roundtrips_with_reference_model(tmp_path):
    """Our state_dict loads into the reference's model class key-for-key and vice versa."""
    sys.path.insert(0, "/root/reference/assignments/assignment1")
    try:
        from model.my_gpt2 import MyGPT2LMHeadModel as RefModel
    except Exception as exc:  # transformers / hub issues
        pytest.skip(f"reference model not importable here: {exc}")
    finally:
        sys.path.pop(0)
    cfg = get_config("gpt2-tiny").without_dropout()
    ours = MyGPT2LMHeadModel(cfg).eval()
    ref = RefModel(cfg, enable_activation_checkpoint=False).eval()
    missing, unexpected = ref.load_state_dict(ours.state_dict(), strict=True), None
    x = torch.randint(0, cfg.vocab_size, (2, 24))
    with torch.no_grad():
        torch.testing.assert_close(ours(x), ref(x), rtol=1e-4, atol=1e-4)
    ours2 = MyGPT2LMHeadModel(cfg)
    ours2.load_state_dict(ref.state_dict())
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())


# ------------------------------------------------------------------ profiler / trace analysis
def test_step_profiler_range_backend_schedule(tmp_path):
    from pytorch_distributed_amd.utils import StepProfiler

    path = tmp_path / "traces" / "rank0_trace.json"
    with StepProfiler(wait=1, warmup=1, active=3, repeat=1, trace_path=str(path), backend="range") as prof:
        for _ in range(8):
            with prof.range("forward"):
                torch.randn(8, 8) @ torch.randn(8, 8)
            prof.step()
    data = json.loads(path.read_text())
    steps = [e["name"] for e in data["traceEvents"] if e.get("name", "").startswith("ProfilerStep#")]
    assert steps == ["ProfilerStep#2", "ProfilerStep#3", "ProfilerStep#4"]
    assert sum(e.get("name") == "forward" for e in data["traceEvents"]) == 3


def test_trace_analysis_breakdown_overlap_and_diff():
    from pytorch_distributed_amd.utils import trace_analysis as ta

    def k(name, ts, dur, cat="kernel"):
        return {"ph": "X", "cat": cat, "name": name, "ts": ts, "dur": dur, "pid": 0, "tid": 7}

    base = [{"ph": "X", "cat": "user_annotation", "name": "ProfilerStep#4", "ts": 0, "dur": 100},
            k("gemm", 0, 40), k("attn", 50, 20)]
    ddp = base + [k("ncclDevKernel_AllReduce_Sum_bf16", 30, 40)]
    b = ta.temporal_breakdown(ddp)
    assert b["compute_us"] == 60 and b["non_compute_us"] == 10 and b["idle_us"] == 0 and b["span_us"] == 70
    o = ta.comm_comp_overlap(ddp)
    assert o["comm_us"] == 40 and o["overlapped_us"] == 30 and math.isclose(o["overlap_pct"], 75.0)
    d = ta.ops_diff(base, ddp, step=4)
    assert [x[0] for x in ta.comm_ops(d)] == ["ncclDevKernel_AllReduce_Sum_bf16"] and not d["deleted"]


def test_memory_estimate_matches_reference_model():
    from pytorch_distributed_amd.utils import estimate_training_memory

    e = estimate_training_memory(774_030_080, "fp32")
    assert math.isclose(e["params_mb"], 774_030_080 * 4 / 2**20) and math.isclose(e["optimizer_mb"], 2 * e["params_mb"])
    s = estimate_training_memory(774_030_080, "bf16", world_size=8, strategy="FULL_SHARD")
    assert math.isclose(s["total_mb"], 774_030_080 * 16 / 8 / 2**20)


def test_planned_mm_cpu_fallback_matches_torch():
    from pytorch_distributed_amd.ops.linear import planned_mm

    torch.manual_seed(0)
    a, w, b = torch.randn(6, 8), torch.randn(5, 8), torch.randn(5)
    torch.testing.assert_close(planned_mm(a, w, trans_b=True, bias=b), torch.nn.functional.linear(a, w, b))
    torch.testing.assert_close(planned_mm(a, w.t().contiguous()), a @ w.t())
    out = torch.empty(6, 5)
    planned_mm(a, w, trans_b=True, out=out)
    torch.testing.assert_close(out, a @ w.t())


def test_attention_dropout_quantisation_and_cpu_path():
    from pytorch_distributed_amd import ops
    from pytorch_distributed_amd.ops.attention import dropout_quantised

    assert dropout_quantised(0.0) == (0.0, 1.0)
    p, scale = dropout_quantised(0.1)
    assert p == 26 / 256 and abs(scale - 256 / 230) < 1e-12
    qkv = torch.randn(1, 16, 3 * 2 * 8)
    torch.manual_seed(5)
    a = ops.causal_attention(qkv, 2, dropout_p=0.5, training=True)
    b = ops.causal_attention(qkv, 2, dropout_p=0.5, training=False)
    assert not torch.equal(a, b) and torch.equal(b, ops.causal_attention(qkv, 2))
    keep = torch.ones(1, 2, 16, 16, dtype=torch.bool)
    torch.testing.assert_close(ops.attention_reference(qkv, 2, keep_mask=keep, keep_scale=1.0),
                               ops.attention_reference(qkv, 2))


def test_capturable_adamw_on_cpu_is_plain_adamw_and_graph_step_needs_it():
    from pytorch_distributed_amd.optim import FusedAdamW
    from pytorch_distributed_amd.utils.graphs import GraphedTrainStep

    torch.manual_seed(0)
    w0 = torch.randn(4, 3)
    p1, p2 = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    o1, o2 = FusedAdamW([p1], lr=1e-2), FusedAdamW([p2], lr=1e-2, capturable=True)
    for _ in range(3):
        g = torch.randn(4, 3)
        p1.grad, p2.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
    torch.testing.assert_close(p1, p2)
    o2.note_replayed_steps(2)
    assert int(o2.state_dict()["state"][0]["step"]) == 5
    with pytest.raises(ValueError, match="capturable"):
        GraphedTrainStep(torch.nn.Linear(2, 2), o1, 1, 4, device=torch.device("cpu"))


def test_init_on_device_meta_and_dtype():
    """Models can be built without host storage (meta) or directly in the training dtype."""
    from pytorch_distributed_amd.models import LlamaForCausalLM, get_config, init_on_device

    with init_on_device("meta"):
        big = LlamaForCausalLM(get_config("llama-7b"))
    assert all(p.is_meta for p in big.parameters())
    assert abs(sum(p.numel() for p in big.parameters()) - 6.738e9) < 1e7
    with init_on_device("cpu", torch.bfloat16):
        m = LlamaForCausalLM(get_config("llama-tiny"))
    assert all(p.dtype == torch.bfloat16 for p in m.parameters())
    assert torch.get_default_dtype() == torch.float32
    ids = torch.randint(0, 512, (2, 17))
    assert torch.isfinite(m(ids[:, :-1], ids[:, 1:]).float())


def test_rocprof_trace_merge_builds_per_rank_chrome_traces(tmp_path):
    """rocprofv3 kernel + marker CSVs -> rank{r}_trace.json: collectives on their own row, ranges as
    user annotations, usable by the analyzer (breakdown, overlap, ProfilerStep window)."""
    from pytorch_distributed_amd.utils import rocprof_trace, trace_analysis as ta

    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    kernel_hdr = '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp","Grid_Size_X","Workgroup_Size_X"\n'
    marker_hdr = '"Domain","Function","Process_Id","Thread_Id","Correlation_Id","Start_Timestamp","End_Timestamp"\n'
    for pid, rank in ((501, 1), (500, 0)):
        base = 1_000_000 * (rank + 1)
        (d / f"{pid}_kernel_trace.csv").write_text(kernel_hdr + "".join([
            f'"KERNEL_DISPATCH","Agent 1",1,0,{pid},1,10,"Cijk_gemm",1,{base + 100_000},{base + 140_000},65536,256\n',
            f'"KERNEL_DISPATCH","Agent 1",2,3,{pid},2,11,"ncclDevKernel_Generic_4(ncclDevKernelArgs)",2,{base + 130_000},{base + 170_000},4096,256\n',
            f'"KERNEL_DISPATCH","Agent 1",1,0,{pid},3,12,"pda::attn_fwd_kernel",3,{base + 150_000},{base + 170_000},65536,256\n',
            f'"KERNEL_DISPATCH","Agent 1",1,0,{pid},4,12,"pda::attn_fwd_kernel",4,{base + 900_000},{base + 910_000},65536,256\n']))
        (d / f"{pid}_marker_api_trace.csv").write_text(marker_hdr + "".join([
            f'"MARKER_CORE_API","pda.rank={rank}",{pid},{pid},1,{base},{base + 10}\n',
            f'"MARKER_CORE_API","ProfilerStep#4",{pid},{pid},2,{base + 100_000},{base + 200_000}\n',
            f'"MARKER_CORE_API","forward",{pid},{pid},3,{base + 100_000},{base + 120_000}\n',
            f'"MARKER_CORE_API","all_gather wait",{pid},{pid + 7},4,{base + 125_000},{base + 126_000}\n']))
    written = rocprof_trace.merge(str(tmp_path / "prof"), str(tmp_path / "traces" / "ddp"), only_profiler_steps=True)
    assert sorted(written) == [0, 1] and written[1].endswith("rank1_trace.json")
    ev = ta.load_trace(written[0])
    kernels = [e for e in ev if e.get("cat") == "kernel"]
    assert len(kernels) == 3  # the kernel after the profiler window was dropped
    assert {e["tid"] for e in kernels if "nccl" in e["name"]} == {rocprof_trace.COLLECTIVE_ROW}
    assert {e["tid"] for e in kernels if "nccl" not in e["name"]} == {100}
    names = {e["name"] for e in ev if e.get("cat") == "user_annotation"}
    assert names == {"ProfilerStep#4", "forward", "all_gather wait"}
    o = ta.comm_comp_overlap(ev, step=4)
    assert o["comm_us"] == 40.0 and o["overlapped_us"] == 30.0  # 130-140 with the GEMM, 150-170 with attention
    b = ta.temporal_breakdown(ev, step=4)
    assert b["compute_us"] == 60.0 and b["non_compute_us"] == 10.0
    res = ta.analyze_dir(str(tmp_path / "traces" / "ddp"))
    assert set(res) == {0, 1}


def test_counter_profile_report_from_synthetic_passes(tmp_path):
    """Counter mode: per-dispatch rows of several rocprofv3 passes are averaged per kernel and turned
    into utilisation / stall / conflict / traffic figures."""
    import csv as _csv
    import importlib.util

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("counter_profile", os.path.join(repo, "counter_profile.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    kern = "void pda::(anonymous namespace)::wgrad_kernel4<0>(unsigned short const*, long)"

    def write(path, header, rows):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w", newline="") as f:
            w = _csv.writer(f)
            w.writerow(header)
            w.writerows(rows)

    sq = {"SQ_BUSY_CYCLES": 1000.0, "SQ_VALU_MFMA_BUSY_CYCLES": 2000.0, "SQ_WAVE_CYCLES": 8000.0,
          "SQ_WAIT_INST_ANY": 2000.0, "SQ_INSTS_VALU": 300.0, "SQ_INSTS_MFMA": 100.0,
          "SQ_LDS_BANK_CONFLICT": 5.0, "SQ_LDS_IDX_ACTIVE": 500.0}
    head = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    write(str(tmp_path / "sq" / "x" / "1_counter_collection.csv"), head,
          [[1, kern, c, v] for c, v in sq.items()] + [[2, kern, c, v] for c, v in sq.items()])
    write(str(tmp_path / "hbm_read" / "x" / "1_counter_collection.csv"), head, [[1, kern, "FETCH_SIZE", 2048.0]])
    write(str(tmp_path / "trace" / "x" / "1_kernel_trace.csv"), ["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"],
          [[1, kern, 1000, 101000], [2, kern, 200000, 300000]])
    rows = cp.collect(str(tmp_path), "wgrad")
    (name, c), = rows.items()
    assert name.startswith("wgrad_kernel4") and c["dispatches"] == 2 and abs(c["duration_us"] - 100.0) < 1e-9
    d = cp.derive(c, simd_hz=2000.0 / 100e-6 / 0.5)  # chosen so that 2000 busy cycles in 100 us are 50 %
    assert abs(d["mfma_busy"] - 0.5) < 1e-9 and abs(d["wait_share"] - 0.25) < 1e-9
    assert abs(d["valu_per_mfma"] - 3.0) < 1e-9 and abs(d["lds_conflict"] - 0.01) < 1e-9
    assert d["l2_hit"] is None and abs(d["hbm_read_mb"] - 2.0) < 1e-9
    assert abs(d["hbm_read_gbps"] - 2048 * 1024 / 100e-6 / 1e9) < 1e-6
    assert cp.collect(str(tmp_path), "attention") == {}
    # window between the last two bursts of a marker kernel: ids (12, 112]
    trace = [{"Dispatch_Id": str(i), "Kernel_Name": "adamw" if i in (10, 11, 12, 110, 111, 112) else "gemm"} for i in range(1, 120)]
    assert cp.window(trace, "adamw") == (12, 112)
    assert cp.window(trace, "") is None and cp.window(trace[:50], "adamw") is None
