"""Merged accumulation passes of the trainers: several of the loader's micro-batches as one forward /
backward. The optimizer must see the same step (up to summation order), the data position must advance by
the loader's micro-batches, and every rank must run the same passes."""
import os
import re

import pytest
import torch

from pytorch_distributed_amd.data import SyntheticTokenLoader
from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config
from pytorch_distributed_amd.optim import CosineAnnealingLR, FusedAdamW
from pytorch_distributed_amd.train import Trainer

from dist_utils import run_distributed


def _trainer(merge, steps=3, global_batch=8, save_every=None, ckpt=None, seed=0):
    torch.manual_seed(seed)
    cfg = get_config("gpt2-tiny").without_dropout()
    model = MyGPT2LMHeadModel(cfg)
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
    sched = CosineAnnealingLR(opt, T_max=10, eta_min=1e-4)
    tr = Trainer(model, opt, sched, max_steps=steps, global_batch_size=global_batch, micro_batch_size=2,
                 log_every_n_steps=1, merge_micro_batches=merge, save_every_n_steps=save_every,
                 checkpoint_dir=ckpt or "checkpoints")
    return tr, cfg


class CountingLoader:
    def __init__(self, inner):
        self.inner, self.drawn, self.shapes = inner, 0, []

    def __iter__(self):
        for item in self.inner:
            self.drawn += 1
            yield item


def _record_first_gradients(tr):
    """Clones of the gradients the first optimizer step sees (AdamW's update hides their scale)."""
    grads, inner = [], tr._optimizer_step

    def step():
        if not grads:
            grads.extend(p.grad.detach().clone() for p in tr.model.parameters())
        inner()
    tr._optimizer_step = step
    return grads


def _losses(out):
    return [float(m) for m in re.findall(r"loss=(\d+\.\d+)", out)]


@pytest.mark.parametrize("merge", [2, 4, "all"])
def test_merged_passes_take_the_same_steps(merge, capsys):
    plain, cfg = _trainer(1)
    plain_grads = _record_first_gradients(plain)
    plain.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=40))
    plain_losses = _losses(capsys.readouterr().out)
    merged, _ = _trainer(merge)
    merged_grads = _record_first_gradients(merged)
    loader = CountingLoader(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=40))
    seen = []
    step = merged.training_step
    merged.training_step = lambda x, y: (seen.append(tuple(x.shape)), step(x, y))[1]
    merged.train(loader)
    merged_losses = _losses(capsys.readouterr().out)
    g = 4 if merge == "all" else merge
    assert merged.grad_accumulation_steps == 4
    assert seen == [(2 * g, 16)] * (3 * 4 // g)
    assert merged.current_step == plain.current_step == 3
    assert merged.batch_count == plain.batch_count == 12     # counted in the loader's micro-batches
    assert loader.drawn == 12                                # nothing drawn beyond the last step
    assert len(plain_losses) == len(merged_losses) == 3
    for a, b in zip(plain_losses, merged_losses):
        assert abs(a - b) < 2e-4
    for a, b in zip(plain_grads, merged_grads):  # the same gradient, scale included
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-7)
    for (k, p), (_, q) in zip(plain.model.state_dict().items(), merged.model.state_dict().items()):
        # AdamW turns a gradient near zero into a step of either sign: a twentieth of lr x steps is the bound
        torch.testing.assert_close(p, q, rtol=1e-4, atol=1.5e-4, msg=lambda m: f"{k}: {m}")


def test_merge_must_divide_the_window():
    tr, cfg = _trainer(3)
    with pytest.raises(ValueError, match="must divide"):
        tr.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=8))


def test_auto_keeps_the_loaders_micro_batches_without_a_gpu(capsys):
    tr, cfg = _trainer("auto", steps=2)
    seen = []
    step = tr.training_step
    tr.training_step = lambda x, y: (seen.append(x.shape[0]), step(x, y))[1]
    tr.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=20))
    assert seen == [2] * 8
    assert "runs as given" in capsys.readouterr().out


def test_auto_decisions_from_a_measured_footprint(monkeypatch, capsys):
    """The decision logic with the allocator statistics replaced: 1 GiB per micro-batch. 12 GiB free fits the
    whole window of 4 and twice its footprint (recomputation off); 6 GiB fits the window but not twice it;
    3 GiB fits two micro-batches per pass."""
    gib = 2 ** 30
    for free, merge, relaxed in ((12 * gib, 4, True), (6 * gib, 4, False), (3 * gib, 2, False)):
        tr, cfg = _trainer("auto", steps=3)
        tr.relax_recompute = True
        assert tr.model.transformer.activation_checkpoint == "selective"
        monkeypatch.setattr(tr, "_free_bytes", lambda device, free=free: free)
        monkeypatch.setattr(tr, "_allocated_now", lambda device: 0)
        monkeypatch.setattr(tr, "_allocated_peak", lambda device: gib)
        seen = []
        step = tr.training_step
        tr.training_step = lambda x, y, seen=seen, step=step: (seen.append(x.shape[0]), step(x, y))[1]
        tr.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=40))
        assert seen == [2] * 4 + [2 * merge] * (2 * 4 // merge), (free, seen)
        assert (tr.model.transformer.activation_checkpoint == "none") == relaxed
        out = capsys.readouterr().out
        assert ("recomputation switched off" in out) == relaxed
        assert f"pass(es) of {2 * merge} sequences" in out


def test_checkpoint_position_and_resume_with_merged_passes(tmp_path):
    ck = str(tmp_path / "ck")
    straight, cfg = _trainer(4, steps=4, save_every=2, ckpt=ck)
    straight.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=40))
    state = torch.load(os.path.join(ck, "checkpoint_step_2.pt"), weights_only=True)
    assert state["step"] == 3 and state["data_state"]["batch_count"] == 12
    assert state["data_state"]["loader"] == {"batches": 12}
    resumed, _ = _trainer(4, steps=4, seed=5)
    resumed.load_checkpoint(os.path.join(ck, "checkpoint_step_2.pt"))
    resumed.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=40))
    assert resumed.current_step == 4 and resumed.batch_count == 16
    for (k, p), (_, q) in zip(straight.model.state_dict().items(), resumed.model.state_dict().items()):
        assert torch.equal(p, q), k


def test_a_window_entered_in_its_middle_is_not_merged():
    tr, cfg = _trainer(4, steps=2)
    tr.batch_count = 1  # as after loading a checkpoint written inside a window
    seen = []
    step = tr.training_step
    tr.training_step = lambda x, y: (seen.append(x.shape[0]), step(x, y))[1]
    tr.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, num_batches=40))
    assert seen == [2, 2, 2, 8]


# ------------------------------------------------------------------ two ranks
def _distributed_merge(rank, world, wrapper):
    from pytorch_distributed_amd.parallel import DDP, FSDP, ShardingStrategy
    from pytorch_distributed_amd.train import DistributedTrainer

    def build(merge):
        torch.manual_seed(3)
        cfg = get_config("gpt2-tiny").without_dropout()
        model = MyGPT2LMHeadModel(cfg)
        if wrapper == "ddp":
            model = DDP(model)
        else:
            for i, block in enumerate(model.transformer.h):
                model.transformer.h[i] = FSDP(block, sharding_strategy=ShardingStrategy.FULL_SHARD)
            model = FSDP(model, sharding_strategy=ShardingStrategy.FULL_SHARD)
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
        tr = DistributedTrainer(model, opt, None, max_steps=3, global_batch_size=16, micro_batch_size=2,
                                log_every_n_steps=100, merge_micro_batches=merge)
        return tr, cfg

    plain, cfg = build(1)
    assert plain.grad_accumulation_steps == 4
    plain.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, "cpu", seed=70 + rank))
    merged, _ = build(2)
    # the ranks agree on the smaller choice: rank 1 asks for 4, rank 0 for 2
    assert merged._agree(2 if rank == 0 else 4, rank == 0) == (2, False)
    merged.train(SyntheticTokenLoader(2, 16, cfg.vocab_size, "cpu", seed=70 + rank))
    assert merged.batch_count == plain.batch_count == 12
    a, b = plain.model.state_dict(), merged.model.state_dict()
    for k in a:
        torch.testing.assert_close(a[k], b[k], rtol=1e-4, atol=1.5e-4, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("wrapper", ["ddp", "fsdp"])
def test_merged_passes_under_data_parallel_wrappers(wrapper):
    run_distributed(_distributed_merge, world=2, args=(wrapper,))


def test_cli_flag_values():
    import argparse

    from pytorch_distributed_amd.utils import cli

    ap = cli.add_common_args(argparse.ArgumentParser())
    assert cli.resolve_merge(ap.parse_args([])) == "auto"            # the scripts' default
    assert cli.resolve_merge(ap.parse_args(["--merge-micro-batches", "off"])) == "off"
    assert cli.resolve_merge(ap.parse_args(["--merge-micro-batches", "ALL"])) == "all"
    assert cli.resolve_merge(ap.parse_args(["--merge-micro-batches", "2"])) == 2
    with pytest.raises(ValueError):
        cli.resolve_merge(ap.parse_args(["--merge-micro-batches", "some"]))
