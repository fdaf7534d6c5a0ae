"""User-level API smoke on one GPU: the model with a stock torch optimizer inside the Trainer, eval / no_grad
forward (bare and FSDP-wrapped at world size 1), save -> from_pretrained, logits vs fused loss.
usage: python tools/api_smoke.py"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_amd import ops  # noqa: E402
from pytorch_distributed_amd.data import SyntheticTokenLoader  # noqa: E402
from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config  # noqa: E402
from pytorch_distributed_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda:0")
cfg = get_config("gpt2", n_layer=4).without_dropout()
torch.manual_seed(0)
model = MyGPT2LMHeadModel(cfg).to(device=dev, dtype=torch.bfloat16)

# 1. a stock optimizer (no fp32 master, grads set to None by zero_grad) under the Trainer with accumulation
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.1)
tr = Trainer(model, opt, None, max_steps=4, global_batch_size=8, micro_batch_size=2, log_every_n_steps=1,
             merge_micro_batches="auto", relax_recompute=True)
tr.train(SyntheticTokenLoader(2, 256, cfg.vocab_size, dev, seed=1))
assert tr.current_step == 4 and tr.batch_count >= 4 * 1

# 2. eval forward: logits, and their cross-entropy equals the fused loss
model.eval()
ids = torch.randint(0, cfg.vocab_size, (2, 257), device=dev)
with torch.no_grad():
    logits = model(ids[:, :-1])
    fused = model(ids[:, :-1], ids[:, 1:])
assert logits.shape == (2, 256, cfg.vocab_size), logits.shape
ce = torch.nn.functional.cross_entropy(logits.float().view(-1, cfg.vocab_size), ids[:, 1:].reshape(-1))
assert abs(ce.item() - fused.item()) < 2e-2, (ce.item(), fused.item())

# 3. save -> from_pretrained gives the same logits
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "model.pt")
    model.save(path)
    again = MyGPT2LMHeadModel.from_pretrained(torch.load(path, weights_only=True), cfg).to(device=dev, dtype=torch.bfloat16).eval()
with torch.no_grad():
    assert torch.equal(again(ids[:, :-1]), logits)

# 4. the same under the FSDP wrapper at world size 1: eval forward without a backward, then training goes on
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
from pytorch_distributed_amd.parallel import FSDP, ShardingStrategy, init_distributed, shutdown  # noqa: E402
init_distributed()
for i, blk in enumerate(again.transformer.h):
    again.transformer.h[i] = FSDP(blk, sharding_strategy=ShardingStrategy.FULL_SHARD, force_collectives=True)
wrapped = FSDP(again, sharding_strategy=ShardingStrategy.FULL_SHARD, force_collectives=True)
wrapped.eval()
with torch.no_grad():
    w_logits = wrapped(ids[:, :-1])
assert (w_logits.float() - logits.float()).abs().max().item() < 5e-2
wrapped.train()
loss = wrapped(ids[:, :-1], ids[:, 1:])
loss.backward()
with wrapped.no_sync():
    wrapped(ids[:, :-1], ids[:, 1:]).backward()
wrapped(ids[:, :-1], ids[:, 1:]).backward()
sd = wrapped.state_dict()
assert set(sd) == set(model.state_dict()), "FSDP state_dict keys differ from the bare model's"
shutdown()
print("api smoke ok: losses", round(ce.item(), 4), round(fused.item(), 4), round(loss.item(), 4))
