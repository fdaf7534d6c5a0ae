#!/bin/bash
# GPU smoke of the training scripts at world size 1 with forced collectives (run from the repo root on the GPU box).
set -o pipefail
export MASTER_ADDR=127.0.0.1
P=29610
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for st in FULL_SHARD SHARD_GRAD_OP NO_SHARD; do
  P=$((P+3))
  timeout -k 10 200 $TR --master-port $P train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 6 --log-every 2 --no-profile --force-collectives --strategy $st --fsdp-resident-gb auto --save-every 3 --checkpoint-dir gpurun_out/ck_$st > gpurun_out/train_fsdp_$st.log 2>&1 || { tail -20 gpurun_out/train_fsdp_$st.log; exit 1; }
  echo "== $st"; grep -E "step=|wrapped|completed" gpurun_out/train_fsdp_$st.log | tail -4
done
echo "== resume FULL_SHARD"
timeout -k 10 200 $TR --master-port 29641 train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 8 --log-every 2 --no-profile --force-collectives --strategy FULL_SHARD --resume latest --checkpoint-dir gpurun_out/ck_FULL_SHARD 2>&1 | grep -E "step=|esum|completed" | tail -5 || exit 1
echo "== ddp"
timeout -k 10 200 $TR --master-port 29651 train_ddp.py --synthetic-data --model gpt2-medium --max-steps 4 --log-every 2 --no-profile --force-collectives 2>&1 | grep -E "step=|completed" | tail -3 || exit 1
rm -rf gpurun_out/ck_*
