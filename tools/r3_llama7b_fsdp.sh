#!/bin/bash
# Llama-7B under FSDP FULL_SHARD on one MI355X (world size 1, RCCL collectives forced): train, write a
# sharded checkpoint, then a SECOND process loads it and continues. Logs, step times, peak memory and
# the save / load seconds go to gpurun_out/r3/; the 81 GB checkpoint itself lives under /tmp and is removed.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
OUT=gpurun_out/r3
CK=/tmp/pda_llama7b_ckpt
mkdir -p $OUT; rm -rf $CK
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; rm -rf $CK; exit $1;; esac; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
COMMON="--model llama-7b --sequence-length 2048 --init-device gpu --force-collectives --strategy FULL_SHARD \
  --synthetic-data --global-batch-size 8 --micro-batch-size 8 --log-every 1 --no-profile"
df -h /tmp | tail -1
timeout -k 10 900 $TR --master-port 29721 train_fsdp.py $COMMON --max-steps 5 --sharded-checkpoint $CK \
    > $OUT/llama7b_fsdp_train.log 2>&1
rc=$?; grep -E "step=|wrapped|Saved|completed|Error|error" $OUT/llama7b_fsdp_train.log | tail -12; fatal $rc train
[ $rc -ne 0 ] && { tail -30 $OUT/llama7b_fsdp_train.log; rm -rf $CK; exit 1; }
du -sh $CK | tee $OUT/llama7b_ckpt_size.txt; ls -la $CK | head
timeout -k 10 900 $TR --master-port 29731 train_fsdp.py $COMMON --max-steps 8 --resume-sharded $CK \
    > $OUT/llama7b_fsdp_resume.log 2>&1
rc=$?; grep -E "step=|Loaded|completed|Error|error" $OUT/llama7b_fsdp_resume.log | tail -12
rm -rf $CK
fatal $rc resume
exit $rc
