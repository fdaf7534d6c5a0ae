#!/bin/bash
# Longer runs of the scripts with their defaults: 120 steps with checkpoints, then a resumed run to 150.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/soak; mkdir -p $OUT; cd $R
CK=/tmp/pda_soak_ckpt; rm -rf $CK
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; rm -rf $CK; exit $1;; esac; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
timeout -k 10 500 $TR --master-port 29741 train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 120 --log-every 20 --no-profile --force-collectives --save-every 50 --checkpoint-dir $CK > $OUT/fsdp_120.log 2>&1; rc=$?; fatal $rc fsdp120
grep -h "step=\|runs as\|switched\|Saved\|Error" $OUT/fsdp_120.log | tail -12
[ $rc -ne 0 ] && { tail -20 $OUT/fsdp_120.log; rm -rf $CK; exit 1; }
timeout -k 10 500 $TR --master-port 29742 train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 150 --log-every 10 --no-profile --force-collectives --resume latest --checkpoint-dir $CK > $OUT/fsdp_resume_150.log 2>&1; rc=$?; fatal $rc fsdp150
grep -h "step=\|Loaded\|runs as\|switched\|Error" $OUT/fsdp_resume_150.log | tail -10
rm -rf $CK
[ $rc -ne 0 ] && { tail -20 $OUT/fsdp_resume_150.log; exit 1; }
timeout -k 10 500 python train_baseline.py --synthetic-data --max-steps 100 --log-every 20 --no-profile > $OUT/baseline_large_100.log 2>&1; rc=$?; fatal $rc baseline
grep -h "step=\|runs as\|switched\|Error" $OUT/baseline_large_100.log | tail -8
exit $rc
