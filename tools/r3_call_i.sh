#!/bin/bash
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests_i.log 2>&1
rc=$?; tail -5 $OUT/gpu_tests_i.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests_i.log | tail -30; exit 1; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
echo "== train scripts with their own defaults (gpt2-medium, global 32 = 4 x 8, selective, dropout 0.1)"
timeout -k 10 300 python train_baseline.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile > $OUT/train_baseline_defaults.log 2>&1; rc=$?; fatal $rc baseline
timeout -k 10 300 $TR --master-port 29761 train_ddp.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile --force-collectives > $OUT/train_ddp_defaults.log 2>&1; rc=$?; fatal $rc ddp
timeout -k 10 300 $TR --master-port 29771 train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile --force-collectives > $OUT/train_fsdp_defaults.log 2>&1; rc=$?; fatal $rc fsdp
grep -h "step=\|fp32\|Error" $OUT/train_*_defaults.log | tail -14
echo "== same with dropout 0"
timeout -k 10 300 python train_baseline.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile --dropout 0 > $OUT/train_baseline_dropout0.log 2>&1; rc=$?; fatal $rc baseline0
timeout -k 10 300 $TR --master-port 29781 train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile --force-collectives --dropout 0 > $OUT/train_fsdp_dropout0.log 2>&1; rc=$?; fatal $rc fsdp0
grep -h "step=" $OUT/train_*_dropout0.log | tail -6
