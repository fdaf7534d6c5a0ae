#!/bin/bash
# The train scripts with their own defaults (merge auto, recomputation auto) against the literal recipe
# (--merge-micro-batches off --activation-checkpoint selective), plus the script-level GPU tests.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/auto; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 600 python -u -m pytest tests/test_bench_contract_gpu.py tests/test_shared_gpu_rehearsal_gpu.py tests/test_parallel_gpu.py -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests.log 2>&1
rc=$?; tail -4 $OUT/gpu_tests.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests.log | tail -30; exit 1; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for model in gpt2-medium gpt2-large; do
  for mode in auto literal; do
    tag=${model}_${mode}
    extra=""; [ $mode = literal ] && extra="--merge-micro-batches off --activation-checkpoint selective"
    echo "== $tag"
    timeout -k 10 300 python train_baseline.py --synthetic-data --model $model --max-steps 13 --log-every 4 --no-profile $extra > $OUT/train_baseline_$tag.log 2>&1; rc=$?; fatal $rc baseline_$tag
    timeout -k 10 300 $TR --master-port 29761 train_ddp.py --synthetic-data --model $model --max-steps 13 --log-every 4 --no-profile --force-collectives $extra > $OUT/train_ddp_$tag.log 2>&1; rc=$?; fatal $rc ddp_$tag
    timeout -k 10 300 $TR --master-port 29771 train_fsdp.py --synthetic-data --model $model --max-steps 13 --log-every 4 --no-profile --force-collectives $extra > $OUT/train_fsdp_$tag.log 2>&1; rc=$?; fatal $rc fsdp_$tag
    grep -h "step=12\|runs as\|switched off\|Error" $OUT/train_baseline_$tag.log $OUT/train_ddp_$tag.log $OUT/train_fsdp_$tag.log
  done
done
