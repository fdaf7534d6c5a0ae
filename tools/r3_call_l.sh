#!/bin/bash
# Merged accumulation passes: the affected GPU tests, then the train scripts with their own defaults
# (--merge-micro-batches auto) against --merge-micro-batches off, and the bench with its merged_window row.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 600 python -u -m pytest tests/test_bench_contract_gpu.py tests/test_shared_gpu_rehearsal_gpu.py -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests_l.log 2>&1
rc=$?; tail -5 $OUT/gpu_tests_l.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests_l.log | tail -30; exit 1; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for model in gpt2-medium gpt2-large; do
  for merge in auto off; do
    tag=${model}_merge_${merge}
    echo "== $tag"
    timeout -k 10 300 python train_baseline.py --synthetic-data --model $model --max-steps 13 --log-every 4 --no-profile --merge-micro-batches $merge > $OUT/train_baseline_$tag.log 2>&1; rc=$?; fatal $rc baseline_$tag
    timeout -k 10 300 $TR --master-port 29761 train_ddp.py --synthetic-data --model $model --max-steps 13 --log-every 4 --no-profile --force-collectives --merge-micro-batches $merge > $OUT/train_ddp_$tag.log 2>&1; rc=$?; fatal $rc ddp_$tag
    timeout -k 10 300 $TR --master-port 29771 train_fsdp.py --synthetic-data --model $model --max-steps 13 --log-every 4 --no-profile --force-collectives --merge-micro-batches $merge > $OUT/train_fsdp_$tag.log 2>&1; rc=$?; fatal $rc fsdp_$tag
    grep -h "step=12\|runs as\|Error" $OUT/train_baseline_$tag.log $OUT/train_ddp_$tag.log $OUT/train_fsdp_$tag.log
  done
done
echo "== bench"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $OUT/bench_merged_window.json 2> $OUT/bench_merged_window.err; rc=$?; fatal $rc bench
cat $OUT/bench_merged_window.json
