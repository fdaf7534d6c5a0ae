#!/bin/bash
# The reference's workflow with the scripts' own defaults (GPT-2 Large, 20 steps, profiler on), then the trace analysis.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/workflow; mkdir -p $OUT; cd $R; rm -rf outputs/traces
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 400 python train_baseline.py --synthetic-data > $OUT/train_baseline.log 2>&1; rc=$?; fatal $rc baseline; [ $rc -ne 0 ] && { tail -20 $OUT/train_baseline.log; exit 1; }
timeout -k 10 400 $TR --master-port 29961 train_ddp.py --synthetic-data --force-collectives > $OUT/train_ddp.log 2>&1; rc=$?; fatal $rc ddp; [ $rc -ne 0 ] && { tail -20 $OUT/train_ddp.log; exit 1; }
for s in FULL_SHARD SHARD_GRAD_OP; do
  timeout -k 10 400 $TR --master-port 29962 train_fsdp.py --synthetic-data --force-collectives --strategy $s > $OUT/train_fsdp_$s.log 2>&1; rc=$?; fatal $rc fsdp_$s; [ $rc -ne 0 ] && { tail -20 $OUT/train_fsdp_$s.log; exit 1; }
done
grep -h "step=10\|Trace\|trace" $OUT/train_*.log | cut -c1-150
ls -la outputs/traces/*/ | head -20
timeout -k 10 300 python analyze_traces.py --plots $OUT/plots > $OUT/analyze_traces.txt 2>&1; rc=$?; head -12 $OUT/analyze_traces.txt; exit $rc
