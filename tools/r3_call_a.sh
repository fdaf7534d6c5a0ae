#!/bin/bash
# GPU call (run from the repo root on the GPU box): GPU tests, the in-process kernel
# tracer through the plain training commands, then the two benchmark lines. Every GPU step runs
# under its own time limit; a step that ends by signal / abort / time-out ends the script.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
OUT=gpurun_out/r3
mkdir -p $OUT
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }

timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests.log 2>&1
rc=$?; tail -5 $OUT/gpu_tests.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests.log | tail -30; }

echo "== in-process kernel trace: python train_baseline.py (no external profiler)"
timeout -k 10 300 python train_baseline.py --synthetic-data --model gpt2 --max-steps 3 --log-every 1 \
    --trace-dir $OUT/traces/baseline > $OUT/train_baseline_trace.log 2>&1
rc=$?; tail -4 $OUT/train_baseline_trace.log; fatal $rc train_baseline
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
timeout -k 10 300 $TR --master-port 29701 train_ddp.py --synthetic-data --model gpt2 --max-steps 3 --log-every 1 \
    --force-collectives --trace-dir $OUT/traces/ddp > $OUT/train_ddp_trace.log 2>&1
rc=$?; tail -3 $OUT/train_ddp_trace.log; fatal $rc train_ddp
timeout -k 10 300 $TR --master-port 29711 train_fsdp.py --synthetic-data --model gpt2 --max-steps 3 --log-every 1 \
    --force-collectives --strategy FULL_SHARD --trace-dir $OUT/traces/fsdp_full_shard > $OUT/train_fsdp_trace.log 2>&1
rc=$?; tail -3 $OUT/train_fsdp_trace.log; fatal $rc train_fsdp
ls -la $OUT/traces/*/ 
timeout 120 python analyze_traces.py --traces $OUT/traces --step 5 > $OUT/analyze_traces.txt 2>&1; head -30 $OUT/analyze_traces.txt
# traces are large: keep the analysis and the baseline trace only
rm -f $OUT/traces/ddp/*.json.keep 2>/dev/null

echo "== bench default (headline + reference recipe)"
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
rc=$?; cat $OUT/bench_default.json; fatal $rc bench; [ $rc -ne 0 ] && tail -20 $OUT/bench_default.err
echo "== bench gpt2-medium wrapped (DDP + FSDP at world size 1, forced collectives)"
timeout -k 10 600 python bench.py --model gpt2-medium --wrap-single --mode both --steps 10 --warmup 3 \
    > $OUT/bench_medium_wrapped.json 2> $OUT/bench_medium_wrapped.err
rc=$?; cat $OUT/bench_medium_wrapped.json; fatal $rc bench_medium; [ $rc -ne 0 ] && tail -20 $OUT/bench_medium_wrapped.err
exit 0
