#!/bin/bash
# GPU call: GPU tests again (new tests), the plain-command traces, kernel breakdown
# of the reference's recipe (4 micro-batches of 8, selective) next to the default step, Llama-7B FSDP.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3
mkdir -p $OUT
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests.log 2>&1
rc=$?; tail -5 $OUT/gpu_tests.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests.log | tail -30; }

echo "== plain-command traces"
rm -rf $OUT/traces
timeout -k 10 300 python train_baseline.py --synthetic-data --model gpt2 --max-steps 3 --log-every 1 \
    --trace-dir $OUT/traces/baseline > $OUT/train_baseline_trace.log 2>&1
rc=$?; tail -2 $OUT/train_baseline_trace.log; fatal $rc train_baseline
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
timeout -k 10 300 $TR --master-port 29701 train_ddp.py --synthetic-data --model gpt2 --max-steps 3 --log-every 1 \
    --force-collectives --trace-dir $OUT/traces/ddp > $OUT/train_ddp_trace.log 2>&1
rc=$?; tail -2 $OUT/train_ddp_trace.log; fatal $rc train_ddp
for st in FULL_SHARD SHARD_GRAD_OP; do
  lower=$(echo $st | tr A-Z a-z)
  timeout -k 10 300 $TR --master-port 29711 train_fsdp.py --synthetic-data --model gpt2 --max-steps 3 --log-every 1 \
      --force-collectives --strategy $st --trace-dir $OUT/traces/fsdp_$lower > $OUT/train_fsdp_${lower}_trace.log 2>&1
  rc=$?; tail -2 $OUT/train_fsdp_${lower}_trace.log; fatal $rc train_fsdp
done
timeout 120 python analyze_traces.py --traces $OUT/traces --step 5 > $OUT/analyze_traces.txt 2>&1; head -12 $OUT/analyze_traces.txt
ls -la $OUT/traces/*/
gzip -f $OUT/traces/*/*.json

echo "== kernel breakdown: reference recipe"
cd /tmp
rm -rf $OUT/prof_ref && mkdir -p $OUT/prof_ref
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ref/prof -- python3 $R/bench.py \
    --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective --no-reference-recipe --steps 3 --warmup 2 \
    > $OUT/prof_ref/bench.json 2> $OUT/prof_ref/prof.log
rc=$?; fatal $rc rocprof_ref
python3 $R/tools/step_breakdown.py $OUT/prof_ref/prof/*/*_kernel_trace.csv 2 4 > $OUT/step_breakdown_reference_recipe.txt
cat $OUT/step_breakdown_reference_recipe.txt | head -24
python3 $R/tools/kernel_avg.py $OUT/prof_ref/prof/*/*_kernel_trace.csv 8 > $OUT/kernel_avg_reference_recipe.txt 2>&1 || true
cp $OUT/prof_ref/prof/*/*_kernel_stats.csv $OUT/kernel_stats_reference_recipe.csv 2>/dev/null
rm -rf $OUT/prof_ref/prof

echo "== kernel breakdown: default step (chunked LM head)"
rm -rf $OUT/prof_def && mkdir -p $OUT/prof_def
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_def/prof -- python3 $R/bench.py \
    --no-reference-recipe --steps 3 --warmup 2 > $OUT/prof_def/bench.json 2> $OUT/prof_def/prof.log
rc=$?; fatal $rc rocprof_def
python3 $R/tools/step_breakdown.py $OUT/prof_def/prof/*/*_kernel_trace.csv 2 8 > $OUT/step_breakdown.txt
head -24 $OUT/step_breakdown.txt
cp $OUT/prof_def/prof/*/*_kernel_stats.csv $OUT/kernel_stats_default.csv 2>/dev/null
rm -rf $OUT/prof_def/prof
cd $R

echo "== bench default vs unchunked LM head (A/B, same box)"
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('chunked  ', round(r['ms_per_step'],2), 'ms', round(100*r['mfu'],2), '%', round(r['peak_mem_gb'],1), 'GB')"
  PDA_LM_HEAD_CHUNK_ROWS=1000000 timeout -k 10 300 python bench.py --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('one piece', round(r['ms_per_step'],2), 'ms', round(100*r['mfu'],2), '%', round(r['peak_mem_gb'],1), 'GB')"
done 2>&1 | tee $OUT/lm_head_chunk_ab.txt

echo "== Llama-7B FSDP FULL_SHARD with sharded checkpoint save / load"
bash tools/r3_llama7b_fsdp.sh
exit 0
