#!/bin/bash
# GPU call: GPU tests (segmented weight-gradient kernel, deferred weight gradients,
# cached AdamW lists), then A/B of the reference's recipe (4 micro-batches of 8, selective) with the
# weight gradients formed per micro-batch and once per window, and its kernel breakdown.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3
mkdir -p $OUT
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests_c.log 2>&1
rc=$?; tail -5 $OUT/gpu_tests_c.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests_c.log | tail -30; exit 1; }

echo "== reference recipe (32 = 4 x 8, selective): per micro-batch vs once per window"
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],2), 'ms', round(100*r['mfu'],2), '%', round(r['peak_mem_gb'],1), 'GB')"; }
for i in 1 2; do
  PDA_DEFER_WGRAD=0 timeout -k 10 300 python bench.py --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
      --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "per micro-batch"
  rc=$?; fatal $rc bench_ab
  timeout -k 10 300 python bench.py --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
      --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "once per window"
  rc=$?; fatal $rc bench_ab
done 2>&1 | tee $OUT/defer_wgrad_ab.txt

echo "== kernel breakdown: reference recipe, weight gradients once per window"
cd /tmp
rm -rf $OUT/prof_ref && mkdir -p $OUT/prof_ref
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ref/prof -- python3 $R/bench.py \
    --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective --no-reference-recipe --steps 3 --warmup 2 \
    > $OUT/prof_ref/bench.json 2> $OUT/prof_ref/prof.log
rc=$?; fatal $rc rocprof_ref
python3 $R/tools/step_breakdown.py $OUT/prof_ref/prof/*/*_kernel_trace.csv 2 4 > $OUT/step_breakdown_reference_recipe_deferred.txt
head -28 $OUT/step_breakdown_reference_recipe_deferred.txt
python3 $R/tools/kernel_avg.py $OUT/prof_ref/prof/*/*_kernel_trace.csv 8 > $OUT/kernel_avg_reference_recipe_deferred.txt 2>&1 || true
cp $OUT/prof_ref/prof/*/*_kernel_stats.csv $OUT/kernel_stats_reference_recipe_deferred.csv 2>/dev/null
rm -rf $OUT/prof_ref/prof
cd $R

echo "== full bench (default + reference recipe)"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $OUT/bench_default_c.json 2> $OUT/bench_default_c.err
rc=$?; fatal $rc bench; cat $OUT/bench_default_c.json
exit 0
