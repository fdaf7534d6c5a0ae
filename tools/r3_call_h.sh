#!/bin/bash
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests_h.log 2>&1
rc=$?; tail -5 $OUT/gpu_tests_h.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests_h.log | tail -30; exit 1; }
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],2), 'ms', round(100*r['mfu'],2), '%', round(r['peak_mem_gb'],1), 'GB')"; }
for i in 1 2; do
  timeout -k 10 300 python bench.py --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
      --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "reference recipe, large "
  rc=$?; fatal $rc bench_ab
  timeout -k 10 300 python bench.py --model gpt2-medium --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
      --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "reference recipe, medium"
  rc=$?; fatal $rc bench_ab
done 2>&1 | tee $OUT/reference_recipe_after_transpose_cache.txt
