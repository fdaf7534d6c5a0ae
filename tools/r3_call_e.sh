#!/bin/bash
# Exhaustive hipBLASLt kernel search at the reference recipe's micro-batch (8 x 1024 tokens), then the
# reference recipe with and without the plan file it wrote.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 500 python -u tools/tune_gemms.py --spec gpt2-large:8 --out $OUT/plans_mb8.json > $OUT/gemm_exhaustive_mb8.log 2>&1
rc=$?; tail -25 $OUT/gemm_exhaustive_mb8.log; fatal $rc tune; [ $rc -ne 0 ] && exit $rc
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],2), 'ms', round(100*r['mfu'],2), '%', round(r['peak_mem_gb'],1), 'GB')"; }
for i in 1 2; do
  timeout -k 10 300 python bench.py --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
      --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "timed in place  "
  rc=$?; fatal $rc bench
  PDA_GEMM_PLAN_FILE=$OUT/plans_mb8.json timeout -k 10 300 python bench.py --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
      --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "exhaustive plan "
  rc=$?; fatal $rc bench
done 2>&1 | tee $OUT/gemm_plan_mb8_ab.txt
