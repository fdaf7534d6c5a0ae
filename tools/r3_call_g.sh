#!/bin/bash
# A/B: fused gelu' epilogue in the backward only (PDA_FUSED_MLP=bwd) against the default, GPT-2 large 64 x 1.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],2), 'ms', round(100*r['mfu'],2), '%', round(r['peak_mem_gb'],1), 'GB')"; }
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "default        "
  rc=$?; fatal $rc bench
  PDA_FUSED_MLP=bwd timeout -k 10 300 python bench.py --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "fused backward "
  rc=$?; fatal $rc bench
  PDA_FUSED_MLP=1 timeout -k 10 300 python bench.py --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "fused both     "
  rc=$?; fatal $rc bench
done 2>&1 | tee $OUT/fused_mlp_ab.txt
