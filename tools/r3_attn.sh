#!/bin/bash
# Attention kernel iteration: phase stamps of the dQ kernel, kernel times, and the numerics tests.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
TAG=${1:-x}
timeout -k 10 120 tools/micro/bin/attn_stamp > $OUT/attn_stamp_$TAG.log 2>&1 || { tail -5 $OUT/attn_stamp_$TAG.log; exit 1; }
cat $OUT/attn_stamp_$TAG.log
timeout -k 10 120 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_time_$TAG.log || exit 1
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "attention" 2>&1 | tail -3
