#!/bin/bash
# Final check of the tree: GPU suite, smoke entry, driver-style bench.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3_final; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $OUT/gpu_tests_final.log 2>&1
rc=$?; tail -4 $OUT/gpu_tests_final.log; fatal $rc pytest
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/gpu_tests_final.log | tail -30; exit 1; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1; rc=$?; tail -2 $OUT/smoke_final.log; fatal $rc smoke
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; rc=$?; fatal $rc bench
cat $OUT/bench_driver_style.json
