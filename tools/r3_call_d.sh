#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
timeout -k 10 300 python -u tools/micro/host_gap_probe.py > $OUT/host_gap_probe.log 2>&1
rc=$?; tail -60 $OUT/host_gap_probe.log; exit $rc
