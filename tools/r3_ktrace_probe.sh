#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3; mkdir -p $OUT
for m in null adds side; do
  timeout -k 10 120 python tools/micro/ktrace_probe.py $m 2>&1 | grep -E "RESULT|mode|COUNTS" || exit 1
done
echo "== under torchrun"
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29761 tools/micro/ktrace_probe.py null 2>&1 | grep -E "RESULT|mode|COUNTS" || exit 1
