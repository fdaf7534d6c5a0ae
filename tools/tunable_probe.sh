#!/bin/bash
# GEMM probe with and without PyTorch TunableOp (hipBLASLt / rocBLAS solution search).
set -e
cd "$(dirname "$0")/.."
B=${1:-32}
echo "== default heuristics, batch $B"
timeout -k 10 300 python tools/gemm_probe.py --batch $B
echo "== TunableOp, batch $B"
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0 \
PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_b${B}.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 \
timeout -k 10 900 python tools/gemm_probe.py --batch $B
