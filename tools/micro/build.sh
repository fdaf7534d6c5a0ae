#!/bin/bash
# Builds the stand-alone probes into tools/micro/bin/ (git-ignored; the binaries travel to the GPU box with the tree).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/bin
for p in stream_probe valu_probe feed_probe mfma_probe coexec_probe; do
  hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/$p tools/micro/$p.hip
done
hipcc --offload-arch=gfx950 -O3 -I pytorch_distributed_amd/csrc/kernels -o tools/micro/bin/wgrad_stamp tools/micro/wgrad_stamp.hip
for p in attn_stamp attn_bwd_time; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pytorch_distributed_amd/csrc/kernels -I pytorch_distributed_amd/csrc -o tools/micro/bin/$p tools/micro/$p.hip
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pytorch_distributed_amd/csrc/kernels -o tools/micro/bin/quad_transpose_check tools/micro/quad_transpose_check.hip
