# Phase stamps of the weight-gradient kernel with n fragment reads / m LDS-DMA loads issued behind one MFMA
# (default 2 / 1). Build each variant first:
#   hipcc --offload-arch=gfx950 -O3 -DPDA_WGRAD_READS_PER_SLOT=n -DPDA_WGRAD_LOADS_PER_SLOT=m \
#     -I pytorch_distributed_amd/csrc/kernels -o tools/micro/bin/ws_n_m tools/micro/wgrad_stamp.hip
for v in 2_1 4_1 8_1 16_1 8_2 8_4 16_8 2_1; do
  echo "== reads_loads $v"
  timeout -k 10 60 tools/micro/bin/ws_$v | head -2 || exit 1
done
