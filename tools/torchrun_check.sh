#!/bin/bash
# A/B of the wrapped one-GPU benchmark started directly and through torch.distributed.run (which
# exports OMP_NUM_THREADS=1 unless it is already set): tools/torchrun_check.sh <model>
MODEL=${1:-gpt2}
run() { "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611"
for mode in fsdp ddp; do
  echo "$MODEL $mode direct:            $(run timeout -k 10 200 python bench.py --model $MODEL --wrap-single --mode $mode --steps 10 --warmup 3)"
  echo "$MODEL $mode torchrun:          $(run timeout -k 10 200 $TR bench.py --gpus 1 --model $MODEL --wrap-single --mode $mode --steps 10 --warmup 3)"
  echo "$MODEL $mode torchrun, OMP=8:   $(OMP_NUM_THREADS=8 run timeout -k 10 200 $TR bench.py --gpus 1 --model $MODEL --wrap-single --mode $mode --steps 10 --warmup 3)"
done
