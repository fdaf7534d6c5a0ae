#!/bin/bash
# FSDP FULL_SHARD wrapper at world size 1 with forced collectives: the reference's recipe (4 micro-batches of 8)
# with and without the resident budget (units stay gathered through the no_sync window).
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read()); f=r['modes']['fsdp']; print('$1', round(f['ms_per_step'],2), 'ms', round(100*f['mfu'],2), '%', 'resident', f['resident_gb'], '| all-gathers/step n/a')"; }
for model in gpt2-large gpt2-medium; do
for i in 1 2; do
  timeout -k 10 300 $TR --master-port 2975$i bench.py --gpus 1 --model $model --wrap-single --mode fsdp --batch-per-gpu 32 --micro-batch 8 \
      --activation-checkpoint selective --no-reference-recipe --steps 8 --warmup 3 2>/dev/null | show "$model re-gather every pass "
  rc=$?; fatal $rc bench
  timeout -k 10 300 $TR --master-port 2976$i bench.py --gpus 1 --model $model --wrap-single --mode fsdp --batch-per-gpu 32 --micro-batch 8 \
      --activation-checkpoint selective --no-reference-recipe --steps 8 --warmup 3 --fsdp-resident-gb auto 2>/dev/null | show "$model resident (auto)      "
  rc=$?; fatal $rc bench
done; done 2>&1 | tee $OUT/fsdp_window_residency_ab.txt
