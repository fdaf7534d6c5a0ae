#!/bin/bash
# Option matrix of the train scripts on one GPU (GPT-2 small, 6 steps each): every run must exit 0 and log 6 losses.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/matrix; mkdir -p $OUT; cd $R
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
C="--synthetic-data --model gpt2 --max-steps 6 --log-every 1 --no-profile --global-batch-size 16 --micro-batch-size 4 --sequence-length 512"
n=0; port=29800
run() { # name, command...
  local name=$1; shift
  n=$((n+1)); port=$((port+1))
  timeout -k 10 240 "$@" > $OUT/$name.log 2>&1; local rc=$?
  case $rc in 124|134|137|139) echo "fatal exit $rc in $name"; exit $rc;; esac
  local losses=$(grep -c "loss=" $OUT/$name.log)
  local last=$(grep "loss=" $OUT/$name.log | tail -1 | sed 's/.*loss=\([0-9.]*\).*mfu=\([0-9.]*%\).*/loss \1 mfu \2/')
  echo "$name: rc=$rc losses=$losses $last $(grep -o 'runs as [^[]*' $OUT/$name.log | head -1)$(grep -q 'switched off' $OUT/$name.log && echo ' recompute-off')"
  [ $rc -ne 0 -o "$losses" != "6" ] && { tail -15 $OUT/$name.log; exit 1; }
  return 0
}
run baseline_default python train_baseline.py $C
run baseline_full_recompute python train_baseline.py $C --activation-checkpoint full
run baseline_merge2 python train_baseline.py $C --merge-micro-batches 2 --activation-checkpoint selective
run baseline_fp32 python train_baseline.py $C --dtype fp32
run baseline_param_accum python train_baseline.py $C --grad-accum-dtype param --merge-micro-batches off
run baseline_llama python train_baseline.py $C --model llama-tiny --sequence-length 128
run ddp_default $TR --master-port $port train_ddp.py $C --force-collectives
run ddp_decompose $TR --master-port $port train_ddp.py $C --force-collectives --ddp-decompose on --merge-micro-batches off
run ddp_reduce_fp32 $TR --master-port $port train_ddp.py $C --force-collectives --reduce-fp32
run ddp_c10d $TR --master-port $port train_ddp.py $C --force-collectives --backend c10d
for s in FULL_SHARD SHARD_GRAD_OP NO_SHARD; do
  run fsdp_$s $TR --master-port $port train_fsdp.py $C --force-collectives --strategy $s
done
run fsdp_units3_resident $TR --master-port $port train_fsdp.py $C --force-collectives --fsdp-unit-blocks 3 --fsdp-resident-gb auto --merge-micro-batches off
run fsdp_full_recompute $TR --master-port $port train_fsdp.py $C --force-collectives --activation-checkpoint full
run fsdp_dual $TR --master-port $port train_fsdp.py $C --force-collectives --comm-dual
echo "matrix ok: $n runs"
