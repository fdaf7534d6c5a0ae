#!/bin/bash
# The real shard readers (mmap loader, rank-partitioned loader, native prefetch thread) under the train scripts on the GPU.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/shards; mkdir -p $OUT; cd $R
D=/tmp/pda_shards; CK=/tmp/pda_shards_ck; rm -rf $D $CK
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
C="--model gpt2 --max-steps 6 --log-every 1 --no-profile --global-batch-size 16 --micro-batch-size 4 --sequence-length 512 --make-synthetic-shards --data-dir $D --num-train-files 2"
check() { local rc=$1 name=$2 want=$3; case $rc in 124|134|137|139) echo "fatal exit $rc in $name"; exit $rc;; esac
  local n=$(grep -c "loss=" $OUT/$name.log); echo "$name: rc=$rc losses=$n $(grep 'loss=' $OUT/$name.log | tail -1 | cut -c1-60)"
  [ $rc -ne 0 -o "$n" != "$want" ] && { tail -15 $OUT/$name.log; exit 1; }; return 0; }
timeout -k 10 240 python train_baseline.py $C > $OUT/baseline.log 2>&1; check $? baseline 6
timeout -k 10 240 $TR --master-port 29951 train_ddp.py $C --force-collectives > $OUT/ddp.log 2>&1; check $? ddp 6
timeout -k 10 240 $TR --master-port 29952 train_ddp.py $C --force-collectives --prefetch-data > $OUT/ddp_prefetch.log 2>&1; check $? ddp_prefetch 6
timeout -k 10 240 $TR --master-port 29953 train_fsdp.py $C --force-collectives --prefetch-data --save-every 2 --checkpoint-dir $CK > $OUT/fsdp_prefetch.log 2>&1; check $? fsdp_prefetch 6
timeout -k 10 240 $TR --master-port 29954 train_fsdp.py $C --force-collectives --prefetch-data --resume latest --checkpoint-dir $CK --max-steps 8 > $OUT/fsdp_resume.log 2>&1; check $? fsdp_resume 3
grep -h "Loaded" $OUT/fsdp_resume.log
# the same token stream with and without the prefetch thread: same losses
diff <(grep -o "loss=[0-9.]*" $OUT/ddp.log) <(grep -o "loss=[0-9.]*" $OUT/ddp_prefetch.log) && echo "prefetch: same losses"
rm -rf $D $CK
echo "shards ok"
