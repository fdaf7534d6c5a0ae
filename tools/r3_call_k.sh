#!/bin/bash
# Kernel breakdown of the reference's recipe under the FSDP FULL_SHARD wrapper (world size 1, forced collectives).
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29801 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT
cd /tmp
rm -rf $OUT/prof_fsdp && mkdir -p $OUT/prof_fsdp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fsdp/prof -- python3 $R/bench.py \
    --gpus 1 --wrap-single --mode fsdp --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective \
    --no-reference-recipe --steps 3 --warmup 2 > $OUT/prof_fsdp/bench.json 2> $OUT/prof_fsdp/prof.log
rc=$?; [ $rc -ne 0 ] && { tail -5 $OUT/prof_fsdp/prof.log; exit $rc; }
python3 $R/tools/step_breakdown.py $OUT/prof_fsdp/prof/*/*_kernel_trace.csv 2 4 > $OUT/step_breakdown_reference_recipe_fsdp.txt
cat $OUT/step_breakdown_reference_recipe_fsdp.txt | head -40
rm -rf $OUT/prof_fsdp/prof
