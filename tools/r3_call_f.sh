#!/bin/bash
# GPT-2 medium: DDP / FSDP wrappers at world size 1 with forced collectives (+ reference recipe), and
# the kernel breakdown of the FSDP-wrapped step.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
fatal() { case $1 in 124|134|137|139) echo "fatal exit $1 in $2"; exit $1;; esac; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
timeout -k 10 400 $TR --master-port 29752 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode both --model gpt2-medium \
    > $OUT/bench_medium_wrapped_c.json 2> $OUT/bench_medium_wrapped_c.err
rc=$?; fatal $rc bench_medium; cat $OUT/bench_medium_wrapped_c.json; tail -3 $OUT/bench_medium_wrapped_c.err
timeout -k 10 300 python bench.py --model gpt2-medium --steps 10 --warmup 3 > $OUT/bench_medium_single_c.json 2> $OUT/bench_medium_single_c.err
rc=$?; fatal $rc bench_medium_single; cat $OUT/bench_medium_single_c.json
cd /tmp
rm -rf $OUT/prof_med && mkdir -p $OUT/prof_med
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_med/prof -- python3 $R/bench.py \
    --model gpt2-medium --no-reference-recipe --steps 3 --warmup 2 > $OUT/prof_med/bench.json 2> $OUT/prof_med/prof.log
rc=$?; fatal $rc rocprof_med
python3 $R/tools/step_breakdown.py $OUT/prof_med/prof/*/*_kernel_trace.csv 2 > $OUT/step_breakdown_medium.txt
head -24 $OUT/step_breakdown_medium.txt
python3 $R/tools/kernel_avg.py $OUT/prof_med/prof/*/*_kernel_trace.csv 2 > $OUT/kernel_avg_medium.txt 2>&1 || true
rm -rf $OUT/prof_med/prof
