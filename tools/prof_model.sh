#!/bin/bash
# Kernel trace + per-step breakdown of bench.py for one model: tools/prof_model.sh <model> [extra bench.py args]
set -o pipefail
R=$GRAFT_REPO_ROOT
MODEL=${1:-gpt2}
shift
O=$R/gpurun_out/prof_$MODEL
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --model $MODEL --steps 4 --warmup 3 "$@" > $O/prof.log 2>&1 || exit 1
python3 $R/tools/step_breakdown.py $O/prof/*/*_kernel_trace.csv 3 > $O/step_breakdown.txt
head -14 $O/step_breakdown.txt; grep -A5 "^queue" $O/step_breakdown.txt
