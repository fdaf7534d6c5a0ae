#!/bin/bash
# Kernel trace of the wrapped (world size 1, RCCL collectives forced) benchmark: tools/prof_wrapped.sh {fsdp|ddp}
set -o pipefail
R=$GRAFT_REPO_ROOT
MODE=${1:-fsdp}
O=$R/gpurun_out/wrapped_$MODE
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 4 --warmup 3 --wrap-single --mode $MODE > $O/prof.log 2>&1 || exit 1
python3 $R/tools/step_breakdown.py $O/prof/*/*_kernel_trace.csv 3 > $O/step_breakdown.txt
head -30 $O/step_breakdown.txt
