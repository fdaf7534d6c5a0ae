#!/bin/bash
# A/B of two builds on one box: kernel traces of bench.py from the working tree and from the baseline
# tree (.base_tree, a checkout of an earlier commit with its own built extension), then plain timed
# runs alternating between the two. usage: tools/ab_profile.sh [extra bench.py args]
# The baseline tree is made on the build machine before the GPU call:
#   git worktree add --detach .base_tree <commit> && (cd .base_tree && python -m pytorch_distributed_amd._build)
# (.base_tree is git-ignored; it travels to the GPU box with the rest of the working directory).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in new base; do
  T=$R; [ $v = base ] && T=$R/.base_tree
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$v -- python3 $T/bench.py --steps 4 --warmup 3 "$@" > $O/prof_$v.log 2>&1 || exit 1
  python3 $R/tools/kernel_avg.py $O/prof_$v/*/*_kernel_trace.csv 3 > $O/kernels_$v.txt
  python3 $R/tools/step_breakdown.py $O/prof_$v/*/*_kernel_trace.csv 3 > $O/breakdown_$v.txt
  rm -rf $O/prof_$v
done
for i in 1 2; do
  for v in new base; do
    T=$R; [ $v = base ] && T=$R/.base_tree
    timeout -k 10 200 python3 $T/bench.py --steps 10 --warmup 3 "$@" > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$O/bench_${v}_$i.json')); print('$v $i', round(d['ms_per_step'],2), 'ms', round(d['mfu']*100,2), '% MFU')"
  done
done
