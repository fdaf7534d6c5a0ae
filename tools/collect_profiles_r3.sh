#!/bin/bash
# Collects the numbers quoted in profiles/README.md (tables of `r3_final/`) into gpurun_out/r3_final (run on the GPU box).
# Every GPU step runs under its own time limit and the script stops at the first failure.
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_final
mkdir -p $O
cd $R
B="python bench.py --steps 10 --warmup 3"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
step() { echo "== $1" ; }
step "bench: large single (default + reference recipe)"
timeout -k 10 300 $B > $O/bench_large_single.json 2> $O/bench.err || exit 1
step "bench: large wrapped"
timeout -k 10 500 $TR --master-port 29751 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode both > $O/bench_large_world1_forced_ddp_fsdp.json 2>> $O/bench.err || exit 1
step "bench: medium single / wrapped"
timeout -k 10 300 $B --model gpt2-medium > $O/bench_medium_single.json 2>> $O/bench.err || exit 1
timeout -k 10 400 $TR --master-port 29752 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode both --model gpt2-medium > $O/bench_medium_world1_forced_ddp_fsdp.json 2>> $O/bench.err || exit 1
step "bench: small, xl"
timeout -k 10 200 $B --model gpt2 > $O/bench_small_single.json 2>> $O/bench.err || exit 1
timeout -k 10 400 $B --model gpt2-xl > $O/bench_xl_single.json 2>> $O/bench.err || exit 1
step "train scripts with their own defaults (gpt2-medium, global 32 = 4 x 8, selective)"
timeout -k 10 300 python train_baseline.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile > $O/train_baseline_defaults.log 2>&1 || exit 1
timeout -k 10 300 $TR --master-port 29761 train_ddp.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile --force-collectives > $O/train_ddp_defaults.log 2>&1 || exit 1
timeout -k 10 300 $TR --master-port 29771 train_fsdp.py --synthetic-data --model gpt2-medium --max-steps 12 --log-every 4 --no-profile --force-collectives > $O/train_fsdp_defaults.log 2>&1 || exit 1
grep -h "step=" $O/train_*_defaults.log | tail -9
step "kernel breakdowns"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-reference-recipe --steps 4 --warmup 3 > $O/prof.log 2>&1 || exit 1
python3 $R/tools/step_breakdown.py $O/prof/*/*_kernel_trace.csv 3 > $O/step_breakdown.txt || exit 1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --batch-per-gpu 32 --micro-batch 8 --activation-checkpoint selective --no-reference-recipe --steps 3 --warmup 2 > $O/prof_ref.log 2>&1 || exit 1
python3 $R/tools/step_breakdown.py $O/prof/*/*_kernel_trace.csv 2 4 > $O/step_breakdown_reference_recipe.txt || exit 1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats_reference_recipe.csv 2>/dev/null
rm -rf $O/prof
step "effective clocks"
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk -- python3 $R/bench.py --no-reference-recipe --steps 3 --warmup 2 > $O/clk.log 2>&1 || exit 1
python3 $R/tools/kernel_clock.py $O/clk/*/*counter_collection.csv $O/clk/*/*kernel_trace.csv > $O/kernel_clocks.txt 2>&1
rm -rf $O/clk
step "counters of one optimizer step"
cd $R
timeout -k 10 1000 python3 counter_profile.py --out $O/counters --limit 200 > $O/counters_step.txt 2> $O/counters_step.err || { tail -5 $O/counters_step.err; exit 1; }
rm -rf $O/counters
step "probes"
timeout -k 10 100 python tools/wgrad_probe.py 65536 > $O/wgrad_probe_64k.log 2>&1 || exit 1
timeout -k 10 100 python tools/wgrad_probe.py 32768 > $O/wgrad_probe_32k.log 2>&1 || exit 1
timeout -k 10 60 python tools/attn_time.py > $O/attention_time.log 2>&1 || exit 1
echo collected
