#!/bin/bash
# Collects the numbers quoted in profiles/README.md into gpurun_out/final (run on the GPU box).
# Every GPU step runs under its own time limit and the script stops at the first failure.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
B="python bench.py --steps 10 --warmup 3"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
M32="--batch-per-gpu 32 --micro-batch 32"
timeout -k 10 250 $B > $O/bench_large_single_mb64.json 2> $O/bench.err || exit 1
timeout -k 10 150 $B $M32 > $O/bench_large_single_mb32.json 2>> $O/bench.err || exit 1
timeout -k 10 250 $B --activation-checkpoint selective > $O/bench_large_single_mb64_selective.json 2>> $O/bench.err || exit 1
timeout -k 10 250 $B --activation-checkpoint full > $O/bench_large_single_mb64_full.json 2>> $O/bench.err || exit 1
timeout -k 10 150 $B --batch-per-gpu 32 --micro-batch 8 > $O/bench_large_single_mb8_dropout0.json 2>> $O/bench.err || exit 1
timeout -k 10 150 $B --batch-per-gpu 32 --micro-batch 8 --dropout 0.1 > $O/bench_large_single_mb8_dropout0.1.json 2>> $O/bench.err || exit 1
timeout -k 10 200 $B --model gpt2-medium > $O/bench_medium_single_mb64.json 2>> $O/bench.err || exit 1
timeout -k 10 150 $B --model gpt2-medium $M32 > $O/bench_medium_single_mb32.json 2>> $O/bench.err || exit 1
timeout -k 10 150 $B --model gpt2 > $O/bench_small_single_mb64.json 2>> $O/bench.err || exit 1
timeout -k 10 300 $B --model gpt2-xl > $O/bench_xl_single_mb64.json 2>> $O/bench.err || exit 1
timeout -k 10 400 $TR --master-port 29751 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode both > $O/bench_large_world1_forced_ddp_fsdp_mb64.json 2>> $O/bench.err || exit 1
timeout -k 10 300 $TR --master-port 29753 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode both $M32 > $O/bench_large_world1_forced_ddp_fsdp_mb32.json 2>> $O/bench.err || exit 1
timeout -k 10 300 $TR --master-port 29752 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode both --model gpt2-medium > $O/bench_medium_world1_forced_ddp_fsdp_mb64.json 2>> $O/bench.err || exit 1
timeout -k 10 100 python tools/wgrad_probe.py > $O/wgrad_probe.log 2>&1 || exit 1
PDA_WGRAD_WAVES=8 timeout -k 10 100 python tools/wgrad_probe.py > $O/wgrad_probe_eight_wave.log 2>&1 || exit 1
timeout -k 10 60 python tools/attn_time.py > $O/attention_time.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 4 --warmup 3 > $O/prof.log 2>&1 || exit 1
python3 $R/tools/step_breakdown.py $O/prof/*/*_kernel_trace.csv 3 > $O/step_breakdown.txt || exit 1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_wgrad -- python3 $R/tools/wgrad_pmc.py > $O/pmc_wgrad.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O/pmc_wgrad > $O/pmc_wgrad_summary.txt 2>&1
rm -rf $O/pmc_wgrad
# effective shader clock per kernel (power throttling shows here, not in cycle counters)
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk -- python3 $R/bench.py --steps 3 --warmup 2 > $O/clk.log 2>&1 || exit 1
python3 $R/tools/kernel_clock.py $O/clk/*/*counter_collection.csv $O/clk/*/*kernel_trace.csv > $O/kernel_clocks.txt 2>&1
rm -rf $O/clk
[ -x $R/tools/micro/bin/stream_probe ] && timeout -k 10 120 $R/tools/micro/bin/stream_probe > $O/stream_probe.log 2>&1
# operand-feed, MFMA-issue and weight-gradient phase probes (built on the CPU box beforehand, see each file's header)
for probe in feed_probe mfma_probe coexec_probe wgrad_stamp; do
  [ -x $R/tools/micro/bin/$probe ] && { timeout -k 10 120 $R/tools/micro/bin/$probe > $O/$probe.log 2>&1 || exit 1; }
done
cd $R && timeout -k 10 400 $TR --master-port 29754 bench.py --gpus 1 --steps 10 --warmup 3 --wrap-single --mode fsdp --fsdp-resident-gb auto > $O/bench_large_world1_forced_fsdp_resident_auto.json 2>> $O/bench.err || exit 1
echo collected
