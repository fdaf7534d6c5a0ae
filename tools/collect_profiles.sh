#!/bin/bash
# Collects the numbers quoted in profiles/README.md into gpurun_out/final (run on the GPU box).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 150 python bench.py --steps 8 --warmup 3 > $O/bench_single_mb32.json 2> $O/bench_single_mb32.err || exit 1
timeout -k 10 150 python bench.py --steps 8 --warmup 3 --micro-batch 16 > $O/bench_single_mb16.json 2>> $O/bench_single_mb32.err || exit 1
timeout -k 10 150 python bench.py --steps 8 --warmup 3 --wrap-single --mode fsdp > $O/bench_fsdp_world1.json 2>> $O/bench_single_mb32.err || exit 1
timeout -k 10 150 python bench.py --steps 8 --warmup 3 --wrap-single --mode ddp > $O/bench_ddp_world1.json 2>> $O/bench_single_mb32.err || exit 1
timeout -k 10 100 python tools/wgrad_probe.py > $O/wgrad_probe.log 2>&1 || exit 1
timeout -k 10 100 python tools/lt_probe.py > $O/gemm_planner_probe.log 2>&1 || exit 1
timeout -k 10 60 python tools/attn_time.py > $O/attention_time.log 2>&1 || exit 1
timeout -k 10 60 ./tools/micro/mfma_peak > $O/mfma_peak.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 4 --warmup 3 > $O/prof.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_attn -- python3 $R/tools/attn_pmc.py > $O/pmc_attn.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_wgrad -- python3 $R/tools/wgrad_pmc.py > $O/pmc_wgrad.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_wgrad_l2 -- python3 $R/tools/wgrad_pmc.py > $O/pmc_wgrad_l2.log 2>&1 || exit 1
echo collected
