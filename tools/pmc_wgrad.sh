#!/bin/bash
# PMC passes over the weight-gradient kernel (one counter group per run).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wgrad
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/wgrad_pmc.py > $O/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tcc -- python3 $R/tools/wgrad_pmc.py > $O/tcc.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ea -- python3 $R/tools/wgrad_pmc.py > $O/ea.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/wgrad_pmc.py > $O/kt.log 2>&1 || exit 1
find $O -name "*.csv" | head -20
