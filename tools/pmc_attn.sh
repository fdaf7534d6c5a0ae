#!/bin/bash
# PMC passes over the attention kernels (one counter group per run).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq -- python3 $R/tools/attn_pmc.py > $O/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/sq2 -- python3 $R/tools/attn_pmc.py > $O/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/attn_pmc.py > $O/kt.log 2>&1 || exit 1
