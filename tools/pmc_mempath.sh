set -e
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_pass.sh mp_tcp1 tools/wgrad_pmc.py TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
bash $R/tools/pmc_pass.sh mp_tcp2 tools/wgrad_pmc.py TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TD_TC_STALL_sum TD_TD_BUSY_sum
bash $R/tools/pmc_pass.sh mp_tcc1 tools/wgrad_pmc.py TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_CYCLE_sum
bash $R/tools/pmc_pass.sh mp_tcp3 tools/wgrad_pmc.py TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
for d in mp_tcp1 mp_tcp2 mp_tcc1 mp_tcp3; do python3 $R/tools/pmc_summary.py $R/gpurun_out/$d wgrad > $R/gpurun_out/$d.txt 2>&1; done
echo ok
