set -e
R=$GRAFT_REPO_ROOT
export PDA_WGRAD_SLICES=$1
bash $R/tools/pmc_pass.sh mp_s$1 tools/wgrad_pmc.py TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
python3 $R/tools/pmc_summary.py $R/gpurun_out/mp_s$1 wgrad_kernel4 > $R/gpurun_out/mp_s$1.txt 2>&1
