set -e
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_pass.sh mp_tcc2 tools/wgrad_pmc.py TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
bash $R/tools/pmc_pass.sh mp_tcc3 tools/wgrad_pmc.py TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_sum
for d in mp_tcc2 mp_tcc3; do python3 $R/tools/pmc_summary.py $R/gpurun_out/$d wgrad_kernel4 > $R/gpurun_out/$d.txt 2>&1; done
echo ok
