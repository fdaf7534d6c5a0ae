cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt_attn
mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/attn_pmc.py > $O.log 2>&1 || exit 1
