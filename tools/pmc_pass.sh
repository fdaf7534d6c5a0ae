#!/bin/bash
# One rocprofv3 counter pass: pmc_pass.sh <out-name> <script.py> <counters...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; script=$2; shift 2
O=$R/gpurun_out/$name
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O -- python3 $R/$script > $O.log 2>&1
