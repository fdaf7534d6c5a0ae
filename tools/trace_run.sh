#!/bin/bash
# Kernel-level trace of a train script with the framework's own tracer (no Kineto):
#   tools/trace_run.sh <config name> <script and its arguments...>
# e.g. tools/trace_run.sh baseline train_baseline.py --model gpt2 --synthetic-data --max-steps 12
# Runs the script under rocprofv3 (kernel dispatches + roctx ranges on one clock), converts the
# result into outputs/traces/<config>/rank{r}_trace.json and prints the analysis.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=$1; shift
O=$R/gpurun_out/trace_$CFG
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R:$PYTHONPATH
SCRIPT=$R/$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --marker-trace --output-format csv -d "$O/prof" -- python3 "$SCRIPT" --profiler-backend range --trace-dir "$O/ranges" "$@" > "$O/run.log" 2>&1 || { tail -20 "$O/run.log"; exit 1; }
cd "$R" && python3 analyze_traces.py --traces "$O/traces" --from-rocprof "$O/prof" --config "$CFG" --active-only | tee "$O/analysis.txt"
