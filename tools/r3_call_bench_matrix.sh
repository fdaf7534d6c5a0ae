#!/bin/bash
# Option matrix of bench.py and the assignment-0 tools on one GPU (small models, a few steps each).
set -o pipefail
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3/bench_matrix; mkdir -p $OUT; cd $R
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
C="--model gpt2 --steps 3 --warmup 1 --no-reference-recipe --batch-per-gpu 16 --micro-batch 16"
n=0; port=29900
run() {
  local name=$1; shift
  n=$((n+1)); port=$((port+1))
  timeout -k 10 240 "$@" > $OUT/$name.json 2> $OUT/$name.err; local rc=$?
  case $rc in 124|134|137|139) echo "fatal exit $rc in $name"; tail -5 $OUT/$name.err; exit $rc;; esac
  if [ $rc -ne 0 ]; then echo "$name: rc=$rc"; tail -15 $OUT/$name.err; exit 1; fi
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
modes = d.get("modes", {})
bad = {k: v["error"] for k, v in modes.items() if "error" in v}
print(f"{sys.argv[2]}: {d['ms_per_step']:.1f} ms, mfu {100 * d['mfu']:.1f} %, {d['config']['parallelism']}" + (f" ERRORS {bad}" if bad else ""))
sys.exit(1 if bad else 0)
PY
}
run single python bench.py $C
run single_accum4 python bench.py $C --micro-batch 4
run single_full python bench.py $C --activation-checkpoint full
run single_dropout python bench.py $C --dropout 0.1
run single_graph python bench.py $C --hip-graph
run single_fp32 python bench.py $C --dtype fp32
run single_llama python bench.py --model llama-tiny --seq-len 128 --steps 3 --warmup 1 --no-reference-recipe --batch-per-gpu 16 --micro-batch 16
run wrapped_both $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode both
run wrapped_accum $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode both --micro-batch 4
run wrapped_decompose $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode ddp --ddp-decompose on
run wrapped_grad_op $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode fsdp --strategy SHARD_GRAD_OP
run wrapped_no_shard $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode fsdp --strategy NO_SHARD
run wrapped_units4_fp32 $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode fsdp --fsdp-unit-blocks 4 --reduce-fp32
run wrapped_c10d $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode both --comm c10d
run wrapped_dual $TR --master-port $port bench.py --gpus 1 $C --wrap-single --mode both --comm-dual on
run self_launch python bench.py --gpus 1 $C --wrap-single
echo "== assignment-0 tools"
timeout -k 10 240 python memory_analysis.py --model gpt2 > $OUT/memory_analysis.log 2>&1; rc=$?; echo "memory_analysis rc=$rc"; [ $rc -ne 0 ] && { tail -15 $OUT/memory_analysis.log; exit 1; }
timeout -k 10 240 python throughput.py --model gpt2 > $OUT/throughput.log 2>&1; rc=$?; echo "throughput rc=$rc"; [ $rc -ne 0 ] && { tail -15 $OUT/throughput.log; exit 1; }
tail -4 $OUT/throughput.log
echo "bench matrix ok: $n runs"
