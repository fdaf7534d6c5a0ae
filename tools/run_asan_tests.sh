#!/bin/bash
# Builds the host-side C++ with AddressSanitizer + UBSan (build/asan/_C.so) and runs the CPU test
# suite against it. CPU only: GPU code is never sanitized.
set -e
cd "$(dirname "$0")/.."
python -m pytorch_distributed_amd._build --sanitize | tail -1
# An overlay package directory whose _C.so is the sanitized build; everything else is the tree.
OVERLAY=build/asan-overlay/pytorch_distributed_amd
rm -rf build/asan-overlay && mkdir -p "$OVERLAY"
for f in pytorch_distributed_amd/*; do
  [ "$(basename "$f")" = "_C.so" ] || ln -s "$(pwd)/$f" "$OVERLAY/$(basename "$f")"
done
ln -s "$(pwd)/build/asan/_C.so" "$OVERLAY/_C.so"
ln -sfn "$(pwd)/tests" build/asan-overlay/tests
# tests that load the entry scripts by path (bench.py, counter_profile.py, ...) look next to tests/
for f in *.py tools; do ln -sfn "$(pwd)/$f" "build/asan-overlay/$f"; done
# the sanitizer runtimes go first; whatever the environment already preloads is kept
export LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)${LD_PRELOAD:+:$LD_PRELOAD}"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:detect_odr_violation=0
export PYTHONPATH="$(pwd)/build/asan-overlay"
cd build/asan-overlay
# the script-level tests spawn torchrun from the repository root and are left to the normal build
if [ $# -eq 0 ]; then set -- tests/test_cpu_units.py tests/test_ddp_cpu.py tests/test_fsdp_cpu.py; fi
exec python -m pytest -q -m "not gpu" -p no:cacheprovider "$@"
