#!/bin/bash
# round 5, GPU session 18: longer fuzz of the batched path on the final build (two seeds, sharded handles included)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s18
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 400 python tools/fuzz_batch.py --seconds 240 --sharded 0.3 --seed 99 > "$OUT/fuzz_seed99.txt" 2>&1; echo "rc $?"; grep -a '"trials"' "$OUT/fuzz_seed99.txt" | tail -1
timeout 300 python tools/fuzz_batch.py --seconds 150 --sharded 0.0 --seed 1234 > "$OUT/fuzz_seed1234.txt" 2>&1; echo "rc $?"; grep -a '"trials"' "$OUT/fuzz_seed1234.txt" | tail -1
timeout 200 python tools/long_run_drift.py > "$OUT/long_run_drift.txt" 2>&1; tail -3 "$OUT/long_run_drift.txt"
