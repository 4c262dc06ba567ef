#!/bin/bash
# round 5, GPU session 8: the whole GPU suite on the single-GEMM build + the default bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s8
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=15 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"; tail -30 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --detail-out "$OUT/bench_detail.json" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; cat "$OUT/bench.json"
