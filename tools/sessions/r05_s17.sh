#!/bin/bash
# round 5, GPU session 17: bench contract test (small-store rule in the torchrun shape) + the scaling-matrix rehearsal on one GPU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s17
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf -p no:cacheprovider --timeout 600 -k "bench_contract" > "$OUT/pytest_bench.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_bench.log"; tail -5 "$OUT/pytest_bench.log"
WAX_SCALE_SAME_DEVICE=1 WAX_SCALE_ROWS="10000 1000000" timeout 600 bash tools/scale_matrix.sh "$OUT/scale_rehearsal.jsonl" 2 > "$OUT/scale_rehearsal.txt" 2>&1
rm -f "$OUT"/scale_rehearsal.jsonl.detail_*; cat "$OUT/scale_rehearsal.txt"
