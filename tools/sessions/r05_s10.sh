#!/bin/bash
# round 5, GPU session 10: whole GPU suite again (planner unit fix)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s10
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"; tail -25 "$OUT/pytest_gpu.log"
