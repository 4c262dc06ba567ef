#!/bin/bash
# round 6, session 7: the whole GPU suite + bench on the build with the incremental mirror and the fused-pick general selection
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s7}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rf --durations=8 -p no:cacheprovider --timeout 500 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
grep -n "ingest while\|passed\|failed\|^FAILED\|^E  " "$OUT/pytest_gpu.log" | head -40
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench rc $? bytes $(wc -c < "$OUT/bench_n1.json")"; cat "$OUT/bench_n1.json"
