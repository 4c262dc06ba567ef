#!/bin/bash
# round 5, last check of the final build: smoke, the whole GPU suite, the driver's bench command (record)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_last}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 600 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench rc $? bytes $(wc -c < "$OUT/bench_n1.json")" > "$OUT/bench_n1.rc"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$OUT/bench_driver_style_detail.json" > "$OUT/bench_driver_style.json" 2> /dev/null
echo "driver-style rc $? bytes $(wc -c < "$OUT/bench_driver_style.json")" >> "$OUT/bench_n1.rc"
tail -2 "$OUT/smoke.log"; tail -4 "$OUT/pytest_gpu.log"; cat "$OUT/bench_n1.rc" "$OUT/bench_n1.json"
