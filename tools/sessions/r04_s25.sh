#!/bin/bash
# round 4, GPU session 25 (last): fan-out cost of the sharded handle on the final build, the full GPU suite, the default bench command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s25
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 400 python tools/sharded_handle_bench.py --parts A,B > "$OUT/sharded_fanout.jsonl" 2> "$OUT/sharded_fanout.err"
timeout 1200 python -m pytest tests -m gpu -q -rf --durations=4 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
t0=$(date +%s)
timeout 900 python bench.py --detail-out "$OUT/bench_default.detail.json" > "$OUT/bench_default.out" 2> /dev/null
echo "bench rc $? wall $(( $(date +%s) - t0 )) s bytes $(tail -1 "$OUT/bench_default.out" | wc -c)" > "$OUT/bench_default.rc"
grep '^{' "$OUT/sharded_fanout.jsonl" | cut -c1-330; tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench_default.rc"; tail -1 "$OUT/bench_default.out" | cut -c1-900
rm -f "$OUT/sharded_fanout.err"
