#!/bin/bash
# round 5, GPU session 16: the default bench command with the CPU baseline over ALL 10M rows (no scaling)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s16
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
free -g | head -2
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; echo "bench rc $? seconds $(( $(date +%s) - t0 )) bytes $(wc -c < "$OUT/bench_n1.json")"
python -c "
import json
d=json.loads(open('$OUT/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['cpu_baseline'])
"
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$OUT/bench_driver_style_detail.json" > "$OUT/bench_driver_style.json" 2> /dev/null; echo "driver-style rc $? seconds $(( $(date +%s) - t0 ))"
