#!/bin/bash
# round 4, GPU session 11: the light sampling kernel — parity, and config 3 pipelined / blocking with it on and off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s11
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf --durations=5 -p no:cacheprovider --timeout 400 -k "light_sampling or onepass or randomised_soak" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
for lite in 0 1 0 1; do
  timeout 300 python bench.py --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --secondary b1m_q256,b1m_q1024,clustered_k10 --tune batch_sample_lite=$lite --detail-out "$OUT/bench_lite${lite}_$RANDOM.detail.json" >> "$OUT/bench_lite$lite.jsonl" 2> /dev/null
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lite" -o p -- python "$R/bench.py" --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --secondary b1m_q256 --tune batch_sample_lite=1 --detail-out "$OUT/prof_lite_detail.json" > "$OUT/prof_lite.out" 2> /dev/null)
find "$OUT/prof_lite" -name "*kernel_stats.csv" -exec cp {} "$OUT/lite_kernel_stats.csv" \; 2>/dev/null; rm -rf "$OUT/prof_lite"
