#!/bin/bash
# round 4, GPU session 13: blocking single-query latency from C against the store size (where does the fused-merge limit bite?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s13
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for n in 5000 10000 20000 20500 40000 100000 300000 1000000; do
  timeout 300 /tmp/latency_c $n 384 4000 2>&1 | grep '"unit gaussian"' | grep '"mode": [015]' >> "$OUT/latency_vs_rows.jsonl"
done
python - "$OUT/latency_vs_rows.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["rows"], "mode", d["mode"], "grid", d["scan_grid"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"], "floor_us_at_8TBps", round(d["rows"] * d["dims"] * 4 / 8e6, 2))
PY
