#!/bin/bash
# round 4, GPU session 23: mid-size stores — fewer, fatter workgroups (grid caps 160 / 256) against the default grid, blocking C latency
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s23
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for n in 5000 20000 40000 100000 300000; do for g in 160 256; do
  WAX_LAT_GRID=$g timeout 300 /tmp/latency_c $n 384 6000 2>&1 | grep '"unit gaussian"' | grep '"mode": [13],' | sed "s/^/cap$g /" >> "$OUT/latency_grid.txt"
done; done
python - "$OUT/latency_grid.txt" <<'PY' | tee "$OUT/latency_grid_table.txt"
import json, sys
for l in open(sys.argv[1]):
    tag, js = l.split(" ", 1)
    d = json.loads(js)
    print(tag if d["mode"] == 3 else "default", d["rows"], "grid", d["scan_grid"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"])
PY
