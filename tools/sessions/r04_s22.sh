#!/bin/bash
# round 4, GPU session 22: small stores with one chunk per wave (twice the workgroups, twice the lists) now that the k-way merge does not care
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s22
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for rep in 1 2; do for n in 2500 5000 10000; do
  timeout 300 /tmp/latency_c $n 384 20000 2>&1 | grep '"mode": 1,' | sed 's/^/halved /' >> "$OUT/latency_halve.txt"
  WAX_SCAN_NO_HALVE=1 timeout 300 /tmp/latency_c $n 384 20000 2>&1 | grep '"mode": 1,' | sed 's/^/one-chunk /' >> "$OUT/latency_halve.txt"
done; done
python - "$OUT/latency_halve.txt" <<'PY' | tee "$OUT/latency_halve_table.txt"
import json, sys
for l in open(sys.argv[1]):
    tag, js = l.split(" ", 1)
    d = json.loads(js)
    print(tag, d["rows"], "top_k", d["top_k"], "grid", d["scan_grid"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"])
PY
