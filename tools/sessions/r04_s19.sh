#!/bin/bash
# round 4, GPU session 19: do mid-size stores (30 - 500 MB: inside the 256 MB Infinity Cache or not) gain from ordinary instead of
# non-temporal row loads? blocking C latency, mode 1 (default) vs mode 6 (scan_plain_mb = everything)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s19
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for n in 20000 40000 70000 100000 150000 200000 300000 600000; do
  timeout 300 /tmp/latency_c $n 384 3000 2>&1 | grep '"unit gaussian"' | grep '"mode": [16]' >> "$OUT/latency_plain_loads.jsonl"
done
python - "$OUT/latency_plain_loads.jsonl" <<'PY' | tee "$OUT/latency_plain_loads.txt"
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["rows"], f"{d['rows']*d['dims']*4/1e6:.0f} MB", "mode", d["mode"], "grid", d["scan_grid"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"], "same", d["same_ids_as_query_args_0"])
PY
