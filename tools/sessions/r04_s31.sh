#!/bin/bash
# round 4, GPU session 31: the k-way merge at top_k 33 .. 64 (library built with SCAN_KWAY_MAX_K = 64): mode 1 (k-way) against mode 5 (merge_kway 0:
# wave lists on small grids, two launches on large ones), blocking C latency
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s31
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for k in 40 64; do for n in 10000 100000; do
  timeout 120 /tmp/latency_c $n 384 4000 $k 2>&1 | grep '"unit gaussian"' | grep '"mode": [15],' >> "$OUT/latency_k.jsonl"
done; done
python - "$OUT/latency_k.jsonl" <<'PY' | tee "$OUT/latency_k.txt"
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("rows", d["rows"], "top_k", d["top_k"], "mode", d["mode"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"], "same", d["same_ids_as_query_args_0"])
PY
