#!/bin/bash
# round 4, GPU session 14: the k-way merge on every default grid (k <= 32: one launch per query at all store sizes):
# parity tests, blocking latency against the store size, the headline and the single-query secondaries with fuse_merge 1 / 0
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s14
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf -x -p no:cacheprovider --timeout 300 -k "fused_final_merge or query_in_kernel_arguments or completion_word or generic_dims or edge or ragged or sharded or full_size_parity_with_oracle" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for n in 10000 20000 40000 100000 300000 1000000; do
  timeout 300 /tmp/latency_c $n 384 4000 2>&1 | grep '"unit gaussian"' | grep '"mode": [0145]' >> "$OUT/latency_vs_rows.jsonl"
done
python - "$OUT/latency_vs_rows.jsonl" > "$OUT/latency_vs_rows.txt" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["rows"], "mode", d["mode"], "grid", d["scan_grid"], "qarg", d["scans_with_query_in_kernel_args"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"], "same", d["same_ids_as_query_args_0"])
PY
for i in 1 2; do
  for f in 1 0; do
    timeout 600 python bench.py --no-cpu-baseline --secondary s10k,s1m,s1250k --tune fuse_merge=$f --detail-out "$OUT/bench_fuse${f}_$i.detail.json" 2> /dev/null | tail -1 >> "$OUT/bench_fuse$f.jsonl"
  done
done
python - "$OUT" > "$OUT/bench_ab.txt" <<'PY'
import json, sys, os
for f in (1, 0):
    for l in open(os.path.join(sys.argv[1], f"bench_fuse{f}.jsonl")):
        d = json.loads(l)
        print("fuse_merge", f, "value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"].get("kernel_avg_ms"),
              " ".join(f"{s['name']} {s['ms_per_step']*1000:.2f}us k {s['kernel_avg_ms']*1000:.2f}us" for s in d.get("secondary", [])))
PY
cat "$OUT/latency_vs_rows.txt" "$OUT/bench_ab.txt"; tail -6 "$OUT/pytest_sel.log"
