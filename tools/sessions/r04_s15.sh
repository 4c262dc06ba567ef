#!/bin/bash
# round 4, GPU session 15: in-kernel k-way merge bounded to 2 GiB stores + roofline.traffic measured inside bench.py (child counter pass)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s15
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf -x -p no:cacheprovider --timeout 900 -k "fused_final_merge or query_in_kernel_arguments or completion_word or full_size_parity_with_oracle or bench_contract" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
t0=$(date +%s)
timeout 900 python bench.py --detail-out "$OUT/bench_default.detail.json" > "$OUT/bench_default.out" 2> "$OUT/bench_default.err"
echo "bench rc $? wall $(( $(date +%s) - t0 )) s bytes $(tail -1 "$OUT/bench_default.out" | wc -c)" > "$OUT/bench_default.rc"
grep -a "live traffic\|\[bench\] rank" "$OUT/bench_default.err" | cut -c1-300 > "$OUT/bench_default.notes"
python - "$OUT" <<'PY'
import json, sys, os
d = json.load(open(os.path.join(sys.argv[1], "bench_default.detail.json")))
r = d["roofline"]
print("value", d["value"], "frac", r["frac"], "traffic", r["traffic"], "alg", r["algorithmic_bytes_per_launch"], "ratio", (r["traffic"] or 0) / r["algorithmic_bytes_per_launch"])
print(r["traffic_source"])
for s in d.get("secondary", []):
    print(s["name"], round(s["ms_per_step"] * 1000, 2), "us", "blocking", s.get("ms_per_step_blocking_call"), "lpq", s["roofline"].get("launches_per_query"))
PY
cat "$OUT/bench_default.rc" "$OUT/bench_default.notes"; tail -1 "$OUT/bench_default.out" | cut -c1-700; tail -8 "$OUT/pytest_sel.log"
rm -f "$OUT/bench_default.err"
