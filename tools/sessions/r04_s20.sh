#!/bin/bash
# round 4, GPU session 20: every BASELINE secondary's roofline.traffic measured inside the bench run (child counter passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s20
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py --detail-out "$OUT/bench_default.detail.json" > "$OUT/bench_default.out" 2> "$OUT/bench_default.err"
echo "bench rc $? wall $(( $(date +%s) - t0 )) s bytes $(tail -1 "$OUT/bench_default.out" | wc -c)" > "$OUT/bench_default.rc"
grep -a "live traffic" "$OUT/bench_default.err" | cut -c1-400 > "$OUT/bench_default.notes"
python - "$OUT" <<'PY' | tee "$OUT/traffic_table.txt"
import json, sys, os
d = json.load(open(os.path.join(sys.argv[1], "bench_default.detail.json")))
def row(name, r):
    t, a = r.get("traffic"), r.get("algorithmic_bytes_per_launch")
    print(f"{name:14s} traffic {t} algorithmic {a} ratio {(t / a) if t else None}  [{str(r.get('traffic_source'))[:60]}]")
row("headline", d["roofline"])
for s in d.get("secondary", []):
    row(s["name"], s["roofline"])
PY
cat "$OUT/bench_default.rc" "$OUT/bench_default.notes"
timeout 900 python -m pytest tests -m gpu -q -rf -x -p no:cacheprovider --timeout 900 -k "bench_contract or device_side_full_retry" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"; tail -5 "$OUT/pytest_sel.log"
rm -f "$OUT/bench_default.err"
