#!/bin/bash
# round 4, GPU session 16: device-side retry re-scores only the survivors the first finish's exact k-th cannot exclude (A/B against all survivors)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s16
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf -x -s -p no:cacheprovider --timeout 600 -k "device_side_full_retry or certificate_bound or dense or clustered or resend" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
for i in 1 2; do
  for dbg in 0 65536; do
    timeout 600 python bench.py --rows 1000000 --steps 50 --warmup 5 --no-cpu-baseline --traffic off --secondary b1m_q256,clustered_k10,clustered_k100,dups17 --tune batch_debug=$dbg --detail-out "$OUT/bench_${dbg}_$i.detail.json" 2> /dev/null | tail -1 >> "$OUT/bench_$dbg.jsonl"
  done
done
python - "$OUT" > "$OUT/retry_ab.txt" <<'PY'
import json, sys, os, glob
for dbg in (0, 65536):
    for path in sorted(glob.glob(os.path.join(sys.argv[1], f"bench_{dbg}_*.detail.json"))):
        d = json.load(open(path))
        parts = []
        for s in d.get("secondary", []):
            parts.append(f"{s['name']} {s['ms_per_step']*1000:.1f}us blk {s.get('ms_per_step_blocking_call', 0)*1000:.1f}us x{s.get('ms_per_step_vs_iid_config3', 0):.2f} dev-retries/step {s.get('full_retries_on_device', s.get('inline_retries_per_step', '?'))} ck {s['last_result_checksum'][:8]}")
        print("retry_all" if dbg else "pruned   ", " | ".join(parts))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o p -- python "$R/bench.py" --rows 1000000 --steps 20 --warmup 2 --no-cpu-baseline --traffic off --secondary clustered_k100,dups17 > /dev/null 2> "$OUT/prof.err")
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for path in glob.glob(os.path.join(out, "prof", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
        for r in rows[:14]:
            f.write(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):10.0f} min {r['MinNs']} max {r['MaxNs']}\n")
PY
rm -rf "$OUT/prof" "$OUT/prof.err"
cat "$OUT/retry_ab.txt"; grep -a "retry\|wax::batch" "$OUT/kernel_stats.txt" | cut -c1-160; grep -a "device retry\|passed\|failed\|rc" "$OUT/pytest_sel.log" | tail -20
