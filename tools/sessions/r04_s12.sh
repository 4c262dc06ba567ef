#!/bin/bash
# round 4, GPU session 12: the k-way merge of the partial lists' heads in the fused final merge (k <= 32) against the wave-list merge
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s12
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -rf -p no:cacheprovider --timeout 300 -k "fused_final_merge or query_in_kernel_arguments or completion_word or generic_dims or edge or ragged" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && timeout 300 /tmp/latency_c 10000 384 20000 > "$OUT/latency_c.jsonl" 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o p -- /tmp/latency_c 10000 384 3000 > /dev/null 2> "$OUT/prof.err")
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for path in glob.glob(os.path.join(out, "prof", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
        for r in rows[:12]:
            f.write(f"{r['Name'][:90]:90s} calls {r['Calls']:>7s} avg_ns {float(r['AverageNs']):10.0f} min {r['MinNs']} max {r['MaxNs']}\n")
PY
# per-mode kernel time: the trace in launch order, 50 warm-up + reps per mode, two corpora
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for path in glob.glob(os.path.join(out, "prof", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "scan_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = 3050
    with open(os.path.join(out, "kernel_time_per_mode.txt"), "w") as f:
        for i in range(0, len(rows), per):
            seg = rows[i + 50:i + per]
            if not seg: continue
            d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
            f.write(f"corpus {i // (per * 6)} mode {(i // per) % 6} {seg[0]['Kernel_Name'][:60]} n {len(d)} median_ns {d[len(d)//2]} mean_ns {sum(d)/len(d):.0f}\n")
PY
rm -rf "$OUT/prof"
cat "$OUT/latency_c.jsonl" | cut -c1-400; cat "$OUT/kernel_time_per_mode.txt"; tail -5 "$OUT/pytest_sel.log"
