#!/bin/bash
# round 5, session 29: kernel timeline of config 3, product mode, three batches in flight (what is left between two filtering GEMMs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s29}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for d in 3 2; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_pl$d" -o t -- python "$R/bench.py" --gpus 1 --rows 1000000 --steps 60 --warmup 5 --no-cpu-baseline --traffic off --batch-depth $d --secondary b1m_q256 --detail-out "$OUT/d.json" > "$OUT/pl$d.out" 2> "$OUT/pl$d.err")
f=$(find "$OUT/prof_pl$d" -name "*kernel_trace.csv" | head -1)
python - "$f" > "$OUT/pipelined_timeline_depth$d.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "batch_gemm_rq_kernel" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
# dispatch order of the bench: 1 mirror + warm-up 5 + blocking 15 + timed 60 + calibration: take the middle of the timed 60
mid = idx[21 + 30]
sel = rows[mid - 20: mid + 25]
t0 = int(sel[0]["Start_Timestamp"])
print("start_us,end_us,dur_us,queue,kernel")
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void wax::", "").replace("wax::", "").split("(")[0][:52]
    print(f"{s / 1e3:9.1f},{e / 1e3:9.1f},{(e - s) / 1e3:7.1f},{r.get('Queue_Id', '?')},{name}")
PY
rm -rf "$OUT/prof_pl$d" "$OUT/d.json"
done
cat "$OUT/pipelined_timeline_depth3.csv"; echo; cat "$OUT/pipelined_timeline_depth2.csv" | head -30
