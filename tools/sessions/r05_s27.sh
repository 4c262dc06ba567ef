#!/bin/bash
# round 5, session 27: kernel timeline of config 3 with two batches in flight (what sits between two filtering GEMMs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s27}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_pl" -o t -- python "$R/bench.py" --gpus 1 --rows 1000000 --steps 20 --warmup 5 --no-cpu-baseline --traffic off --events bracket --secondary b1m_q256 --detail-out "$OUT/d.json" > "$OUT/pl.out" 2> "$OUT/pl.err")
f=$(find "$OUT/prof_pl" -name "*kernel_trace.csv" | head -1)
python - "$f" > "$OUT/pipelined_timeline.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the steady state of the pipelined batched region: find GEMM filtering launches, print 60 dispatches around the 100th
idx = [i for i, r in enumerate(rows) if "batch_gemm_rq_kernel" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
mid = idx[len(idx) // 2]
sel = rows[mid - 25: mid + 25]
t0 = int(sel[0]["Start_Timestamp"])
print("start_us,end_us,dur_us,queue,kernel")
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void wax::", "").replace("wax::", "").split("(")[0][:52]
    print(f"{s / 1e3:9.1f},{e / 1e3:9.1f},{(e - s) / 1e3:7.1f},{r.get('Queue_Id', '?')},{name}")
PY
rm -rf "$OUT/prof_pl" "$OUT/d.json"
cat "$OUT/pipelined_timeline.csv"
