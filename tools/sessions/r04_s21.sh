#!/bin/bash
# round 4, GPU session 21: 10K / 5K / 20K rows — ordinary (mode 1 / 6) against non-temporal (mode 7) row loads in the query-in-arguments kernel, same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s21
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for rep in 1 2; do for n in 5000 10000 20000; do
  timeout 300 /tmp/latency_c $n 384 20000 2>&1 | grep '"unit gaussian"' | grep '"mode": [1467]' >> "$OUT/latency_nt.jsonl"
done; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o p -- /tmp/latency_c 10000 384 3000 > /dev/null 2> "$OUT/prof.err")
python - "$OUT" <<'PY' | tee "$OUT/latency_nt.txt"
import csv, glob, json, os, sys
out = sys.argv[1]
for l in open(os.path.join(out, "latency_nt.jsonl")):
    d = json.loads(l)
    print(d["rows"], "mode", d["mode"], "mean", d["mean_us"], "median", d["median_us"], "p99", d["p99_us"])
for path in glob.glob(os.path.join(out, "prof", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "scan_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = 3050
    for i in range(0, len(rows), per):
        seg = rows[i + 50:i + per]
        if not seg: continue
        dd = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
        print(f"kernel time: corpus {i // (per * 8)} mode {(i // per) % 8} {seg[0]['Kernel_Name'][:58]} median_ns {dd[len(dd)//2]} mean_ns {sum(dd)/len(dd):.0f}")
PY
rm -rf "$OUT/prof" "$OUT/prof.err"
