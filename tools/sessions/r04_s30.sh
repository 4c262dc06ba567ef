#!/bin/bash
# round 4, GPU session 30: kernel time of the one-launch path at 100K and 1M rows (rocprofv3 kernel trace of tools/latency_c.c, per mode)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s30
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm || exit 1
for n in 100000 1000000; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_$n" -o p -- /tmp/latency_c $n 384 1000 > /dev/null 2> /dev/null)
  python - "$OUT/prof_$n" $n >> "$OUT/kernel_time_by_mode.txt" <<'PY'
import csv, glob, os, sys
n = int(sys.argv[2])
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "scan_kernel" in r["Kernel_Name"] or "merge_keys" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    scans = [r for r in rows if "scan_kernel" in r["Kernel_Name"]]
    merges = [r for r in rows if "merge_keys" in r["Kernel_Name"]]
    per = 1050
    for i in range(per * 8, len(scans), per):          # second corpus (unit gaussian, top-10): modes 0..7
        seg = scans[i + 50:i + per]
        if not seg: continue
        d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
        lo, hi = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
        mm = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in merges if lo <= int(r["Start_Timestamp"]) <= hi)
        print(f"rows {n} mode {(i // per) % 8} {seg[0]['Kernel_Name'][:52]} scan median_ns {d[len(d)//2]} = {n*384*4/d[len(d)//2]:.0f} GB/s"
              + (f"; merge kernel median_ns {mm[len(mm)//2]} x{len(mm)}" if mm else "; no merge kernel"))
PY
  rm -rf "$OUT/prof_$n"
done
cat "$OUT/kernel_time_by_mode.txt"
