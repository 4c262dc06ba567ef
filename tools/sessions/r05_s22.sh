#!/bin/bash
# round 5, session 22: the two HIP-event modes against rocprofv3's per-dispatch durations ON THE SAME LAUNCHES (blocking calls, one process
# under rocprofv3 --kernel-trace); pipelined throughput with the in-kernel merge against the two-launch form by store size.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s22}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for rows in 1250000 10000000; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_ev_$rows" -o t -- python "$R/tools/event_modes.py" --rows $rows --n 30 --nq $([ $rows = 1250000 ] && echo 256 || echo 0) > "$OUT/event_modes_$rows.txt" 2> "$OUT/event_modes_$rows.err")
  f=$(find "$OUT/prof_ev_$rows" -name "*kernel_trace.csv" | head -1)
  python - "$f" >> "$OUT/event_modes_$rows.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
scans = [dur(r) for r in rows if "scan_kernel" in r["Kernel_Name"]]
gemms = [dur(r) for r in rows if "batch_gemm_rq_kernel" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
print("rocprofv3 scan dispatches:", len(scans))
for i in range(4, len(scans), 30):
    blk = scans[i:i + 30]
    if blk: print("  block of", len(blk), "mean %.2f us median %.2f min %.2f" % (sum(blk) / len(blk), sorted(blk)[len(blk) // 2], min(blk)))
print("rocprofv3 filtering-GEMM dispatches:", len(gemms))
for i in range(3, len(gemms), 30):
    blk = gemms[i:i + 30]
    if blk: print("  block of", len(blk), "mean %.2f us median %.2f min %.2f" % (sum(blk) / len(blk), sorted(blk)[len(blk) // 2], min(blk)))
PY
  rm -rf "$OUT/prof_ev_$rows"
done
# pipelined (depth 4) single-query throughput: merge in the scan kernel's last arriver (default up to 2 GiB) against the separate merge launch
for rows in 60000 100000 200000 400000 700000 1000000 1250000; do
  for fm in 1 0; do
    timeout 200 python bench.py --gpus 1 --rows $rows --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary --traffic off --tune fuse_merge=$fm \
        --detail-out "$OUT/d.json" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rows $rows fuse_merge $fm  %.1f q/s  %.2f us/step  kernel %.2f us (%s; bracketed %s)  frac %.4f' % (d['value'], d['ms_per_step']*1e3, r['kernel_avg_ms']*1e3, r.get('events'), r.get('kernel_avg_ms_bracketed'), r['frac']))" >> "$OUT/pipelined_merge_forms.txt"
  done
done
rm -f "$OUT/d.json"
cat "$OUT"/event_modes_*.txt "$OUT/pipelined_merge_forms.txt"
