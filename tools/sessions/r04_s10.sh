#!/bin/bash
# round 4, GPU session 10: is the 10M x 768 re-fetch (17 - 25 GB against 15.36 GB in two of three passes) cured by the pace gate?
# three counters-only passes each with the gate on (batch_debug 0) and off (4096), per-launch values kept
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s10
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for i in 1 2 3; do
  for dbg in 0 4096; do
    (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_${i}_$dbg" -o p -- python "$R/tools/batch_bench.py" --dims 768 --rows 10000000 --nq 1024 --reps 2 --rega 5 --debug $dbg > "$OUT/fetch_${i}_$dbg.out" 2> "$OUT/fetch_${i}_$dbg.err")
    python - "$OUT/prof_${i}_$dbg" "$i" "$dbg" >> "$OUT/fetch_per_launch.txt" <<'PY'
import csv, glob, os, sys
vals = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "batch_gemm_wide" in (r.get("Kernel_Name") or "") and (r.get("Counter_Name") == "FETCH_SIZE"):
            vals.append(float(r["Counter_Value"]) * 2048 / 1e9)
print(f"pass {sys.argv[2]} batch_debug {sys.argv[3]}: GB per launch " + " ".join(f"{v:.2f}" for v in vals))
PY
    rm -rf "$OUT/prof_${i}_$dbg"
  done
done
cat "$OUT/fetch_per_launch.txt"
