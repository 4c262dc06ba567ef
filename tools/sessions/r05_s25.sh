#!/bin/bash
# round 5, session 25: where a blocking batched call (config 3) spends the time around its filtering GEMM: kernel timeline of blocking calls
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s25}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python tools/blocking_batch_timeline.py > "$OUT/blocking.txt" 2> "$OUT/blocking.err"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_bt" -o t -- python "$R/tools/blocking_batch_timeline.py" --calls 40 >> "$OUT/blocking.txt" 2>> "$OUT/blocking.err")
f=$(find "$OUT/prof_bt" -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py "$f" 24 > "$OUT/blocking_timeline.csv"
rm -rf "$OUT/prof_bt"
cat "$OUT/blocking.txt" "$OUT/blocking_timeline.csv"
