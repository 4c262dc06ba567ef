#!/bin/bash
# round 5, session 24: the sharded handle (8 shards on one GPU) and the two-rank rehearsal with the "merge_overlap_mb" rule; the repaired test.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s24}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "query_in_kernel_arguments or stream_of_scans" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_new.log"
for mb in 400 0; do
  timeout 600 python tools/sharded_handle_bench.py --parts B --tune merge_overlap_mb=$mb 2>> "$OUT/fanout.err" | sed "s/^/merge_overlap_mb=$mb /" >> "$OUT/fanout_B.jsonl"
done
WAX_SCALE_SAME_DEVICE=1 WAX_SCALE_ROWS="10000 1000000 10000000" timeout 900 bash tools/scale_matrix.sh "$OUT/scale_rehearsal.jsonl" 2 > "$OUT/scale_rehearsal.txt" 2>&1
rm -f "$OUT"/scale_rehearsal.jsonl.detail_*
tail -3 "$OUT/pytest_new.log"; cat "$OUT/fanout_B.jsonl" "$OUT/scale_rehearsal.txt"; tail -5 "$OUT/scale_rehearsal.jsonl.err"
