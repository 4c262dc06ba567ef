#!/bin/bash
# round 5, session 23: the new tests first, then the whole GPU suite and the default bench command on the build with "merge_overlap_mb".
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s23}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 300 \
    -k "stream_of_scans or kernel_bound or completion_word or litmus" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_new.log"
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; echo "bench rc $?" > "$OUT/bench_n1.rc"
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 600 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
tail -5 "$OUT/pytest_new.log"; tail -15 "$OUT/pytest_gpu.log"; tail -2 "$OUT/smoke.log"; cat "$OUT/bench_n1.json"
