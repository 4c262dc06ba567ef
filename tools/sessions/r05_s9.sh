#!/bin/bash
# round 5, GPU session 9: the tests that failed in session 8 (planner threshold in rows, env leak), the new litmus, the fuzz on the single-GEMM
# build, fan-out bench parts A and B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s9
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf -p no:cacheprovider --timeout 400 -k "bench_contract or onepass or litmus or small_store or ticket_path" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"; tail -12 "$OUT/pytest_sel.log"
timeout 200 python tools/fuzz_batch.py --seconds 90 --sharded 0.3 > "$OUT/fuzz.txt" 2>&1; echo "fuzz rc $?"; tail -3 "$OUT/fuzz.txt"
timeout 600 python tools/sharded_handle_bench.py --parts A,B > "$OUT/fanout_AB.jsonl" 2> "$OUT/fanout_AB.err"; cat "$OUT/fanout_AB.jsonl"
