#!/bin/bash
# round 4, last session: the full GPU suite, fuzz (also against sharded handles), a long single-query run (completion word) and the
# end-of-round measurement set on the final build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export WAX_TAG=${WAX_TAG:-r04_final5}
OUT=$R/gpurun_out/$WAX_TAG
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=6 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
timeout 400 python tools/fuzz_batch.py --seconds 120 --seed 21 --sharded 0.4 > "$OUT/fuzz_batch_sharded.txt" 2>&1
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && timeout 300 /tmp/latency_c 10000 384 200000 > "$OUT/latency_c_200k_calls.jsonl" 2>&1
bash tools/sessions/r04_final.sh
