#!/bin/bash
# round 4, GPU session 6: full GPU suite on the build with the completion word; 10K-row latency; single-query sizes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s6
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && timeout 120 /tmp/latency_c 10000 384 3000 > "$OUT/latency_c.jsonl" 2> "$OUT/latency_c.err"
timeout 120 /tmp/latency_c 10000 768 2000 >> "$OUT/latency_c.jsonl" 2>> "$OUT/latency_c.err"
timeout 120 python tools/reference_harness_bench.py > "$OUT/reference_harness.json" 2>> "$OUT/latency_c.err"
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --secondary s10k,s1m,s1250k --detail-out "$OUT/bench_single_detail.json" > "$OUT/bench_single.json" 2> /dev/null
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --secondary s10k --tune done_flag=0 --detail-out "$OUT/bench_single_noflag_detail.json" > "$OUT/bench_single_noflag.json" 2> /dev/null
timeout 200 python tools/fuzz_batch.py --seconds 100 > "$OUT/fuzz_batch.txt" 2>&1
