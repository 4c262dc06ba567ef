#!/bin/bash
# round 4, GPU session 1: the wide 768-d GEMM against the K-split kernel, the new sharded-handle tests, the compact bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s1
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf --durations=8 -p no:cacheprovider --timeout 400 \
   -k "variants_agree or split_barrier or randomised_soak or host_staged or blocking_shard or non_zero_ordinal or distinct_devices or full_size_parity or edge_shapes" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
# A/B: wide (rega 5: debug 0 = 3 buffers, 256 = 2 buffers, 512 = 2 chains, 768 = read-ahead 3) vs K-split (rega 7)
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 5 --debug 0 256 512 768 0 256 > "$OUT/wide_ab.jsonl" 2> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 7 5 7 5 >> "$OUT/wide_ab.jsonl" 2>> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 256 --reps 8 --rega 7 5 7 5 >> "$OUT/wide_ab.jsonl" 2>> "$OUT/wide_ab.err"
# selection off / hot test only (where the time goes)
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 5 --rega 5 --debug 8 64 > "$OUT/wide_components.jsonl" 2>> "$OUT/wide_ab.err"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench rc $? bytes $(wc -c < "$OUT/bench_default.json")" > "$OUT/bench_default.rc"
cp bench_detail.json "$OUT/bench_detail.json" 2>/dev/null
