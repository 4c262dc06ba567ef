#!/bin/bash
# round 4, GPU session 3: SGPR survivor counters + read-ahead 3 in the wide 768-d GEMM, the finish kernel's inline full retry
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s3
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -rf -s --durations=8 -p no:cacheprovider --timeout 400 \
   -k "inline_full_retry or variants_agree or split_barrier or randomised_soak or resends_parts or full_size_parity or falls_back_on_ties or adversarial or onepass or config5 or submit_collect_device or edge_shapes" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 5 --debug 0 768 0 768 512 256 7 > "$OUT/wide_ab.jsonl" 2> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 7 5 >> "$OUT/wide_ab.jsonl" 2>> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 5 --rega 5 --debug 8 64 > "$OUT/wide_components.jsonl" 2>> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 10000000 --nq 1024 --reps 3 --rega 7 5 > "$OUT/wide_c5_full.jsonl" 2>> "$OUT/wide_ab.err"
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k10,clustered_k100,dups17,c5_shard --detail-out "$OUT/bench_dense_detail.json" > "$OUT/bench_dense.json" 2> /dev/null
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k100,dups17 --tune batch_retry=2 --detail-out "$OUT/bench_dense_hostretry_detail.json" > "$OUT/bench_dense_hostretry.json" 2> /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lat" -o lat -- /bin/bash -c "gcc -O2 -I$R/include $R/tools/latency_c.c -o /tmp/latency_c -L$R/wax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && /tmp/latency_c 10000 384 500" > "$OUT/latency_prof.log" 2>&1)
find "$OUT/prof_lat" -name "*kernel_stats.csv" -exec cp {} "$OUT/latency_kernel_stats.csv" \; 2>/dev/null
find "$OUT/prof_lat" -name "*.csv" -size +1M -delete 2>/dev/null
