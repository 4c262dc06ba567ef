#!/bin/bash
# round 4, GPU session 7: the retry kernel with its row list (was 164 us per launch), DMA pieces spread through the wide kernel's K loop
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s7
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf -s --durations=5 -p no:cacheprovider --timeout 400 \
   -k "device_side_full_retry or resends_parts or falls_back_on_ties or adversarial or variants_agree or randomised_soak or onepass" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 5 --debug 0 768 0 768 0 768 > "$OUT/spread_768.jsonl" 2> "$OUT/ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 10000000 --nq 1024 --reps 3 --rega 5 --debug 0 768 > "$OUT/spread_768_full.jsonl" 2>> "$OUT/ab.err"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_dense" -o p -- python "$R/bench.py" --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k100,dups17 --detail-out "$OUT/bench_dense_detail.json" > "$OUT/bench_dense.json" 2> /dev/null)
find "$OUT/prof_dense" -name "*kernel_stats.csv" -exec cp {} "$OUT/dense_kernel_stats.csv" \; 2>/dev/null; rm -rf "$OUT/prof_dense"
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k100,dups17 --detail-out "$OUT/bench_dense2_detail.json" > "$OUT/bench_dense2.json" 2> /dev/null
timeout 200 python tools/fuzz_batch.py --seconds 90 > "$OUT/fuzz_batch.txt" 2>&1
