#!/bin/bash
# round 4, GPU session 4: device-side full retry; wide-kernel cold path without its store (where do the 130 us go?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s4
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -rf -s --durations=8 -p no:cacheprovider --timeout 400 \
   -k "device_side_full_retry or resends_parts or falls_back_on_ties or adversarial or onepass or submit_collect_device or variants_agree or randomised_soak" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k10,clustered_k100,dups17,detembed --detail-out "$OUT/bench_dense_detail.json" > "$OUT/bench_dense.json" 2> /dev/null
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 6 --rega 5 --debug 0 8192 64 8 0 8192 > "$OUT/wide_store_ab.jsonl" 2> "$OUT/wide_ab.err"
timeout 200 python tools/fuzz_batch.py --seconds 60 > "$OUT/fuzz_batch.txt" 2>&1
