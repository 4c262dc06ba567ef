#!/bin/bash
# round 4, GPU session 8: the measured certificate bound — full GPU suite, fuzz, dense secondaries
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s8
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf -s --durations=6 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 300 python tools/fuzz_batch.py --seconds 150 > "$OUT/fuzz_batch.txt" 2>&1
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,b1m_q1024,clustered_k10,clustered_k100,dups17,detembed,c5_shard --detail-out "$OUT/bench_dense_detail.json" > "$OUT/bench_dense.json" 2> /dev/null
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k100,dups17 --tune batch_eps_measured=0 --detail-out "$OUT/bench_dense_worstcase_detail.json" > "$OUT/bench_dense_worstcase.json" 2> /dev/null
