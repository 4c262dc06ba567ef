#!/bin/bash
# round 4, GPU session 5: asymmetric wave priority in the filtering GEMMs, device retry for k' > 192, 10K-row latency sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s5
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf -s --durations=5 -p no:cacheprovider --timeout 400 \
   -k "device_side_full_retry or resends_parts or falls_back_on_ties or adversarial or query_in_kernel_arguments or fused_final_merge" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 5 --debug 0 32768 0 32768 0 32768 > "$OUT/prio_768.jsonl" 2> "$OUT/ab.err"
timeout 300 python tools/batch_bench.py --dims 384 --rows 1000000 --nq 256 1024 --reps 10 --rega 5 --debug 0 32768 0 32768 > "$OUT/prio_384.jsonl" 2>> "$OUT/ab.err"
timeout 300 python tools/batch_bench.py --dims 384 --rows 1000000 --nq 256 --reps 10 --rega 1 --debug 0 32768 0 32768 >> "$OUT/prio_384.jsonl" 2>> "$OUT/ab.err"
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && timeout 120 /tmp/latency_c 10000 384 3000 > "$OUT/latency_c.jsonl" 2> "$OUT/latency_c.err"
timeout 200 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --secondary s10k --detail-out "$OUT/bench_s10k_detail.json" > "$OUT/bench_s10k.json" 2> /dev/null
timeout 200 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --secondary s10k --tune grid_blocks=80 --rows 1000000 --detail-out "$OUT/bench_s10k_g80_detail.json" > "$OUT/bench_s10k_g80.json" 2> /dev/null
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary b1m_q256,clustered_k100,dups17 --detail-out "$OUT/bench_dense_detail.json" > "$OUT/bench_dense.json" 2> /dev/null
