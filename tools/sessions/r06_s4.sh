#!/bin/bash
# round 6, session 4: lane-parallel cold path (256) on top of 32 / 96; batched GPU tests with the new variant as the default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s4}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --opts 0 32 96 288 352 --ab-rounds 6 --reps 10 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --topk 100 --opts 0 32 288 352 --ab-rounds 5 --reps 10 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024 --opts 0 32 288 352 --ab-rounds 5 --reps 6 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
python tools/phase_table.py "$OUT/phase_budget.jsonl" | tee "$OUT/phase_table.txt"
WAX_HIP_BATCH_OPT=352 timeout 1500 python -m pytest tests -m gpu -q -x -k "batch or config or certificate or retry or fuzz or gemm" -p no:cacheprovider --timeout 400 > "$OUT/pytest_batch_opt352.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_batch_opt352.log"
tail -5 "$OUT/pytest_batch_opt352.log"
WAX_HIP_BATCH_OPT=352 timeout 300 python tools/fuzz_batch.py --seconds 60 > "$OUT/fuzz_opt352.txt" 2>&1; tail -3 "$OUT/fuzz_opt352.txt"
tail -3 "$OUT/phase.err"
