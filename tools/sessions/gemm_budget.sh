#!/bin/bash
# Per-phase cycle budget of the filtering GEMM (s_memtime build, "batch_prof_ptr") at BASELINE configs 3 / 5 and at top-100, with the
# product kernel's HIP-event time beside it, and the interleaved A/B of the workgroup barrier (default) against the split tile barrier
# ("batch_rega" 1 / 5) and of the tail pool against fixed shares ("batch_dyn_tail" 1 / 0).
# Output: gpurun_out/$WAX_TAG/phase_budget.jsonl + phase_table.txt + tail_pool_ab.jsonl.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-gemm_budget}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --opts 1 5 --ab-rounds 5 --reps 10 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --topk 100 --opts 1 --ab-rounds 3 --reps 10 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 1024 --opts 1 --ab-rounds 3 --reps 6 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024 --opts 1 5 --ab-rounds 4 --reps 6 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
# the tail pool against fixed shares ("batch_dyn_tail" 1 / 0), interleaved
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --ab-key batch_dyn_tail --opts 0 1 --ab-rounds 6 --reps 10 --out "$OUT/tail_pool_ab.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024 --ab-key batch_dyn_tail --opts 0 1 --ab-rounds 6 --reps 6 --out "$OUT/tail_pool_ab.jsonl" > /dev/null 2>> "$OUT/phase.err"
python tools/phase_table.py "$OUT/phase_budget.jsonl" | tee "$OUT/phase_table.txt"
