#!/bin/bash
# round 6, session 3: VALU-only hot test (32), returning arrive (64), both (96), + late select after DMA (97), 32 + select priority (34)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s3}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --opts 0 32 64 96 97 34 --ab-rounds 7 --reps 10 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024 --opts 0 32 64 96 97 34 --ab-rounds 5 --reps 6 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 900 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --topk 100 --opts 0 32 96 --ab-rounds 5 --reps 10 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
python tools/phase_table.py "$OUT/phase_budget.jsonl"
tail -3 "$OUT/phase.err"
