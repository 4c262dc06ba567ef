#!/bin/bash
# A/B of the filtering GEMM's tail pool ("batch_dyn_tail" 0 / 1) on one box, interleaved, with the per-phase budget of both, after the
# batched parity tests. Output: gpurun_out/$WAX_TAG/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-dyn_tail}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "batch or Batch or gemm or onepass or certif" > "$OUT/pytest_batch.txt" 2>&1
tail -3 "$OUT/pytest_batch.txt"
for cfg in "1000000 384 256 10 10" "1000000 384 256 100 10" "1000000 384 1024 10 6" "1250000 768 1024 10 6" "1250000 768 256 10 6" "2000000 128 256 10 6"; do
  set -- $cfg
  timeout 900 python tools/gemm_phase_budget.py --rows $1 --dims $2 --nq $3 --topk $4 --ab-key batch_dyn_tail --opts 0 1 --ab-rounds 6 --reps $5 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
done
python tools/phase_table.py "$OUT/phase_budget.jsonl" | tee "$OUT/phase_table.txt"
tail -5 "$OUT/phase.err"
