#!/bin/bash
# round 6, session 2: schedule variants of the filtering GEMM ("batch_opt"), phase budget + product-kernel time each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s2}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 600 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --opts 0 1 2 3 7 11 19 27 0 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 600 python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024 --opts 0 1 2 3 7 11 19 27 0 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 600 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 1024 --opts 0 3 11 27 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
python - <<'PY'
import json,os
out=os.environ.get("WAX_TAG","r06_s2")
for l in open(f"gpurun_out/{out}/phase_budget.jsonl"):
    d=json.loads(l); p=d["phases"]
    f=lambda n,w: round(p[n][w]["mean_cycles_per_tile"])
    print(d["dims"],d["nq"],"opt",d["batch_opt"],"prod_us %.1f prof_us %.1f eq %s ghz %.2f | early sel %d wait %d dma %d k %d dmaw %d | late sel %d wait %d dma %d k %d dmaw %d | loop/tile %d"%(
        d["product_kernel_us_hip_events"],d["prof_kernel_us_hip_events"],d["prof_answers_equal_product"],d["shader_clock_ghz_median_wave"],
        f("select","early"),f("wait_arrivals","early"),f("dma_issue","early"),f("kloop","early"),f("dma_wait","early"),
        f("select","late"),f("wait_arrivals","late"),f("dma_issue","late"),f("kloop","late"),f("dma_wait","late"),f("loop","early")))
PY
tail -3 "$OUT/phase.err"
