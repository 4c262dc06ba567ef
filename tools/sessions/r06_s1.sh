#!/bin/bash
# round 6, session 1: per-phase cycle budget of the filtering GEMM (configs 3 / 5), general selection by k, the 1M-row bimodality probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s1}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
for i in 1 2; do
timeout 300 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
done
timeout 300 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 1024 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 300 python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 300 python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256 --topk 100 --out "$OUT/phase_budget.jsonl" > /dev/null 2>> "$OUT/phase.err"
timeout 400 python tools/general_select_bench.py --rows 1000000 --dims 384 --topk 10 100 192 195 300 1000 5000 --out "$OUT/general_select.jsonl" > /dev/null 2> "$OUT/gs.err"
timeout 600 python tools/general_select_bench.py --rows 10000000 --dims 384 --topk 10 192 195 300 1000 --steps 30 --out "$OUT/general_select.jsonl" > /dev/null 2>> "$OUT/gs.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_gs" -o gs -- python $R/tools/general_select_bench.py --rows 10000000 --dims 384 --topk 300 --steps 20 > /dev/null 2>> "$OUT/gs.err"
cd "$R"
for i in $(seq 0 11); do
WAX_PROBE_PAD_KB=$((i * 832)) timeout 200 python tools/bimodal_probe.py --rows 1000000 --out "$OUT/bimodal.jsonl" > /dev/null 2>> "$OUT/bimodal.err"
done
find "$OUT/prof_gs" -name "*kernel_stats.csv" -exec cp {} "$OUT/gs_k300_kernel_stats.csv" \;
rm -rf "$OUT/prof_gs"
tail -2 "$OUT/smoke.log"; cat "$OUT/phase_budget.jsonl" | cut -c1-1500; cat "$OUT/general_select.jsonl"; cat "$OUT/bimodal.jsonl"; tail -5 "$OUT"/*.err
