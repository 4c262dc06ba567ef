#!/bin/bash
# round 4, GPU session 2: full GPU suite on the new build; 10K-row latency with the query in the kernel arguments; the wide
# 768-d GEMM (store wait moved, split-barrier build) against the K-split kernel; SQ counters of both
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s2
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=12 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && timeout 120 /tmp/latency_c 10000 384 3000 > "$OUT/latency_c.jsonl" 2> "$OUT/latency_c.err"
timeout 120 /tmp/latency_c 10000 768 2000 >> "$OUT/latency_c.jsonl" 2>> "$OUT/latency_c.err"
timeout 120 python tools/reference_harness_bench.py > "$OUT/reference_harness.json" 2>> "$OUT/latency_c.err"
timeout 200 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --secondary s10k,s1m --detail-out "$OUT/bench_s10k_detail.json" > "$OUT/bench_s10k.json" 2> /dev/null
timeout 200 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --secondary s10k --tune query_args=0 --detail-out "$OUT/bench_s10k_qa0_detail.json" > "$OUT/bench_s10k_qa0.json" 2> /dev/null
# wide vs K-split (same process, alternating), then the wide build variants
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 7 5 7 5 > "$OUT/wide_ab.jsonl" 2> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 5 --debug 0 512 256 768 0 512 >> "$OUT/wide_ab.jsonl" 2>> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 5 --rega 5 --debug 8 64 > "$OUT/wide_components.jsonl" 2>> "$OUT/wide_ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 10000000 --nq 1024 --reps 3 --rega 7 5 --debug 0 512 > "$OUT/wide_c5_full.jsonl" 2>> "$OUT/wide_ab.err"
# SQ counters, one pass per kernel (counters only: --pmc with --kernel-trace, nothing else)
for mode in 5 7; do
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$OUT/prof_pmc_$mode" -o g -- python "$R/tools/batch_bench.py" --dims 768 --rows 1250000 --nq 1024 --reps 2 --rega $mode > "$OUT/pmc_$mode.log" 2>&1)
  python tools/pmc_summary.py "$OUT/prof_pmc_$mode" > "$OUT/pmc_sq_768_rega$mode.json" 2>> "$OUT/pmc_$mode.log"
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
      --kernel-trace --output-format csv -d "$OUT/prof_pmc2_$mode" -o g -- python "$R/tools/batch_bench.py" --dims 768 --rows 1250000 --nq 1024 --reps 2 --rega $mode > "$OUT/pmc2_$mode.log" 2>&1)
  python tools/pmc_summary.py "$OUT/prof_pmc2_$mode" > "$OUT/pmc_sq2_768_rega$mode.json" 2>> "$OUT/pmc2_$mode.log"
  find "$OUT/prof_pmc_$mode" "$OUT/prof_pmc2_$mode" -name "*.csv" -size +1M -delete 2>/dev/null
done
