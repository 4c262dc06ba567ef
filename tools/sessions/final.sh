#!/bin/bash
# The end-of-round measurement set, one gpurun call (tools/collect_profiles.py copies its summaries into profiles/<round>/):
# the default bench command (record + under rocprofv3), the headline chained under rocprofv3, one rocprofv3 run per GEMM workload and
# one for the general selection, FETCH_SIZE passes (headline scan, the rq GEMM at 768-d and 384-d), SQ counters of the rq kernel, the
# one-process shape of bench.py (two shards on one GPU), the N > 1 rehearsal of tools/scale_matrix.sh on one GPU, the full GPU suite,
# smoke, fuzz, blocking C latency, the filtering GEMM's phase budget. Counter passes are counters only (--pmc with --kernel-trace).
#   gpurun --timeout 5400 -- 'WAX_TAG=r06_final bash tools/sessions/final.sh'
# WAX_SHORT=1: correctness + the bench records + the rocprofv3 rows of the headline and of configs 3 / 5 only (about 17 minutes).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-final}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
stats() {  # stats <name> <cmd...>: rocprofv3 --kernel-trace --stats of a command, keep the kernel_stats csv
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o p -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err")
  find "$OUT/prof_$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${name}_kernel_stats.csv" \; 2>/dev/null
  rm -rf "$OUT/prof_$name"
}
pmc() {    # pmc <name> <counters> <cmd...>
  local name=$1 ctrs=$2; shift 2
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/prof_$name" -o p -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err")
  python tools/pmc_summary.py "$OUT/prof_$name" > "$OUT/${name}.json" 2>> "$OUT/$name.err"
  rm -rf "$OUT/prof_$name"
}
# 0. correctness first
timeout 600 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 200 python tools/fuzz_batch.py --seconds 120 --sharded 0.3 --seed 5 > "$OUT/fuzz.txt" 2>&1
# 1. the driver's command, twice (record)
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench rc $? bytes $(wc -c < "$OUT/bench_n1.json")" > "$OUT/bench_n1.rc"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$OUT/bench_driver_style_detail.json" > "$OUT/bench_driver_style.json" 2> /dev/null
echo "driver-style rc $? bytes $(wc -c < "$OUT/bench_driver_style.json")" >> "$OUT/bench_n1.rc"
# 2. rocprofv3 summaries
stats headline_chained python "$R/bench.py" --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --chain-timed-region --detail-out "$OUT/headline_chained_detail.json"
[ -z "${WAX_SHORT:-}" ] && stats default_cmd python "$R/bench.py" --gpus 1 --no-cpu-baseline --detail-out "$OUT/default_cmd_detail.json"
WORKLOADS="b1m_q256 b1m_q1024 c5_shard c5_full clustered_k100 s10m_k300 s1m_k1000"
[ -n "${WAX_SHORT:-}" ] && WORKLOADS="b1m_q256 c5_shard"
for w in $WORKLOADS; do
  stats gemm_$w python "$R/bench.py" --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary $w --detail-out "$OUT/gemm_${w}_detail.json"
done
if [ -n "${WAX_SHORT:-}" ]; then
  timeout 200 python tools/fuzz_select.py --seconds 100 --seed 17 > "$OUT/fuzz_select.txt" 2>&1
  for r in 1 2; do python tools/blocking_batch_timeline.py --calls 400; done 2>/dev/null > "$OUT/blocking_call.txt"
  ls -la "$OUT" > "$OUT/listing.txt"; tail -4 "$OUT/pytest_gpu.log"; cat "$OUT/bench_n1.json"; exit 0
fi
# 3. counters
pmc fetch_headline FETCH_SIZE python "$R/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --detail-out "$OUT/fetch_headline_detail.json"
pmc fetch_768_shard FETCH_SIZE python "$R/tools/batch_bench.py" --dims 768 --rows 1250000 --nq 1024 --reps 2
pmc fetch_768_full FETCH_SIZE python "$R/tools/batch_bench.py" --dims 768 --rows 10000000 --nq 1024 --reps 2
pmc fetch_384_q256 FETCH_SIZE python "$R/tools/batch_bench.py" --dims 384 --rows 1000000 --nq 256 1024 --reps 2
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc sq_768 "$SQ1" python "$R/tools/batch_bench.py" --dims 768 --rows 1250000 --nq 1024 --reps 2
pmc sq2_768 "SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS" python "$R/tools/batch_bench.py" --dims 768 --rows 1250000 --nq 1024 --reps 2
pmc sq_384 "$SQ1" python "$R/tools/batch_bench.py" --dims 384 --rows 1000000 --nq 256 1024 --reps 2
# 4. the one-process shape (what `python bench.py --gpus N` runs), two shards on this one GPU: 10M + 1M + 10K + config 5, checksums = N = 1
WAX_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 40 --warmup 8 --detail-out "$OUT/one_process_2_shards_detail.json" > "$OUT/one_process_2_shards.json" 2> "$OUT/one_process_2_shards.err"
timeout 900 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --traffic off --secondary s1m,s10k,c5_full --detail-out "$OUT/one_process_n1_detail.json" > "$OUT/one_process_n1.json" 2> /dev/null
# 5. the scaling matrix script, rehearsed on one GPU (all ranks on GPU 0, host exchange): plumbing only
WAX_SCALE_SAME_DEVICE=1 WAX_SCALE_ROWS="10000 1000000" timeout 600 bash tools/scale_matrix.sh "$OUT/scale_rehearsal.jsonl" 2 > "$OUT/scale_rehearsal.txt" 2>&1
rm -f "$OUT"/scale_rehearsal.jsonl.detail_*
# 6. blocking C calls, fan-out
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && for n in 10000 100000 1000000; do timeout 120 /tmp/latency_c $n 384 3000 2>/dev/null | grep '"mode": 1,' >> "$OUT/latency_c.jsonl"; done
timeout 600 python tools/sharded_handle_bench.py --parts A,B,C > "$OUT/fanout_ABC.jsonl" 2> "$OUT/fanout_ABC.err"
WAX_TAG=${WAX_TAG:-final} bash tools/sessions/gemm_budget.sh > /dev/null 2>&1
ls -la "$OUT" > "$OUT/listing.txt"
tail -4 "$OUT/pytest_gpu.log"; cat "$OUT/bench_n1.json"
