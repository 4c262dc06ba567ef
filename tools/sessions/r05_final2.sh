#!/bin/bash
# round 5, second session, final measurements on the build with "merge_overlap_mb" and the kernel-bound HIP events: smoke, the full GPU
# suite, fuzz, the driver's command (record + under rocprofv3), the headline and the 1.25M-row shard chained under rocprofv3, one rocprofv3
# run per GEMM workload, the FETCH_SIZE pass of the headline, the one-process shape (two shards on one GPU), blocking C latency.
# Counter passes are counters only (--pmc with --kernel-trace). Collected by: python tools/collect_profiles.py gpurun_out/r05_final2 profiles/r05
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_final2}
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
timeout 600 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 150 python tools/fuzz_batch.py --seconds 90 --sharded 0.3 --seed 9 > "$OUT/fuzz.txt" 2>&1
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench rc $? bytes $(wc -c < "$OUT/bench_n1.json")" > "$OUT/bench_n1.rc"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$OUT/bench_driver_style_detail.json" > "$OUT/bench_driver_style.json" 2> /dev/null
echo "driver-style rc $? bytes $(wc -c < "$OUT/bench_driver_style.json")" >> "$OUT/bench_n1.rc"
stats headline_chained python "$R/bench.py" --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --chain-timed-region --detail-out "$OUT/headline_chained_detail.json"
stats shard_1250k_chained python "$R/bench.py" --gpus 1 --rows 1250000 --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --traffic off --chain-timed-region --detail-out "$OUT/shard_1250k_chained_detail.json"
stats default_cmd python "$R/bench.py" --gpus 1 --no-cpu-baseline --detail-out "$OUT/default_cmd_detail.json"
for w in b1m_q256 b1m_q1024 c5_shard c5_full clustered_k100; do
  stats gemm_$w python "$R/bench.py" --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --secondary $w --detail-out "$OUT/gemm_${w}_detail.json"
done
pmc fetch_headline FETCH_SIZE python "$R/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --detail-out "$OUT/fetch_headline_detail.json"
WAX_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 40 --warmup 8 --detail-out "$OUT/one_process_2_shards_detail.json" > "$OUT/one_process_2_shards.json" 2> "$OUT/one_process_2_shards.err"
timeout 900 python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --traffic off --secondary s1m,s10k,c5_full --detail-out "$OUT/one_process_n1_detail.json" > "$OUT/one_process_n1.json" 2> /dev/null
gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$R/wax_amd/lib -lm && for n in 10000 100000 1000000; do timeout 120 /tmp/latency_c $n 384 3000 2>/dev/null | grep '"mode": 1,' >> "$OUT/latency_c.jsonl"; done
ls -la "$OUT" > "$OUT/listing.txt"
tail -4 "$OUT/pytest_gpu.log"; tail -3 "$OUT/fuzz.txt"; cat "$OUT/bench_n1.json"
