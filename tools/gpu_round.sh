#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 traces, tuning sweep.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag> [stages...]
# Everything is written under gpurun_out/<tag>/ ; each stage has its own timeout so a hang
# cannot eat the whole session.
set -u
TAG=${1:-r01}; shift || true
STAGES=${*:-"smoke tests bench prof pmc sweep"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
echo "== $(date) stages: $STAGES" | tee "$OUT/session.log"
rocm-smi --showproductname 2>/dev/null | head -8 >> "$OUT/session.log"
nproc >> "$OUT/session.log"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/session.log"; grep -m1 -o -w "avx512f" /proc/cpuinfo >> "$OUT/session.log"

for s in $STAGES; do
  t0=$(date +%s)
  case $s in
    smoke)
      timeout 600 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; rc=$? ;;
    tests)
      timeout 600 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider --timeout 150 > "$OUT/pytest_gpu.log" 2>&1; rc=$? ;;
    tests3)
      timeout 1500 python -m pytest tests -m gpu -q -rf --durations=25 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1; rc=$? ;;
    onepasstests)
      timeout 900 python -m pytest tests -m gpu -q -x -rf --durations=10 -p no:cacheprovider --timeout 400 \
          -k "onepass or batch_mfma or randomised_soak or variants_agree or edge_shapes or special_values or dot_and_l2" > "$OUT/pytest_onepass.log" 2>&1; rc=$? ;;
    benchsec)
      timeout 600 python bench.py --gpus 1 --no-cpu-baseline --steps 60 --warmup 10 --secondary ${WAX_SEC:-b1m_q256,clustered_k100,dups17} > "$OUT/bench_sec.json" 2> "$OUT/bench_sec.err"; rc=$? ;;
    multiscan)
      timeout 600 python tools/multiscan_bench.py --dims ${WAX_DIMS:-384 768} > "$OUT/multiscan_bench.jsonl" 2> "$OUT/multiscan_bench.err"; rc=$?
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ms" -o ms -- \
          python "$R/tools/multiscan_bench.py" --dims 384 --nq 16 48 --reps 5 > "$OUT/multiscan_prof.log" 2>&1)
      find "$OUT/prof_ms" -name "*kernel_stats.csv" -exec cp {} "$OUT/multiscan_kernel_stats.csv" \; 2>/dev/null
      rm -rf "$OUT/prof_ms" ;;
    multiscanpmc)
      timeout 300 python tools/multiscan_bench.py --dims 384 --nq 2 16 --k 1 10 60 --reps 10 > "$OUT/multiscan_k.jsonl" 2> "$OUT/multiscan_k.err"
      timeout 300 python tools/multiscan_bench.py --dims 384 --rows 4000000 --nq 16 --k 10 --reps 5 >> "$OUT/multiscan_k.jsonl" 2>> "$OUT/multiscan_k.err"
      (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
          --kernel-trace --output-format csv -d "$OUT/prof_mspmc" -o g -- python "$R/tools/multiscan_bench.py" --dims 384 --nq 16 --reps 3 > "$OUT/multiscan_pmc.log" 2>&1); rc=$?
      python tools/pmc_summary.py "$OUT/prof_mspmc" > "$OUT/multiscan_pmc_summary.json" 2>> "$OUT/multiscan_pmc.log"
      rm -rf "$OUT/prof_mspmc" ;;
    mfmaprobe)
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/mfma_probe.hip -o /tmp/mfma_probe > "$OUT/mfma_probe.err" 2>&1 && timeout 120 /tmp/mfma_probe > "$OUT/mfma_probe.jsonl" 2>> "$OUT/mfma_probe.err"; rc=$? ;;
    shardbench)
      timeout 900 python tools/sharded_handle_bench.py --parts ${WAX_PARTS:-A,B,C} > "$OUT/sharded_handle_bench.jsonl" 2> "$OUT/sharded_handle_bench.err"; rc=$? ;;
    fuzz)
      timeout 400 python tools/fuzz_batch.py --seconds ${WAX_FUZZ_S:-120} --seed 7 > "$OUT/fuzz_batch.jsonl" 2> "$OUT/fuzz_batch.err"; rc=$? ;;
    fuzzsharded)
      timeout 500 python tools/fuzz_batch.py --seconds ${WAX_FUZZ_S:-180} --seed ${WAX_FUZZ_SEED:-11} --sharded 0.5 > "$OUT/fuzz_sharded.jsonl" 2> "$OUT/fuzz_sharded.err"; rc=$? ;;
    filtered)
      timeout 300 python tools/filtered_bench.py > "$OUT/filtered_bench.json" 2> "$OUT/filtered_bench.err"; rc=$? ;;
    cpusweep)
      timeout 300 python tools/cpu_sweep.py > "$OUT/cpu_sweep.json" 2> "$OUT/cpu_sweep.err"; rc=$? ;;
    bench)
      timeout 900 python bench.py --gpus 1 > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$? ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o bench -- \
          python "$R/bench.py" --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"); rc=$?
      find "$OUT/prof_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \; 2>/dev/null
      find "$OUT/prof_stats" -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > "$2"' _ {} "$OUT/kernel_trace_head.csv" \; 2>/dev/null
      find "$OUT/prof_stats" -name "*kernel_trace.csv" -delete 2>/dev/null ;;
    pmc)
      (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_pmc" -o pmc -- \
          python "$R/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/pmc_bench.json" 2> "$OUT/pmc.err"); rc=$?
      python tools/pmc_summary.py "$OUT/prof_pmc" > "$OUT/pmc_summary.json" 2>> "$OUT/pmc.err"
      find "$OUT/prof_pmc" -name "*.csv" -size +2M -delete 2>/dev/null ;;
    batch)
      timeout 900 python tools/batch_bench.py --nq 64 256 1024 > "$OUT/batch_bench.log" 2>&1; rc=$? ;;
    trace1m)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_1m" -o g -- \
          python "$R/bench.py" --gpus 1 --rows ${WAX_ROWS:-1000000} --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/trace1m_bench.json" 2> "$OUT/trace1m.err"); rc=$?
      f=$(find "$OUT/prof_1m" -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python "$R/tools/trace_tail.py" "$f" 40 > "$OUT/trace1m_tail.csv" 2>/dev/null
      find "$OUT/prof_1m" -name "*kernel_trace.csv" -delete 2>/dev/null ;;
    refharness)
      timeout 300 python tools/reference_harness_bench.py > "$OUT/reference_harness.json" 2> "$OUT/reference_harness.err"; rc=$? ;;
    hosttrace)
      WAX_HIP_BATCH_TRACE=1 timeout 300 python tools/batch_bench.py --nq 256 1024 --reps 3 > "$OUT/hosttrace.log" 2>&1; rc=$? ;;
    profdefault)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_default" -o bench -- \
          python "$R/bench.py" --gpus 1 > "$OUT/profdefault_bench.json" 2> "$OUT/profdefault.err"); rc=$?
      find "$OUT/prof_default" -name "*kernel_stats.csv" -exec cp {} "$OUT/default_kernel_stats.csv" \; 2>/dev/null
      f=$(find "$OUT/prof_default" -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python "$R/tools/split_kernel_trace.py" "$f" > "$OUT/default_kernel_stats_split.csv" 2>/dev/null
      find "$OUT/prof_default" -name "*kernel_trace.csv" -delete 2>/dev/null
      # the headline alone (no secondary configurations): scan_kernel's --stats row is then one workload
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_headline" -o bench -- \
          python "$R/bench.py" --gpus 1 --no-secondary --no-cpu-baseline > "$OUT/profheadline_bench.json" 2> "$OUT/profheadline.err")
      find "$OUT/prof_headline" -name "*kernel_stats.csv" -exec cp {} "$OUT/headline_kernel_stats.csv" \; 2>/dev/null
      find "$OUT/prof_headline" -name "*kernel_trace.csv" -delete 2>/dev/null ;;
    batchprof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_batch" -o batch -- \
          python "$R/tools/batch_bench.py" --nq 256 --reps 3 > "$OUT/batchprof.log" 2>&1); rc=$?
      find "$OUT/prof_batch" -name "*kernel_stats.csv" -exec cp {} "$OUT/batch_kernel_stats.csv" \; 2>/dev/null
      find "$OUT/prof_batch" -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > "$2"; grep "wax::" "$1" | tail -300 >> "$2"' _ {} "$OUT/batch_kernel_trace.csv" \; 2>/dev/null
      find "$OUT/prof_batch" -name "*kernel_trace.csv" -delete 2>/dev/null ;;
    multi)
      # N>1 code path on a 1-GPU box: two ranks share GPU 0, exchange over gloo (RCCL refuses duplicate GPUs).
      timeout 600 python bench.py --gpus 1 --rows 2000000 --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/multi_n1.json" 2> "$OUT/multi_n1.err"
      WAX_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
          bench.py --gpus 2 --rows 2000000 --steps 50 --warmup 5 --exchange host > "$OUT/multi_n2_host.json" 2> "$OUT/multi_n2_host.err"; rc=$?
      WAX_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29512 \
          bench.py --gpus 3 --rows 2000000 --steps 50 --warmup 5 --exchange host > "$OUT/multi_n3_host.json" 2> "$OUT/multi_n3_host.err"
      timeout 300 python tools/shard_overhead.py > "$OUT/shard_overhead.log" 2>&1
      timeout 300 python tools/batch_bench.py --rows 2000000 --nq 256 --reps 3 > "$OUT/multi_batch_n1.json" 2> "$OUT/multi_batch_n1.err"
      WAX_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 \
          tools/batch_bench.py --gpus 2 --rows 2000000 --nq 256 --reps 3 --exchange host > "$OUT/multi_batch_n2_host.json" 2> "$OUT/multi_batch_n2_host.err"
      grep -h result_checksum "$OUT/multi_batch_n1.json" "$OUT/multi_batch_n2_host.json" | cut -c1-400 >> "$OUT/session.log"
      python - <<PYEOF >> "$OUT/session.log"
import json
for f in ("multi_n1","multi_n2_host","multi_n3_host"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["config"]["last_result_checksum"], d["roofline"]["kernel_avg_ms"])
    except Exception as e: print(f, "ERR", e)
PYEOF
      ;;
    gemmpmc)
      (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
          --kernel-trace --output-format csv -d "$OUT/prof_gemmpmc" -o g -- python "$R/tools/batch_bench.py" --dims ${WAX_DIMS:-384} --nq ${WAX_NQ:-1024} --reps 2 > "$OUT/gemmpmc.log" 2>&1); rc=$?
      python tools/pmc_summary.py "$OUT/prof_gemmpmc" > "$OUT/gemmpmc_summary.json" 2>> "$OUT/gemmpmc.log"
      find "$OUT/prof_gemmpmc" -name "*.csv" -size +1M -delete 2>/dev/null ;;
    gemmfetch)
      # HBM bytes the filtering GEMMs actually fetch (FETCH_SIZE, KiB, x2 on gfx950) against the mirror's size: re-reads by the query groups
      for cfg in "384 1000000 256" "384 1000000 1024" "768 1250000 1024" "768 10000000 1024"; do
        set -- $cfg
        (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_fetch_$1_$2_$3" -o f -- \
            python "$R/tools/batch_bench.py" --dims $1 --rows $2 --nq $3 --reps 2 > "$OUT/gemmfetch_$1_$2_$3.log" 2>&1); rc=$?
        python tools/pmc_summary.py "$OUT/prof_fetch_$1_$2_$3" > "$OUT/gemmfetch_$1_$2_$3.json" 2>> "$OUT/gemmfetch_$1_$2_$3.log"
        rm -rf "$OUT/prof_fetch_$1_$2_$3"
      done ;;
    shardtests)
      timeout 900 python -m pytest tests/test_sharded_engine_gpu.py tests/test_parity_gpu.py -q -m gpu -x -p no:cacheprovider --timeout 400 \
          -k "sharded or bench_contract or submit_collect_device" > "$OUT/pytest_shard.log" 2>&1; rc=$? ;;
    gridsweep)
      # scan grid / pipeline depth under the overlapped product mode (value and pipeline_frac are what moves; frac is per launch)
      for g in 256 384 512 768 1024; do
        timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-secondary --steps 150 --warmup 20 --tune grid_blocks=$g >> "$OUT/gridsweep_10m.jsonl" 2>> "$OUT/gridsweep.err"
        timeout 300 python bench.py --gpus 1 --rows 1000000 --no-cpu-baseline --no-secondary --steps 600 --warmup 50 --tune grid_blocks=$g >> "$OUT/gridsweep_1m.jsonl" 2>> "$OUT/gridsweep.err"
      done
      for d in 2 3 6 8; do
        timeout 300 python bench.py --gpus 1 --rows 1000000 --no-cpu-baseline --no-secondary --steps 600 --warmup 50 --depth $d >> "$OUT/depthsweep_1m.jsonl" 2>> "$OUT/gridsweep.err"
      done; rc=$? ;;
    profchain)
      # the headline with every scan of the timed region chained and timed (one kernel at a time): the run whose rocprofv3 average
      # the per-launch `roofline.frac` of the default command (calibration pass) is compared with
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_chain" -o bench -- \
          python "$R/bench.py" --gpus 1 --chain-timed-region --no-secondary --no-cpu-baseline > "$OUT/profchain_bench.json" 2> "$OUT/profchain.err"); rc=$?
      find "$OUT/prof_chain" -name "*kernel_stats.csv" -exec cp {} "$OUT/chained_kernel_stats.csv" \; 2>/dev/null
      find "$OUT/prof_chain" -name "*kernel_trace.csv" -delete 2>/dev/null ;;
    pmc1m)
      (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_pmc1m" -o pmc -- \
          python "$R/bench.py" --gpus 1 --rows 1000000 --steps 40 --warmup 4 --no-cpu-baseline --no-secondary > "$OUT/pmc1m_bench.json" 2> "$OUT/pmc1m.err"); rc=$?
      python tools/pmc_summary.py "$OUT/prof_pmc1m" > "$OUT/pmc1m_summary.json" 2>> "$OUT/pmc1m.err"
      find "$OUT/prof_pmc1m" -name "*.csv" -size +2M -delete 2>/dev/null ;;
    onepassprof)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_onepass" -o op -- \
          python "$R/tools/batch_bench.py" --nq ${WAX_NQ:-256} --reps 3 > "$OUT/onepassprof.log" 2>&1); rc=$?
      find "$OUT/prof_onepass" -name "*kernel_stats.csv" -exec cp {} "$OUT/onepass_kernel_stats.csv" \; 2>/dev/null
      f=$(find "$OUT/prof_onepass" -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python "$R/tools/trace_tail.py" "$f" 60 > "$OUT/onepass_trace_tail.csv" 2>/dev/null
      find "$OUT/prof_onepass" -name "*kernel_trace.csv" -delete 2>/dev/null ;;
    sweep)
      timeout 1200 python tools/sweep.py --tag "$TAG" > "$OUT/sweep.log" 2>&1; rc=$?
      cp gpurun_out/sweep_$TAG.json "$OUT/" 2>/dev/null ;;
    *) echo "unknown stage $s"; rc=99 ;;
  esac
  echo "== stage $s rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/session.log"
done
tail -5 "$OUT/pytest_gpu.log" 2>/dev/null
cat "$OUT/bench.json" 2>/dev/null
