#!/bin/bash
# round 5, session 21: kernel-bound HIP events ("time_kernels" = 2: hipExtLaunchKernel start / stop pair) against the hipEventRecord bracket
# ("time_kernels" = 1) and against rocprofv3's per-dispatch durations, on the scan kernel (1.25M / 1M / 10M rows) and the filtering GEMMs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s21}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
stats() {  # stats <name> <cmd...>: rocprofv3 --kernel-trace --stats of a command, keep the kernel_stats csv
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o p -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err")
  find "$OUT/prof_$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${name}_kernel_stats.csv" \; 2>/dev/null
  rm -rf "$OUT/prof_$name"
}
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 300 \
    -k "kernel_bound or completion_word or fused_final_merge_equals" > "$OUT/pytest_timing.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_timing.log"
timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --traffic off --secondary s1250k,s1m,s10k,b1m_q256,b1m_q1024,c5_shard \
    --detail-out "$OUT/bench_events_detail.json" > "$OUT/bench_events.json" 2> "$OUT/bench_events.err"; echo "bench rc $?" > "$OUT/bench_events.rc"
# rocprofv3's own per-dispatch durations of the same kernels, one workload per run, scans chained inside the timed region
stats scan_1250k python "$R/bench.py" --gpus 1 --rows 1250000 --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --traffic off --chain-timed-region --detail-out "$OUT/scan_1250k_detail.json"
stats scan_1m python "$R/bench.py" --gpus 1 --rows 1000000 --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --traffic off --chain-timed-region --detail-out "$OUT/scan_1m_detail.json"
stats gemm_q256 python "$R/bench.py" --gpus 1 --rows 1000000 --steps 20 --warmup 5 --no-cpu-baseline --traffic off --secondary b1m_q256 --detail-out "$OUT/gemm_q256_detail.json"
python - "$OUT" <<'PY'
import json, sys, csv, glob, os
out = sys.argv[1]
try:
    d = json.load(open(os.path.join(out, "bench_events_detail.json")))
    c = d["roofline"]["calibration"]
    print("headline", d["roofline"]["kernel_avg_ms"], c.get("events"), "bracketed", c.get("kernel_avg_ms_bracketed"), "bound", c.get("kernel_avg_ms_kernel_bound"), "frac", d["roofline"]["frac"])
    for s in d.get("secondary", []):
        r = s.get("roofline", {})
        c = r.get("calibration", {})
        print(s.get("name"), "kernel_avg_ms", r.get("kernel_avg_ms"), "events", c.get("events", r.get("events")), "bracketed", c.get("kernel_avg_ms_bracketed", r.get("kernel_avg_ms_bracketed")),
              "frac", r.get("frac"), "ms_per_step", s.get("ms_per_step"), s.get("error"))
except Exception as e:
    print("detail ERR", e)
for f in sorted(glob.glob(os.path.join(out, "*_kernel_stats.csv"))):
    for r in csv.DictReader(open(f)):
        if "scan_kernel" in r["Name"] or "batch_gemm" in r["Name"]:
            print(os.path.basename(f), r["Name"][:70], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
tail -3 "$OUT/pytest_timing.log"
