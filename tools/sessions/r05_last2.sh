#!/bin/bash
# round 5, last build: the rocprofv3 --kernel-trace --stats rows of the default command and of each GEMM workload (refreshes
# z_default_cmd_kernel_stats.csv / z_gemm_rows_per_workload.csv for the build with the fragment-ordered queries)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_last2}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
stats() {
  local name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o p -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err")
  find "$OUT/prof_$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${name}_kernel_stats.csv" \; 2>/dev/null
  rm -rf "$OUT/prof_$name"
}
for w in b1m_q256 c5_shard b1m_q1024 clustered_k100 c5_full; do
  stats gemm_$w python "$R/bench.py" --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --traffic off --secondary $w --detail-out "$OUT/gemm_${w}_detail.json"
done
stats default_cmd python "$R/bench.py" --gpus 1 --no-cpu-baseline --traffic off --detail-out "$OUT/default_cmd_detail.json"
ls "$OUT" | head -40
