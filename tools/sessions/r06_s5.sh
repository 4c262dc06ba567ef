#!/bin/bash
# round 6, session 5: the whole GPU suite and the bench on the build with the new hot test, the 11-bit general selection and the bench changes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s5}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
timeout 400 python tools/general_select_bench.py --rows 1000000 --dims 384 --topk 10 192 195 300 1000 5000 10000 --out "$OUT/general_select.jsonl" > /dev/null 2> "$OUT/gs.err"
timeout 600 python tools/general_select_bench.py --rows 10000000 --dims 384 --topk 10 195 300 1000 10000 --steps 30 --out "$OUT/general_select.jsonl" > /dev/null 2>> "$OUT/gs.err"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_gs" -o gs -- python $R/tools/general_select_bench.py --rows 10000000 --dims 384 --topk 300 --steps 20 > /dev/null 2>> "$OUT/gs.err")
find "$OUT/prof_gs" -name "*kernel_stats.csv" -exec cp {} "$OUT/gs_k300_10m_kernel_stats.csv" \;
rm -rf "$OUT/prof_gs"
cat "$OUT/general_select.jsonl"
timeout 2400 python -m pytest tests -m gpu -q -rf --durations=8 -p no:cacheprovider --timeout 500 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
tail -30 "$OUT/pytest_gpu.log"
timeout 900 python bench.py --gpus 1 --detail-out "$OUT/bench_n1_detail.json" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench rc $? bytes $(wc -c < "$OUT/bench_n1.json")"; cat "$OUT/bench_n1.json"
tail -3 "$OUT/smoke.log"
