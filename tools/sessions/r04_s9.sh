#!/bin/bash
# round 4, GPU session 9: the wide kernel's pace gate — traffic and time at 10M x 768 (gate on / off = batch_debug bit 12), parity
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s9
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rf --durations=5 -p no:cacheprovider --timeout 400 \
   -k "variants_agree or randomised_soak or full_size_parity_on_one_gpu or config5 or split_barrier" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
timeout 400 python tools/batch_bench.py --dims 768 --rows 10000000 --nq 1024 --reps 3 --rega 5 --debug 0 4096 0 4096 > "$OUT/gate_full.jsonl" 2> "$OUT/ab.err"
timeout 300 python tools/batch_bench.py --dims 768 --rows 1250000 --nq 1024 --reps 8 --rega 5 --debug 0 4096 0 4096 > "$OUT/gate_shard.jsonl" 2>> "$OUT/ab.err"
for dbg in 0 4096; do
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_f$dbg" -o p -- python "$R/tools/batch_bench.py" --dims 768 --rows 10000000 --nq 1024 --reps 2 --rega 5 --debug $dbg > "$OUT/fetch_$dbg.out" 2> "$OUT/fetch_$dbg.err")
  python tools/pmc_summary.py "$OUT/prof_f$dbg" > "$OUT/fetch_full_debug$dbg.json" 2>> "$OUT/fetch_$dbg.err"; rm -rf "$OUT/prof_f$dbg"
done
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_fs" -o p -- python "$R/tools/batch_bench.py" --dims 768 --rows 1250000 --nq 1024 --reps 2 --rega 5 > "$OUT/fetch_s.out" 2> "$OUT/fetch_s.err")
python tools/pmc_summary.py "$OUT/prof_fs" > "$OUT/fetch_shard.json" 2>> "$OUT/fetch_s.err"; rm -rf "$OUT/prof_fs"
