#!/bin/bash
# round 5, session 30: fragment-ordered query copy ("batch_qfrag" 1 / 0): the new test, kernel durations of a blocking config-3 call, pipelined A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s30}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 200 -k "fragment_ordered or batch_gemm_variants_agree" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_new.log"
for qf in 1 0; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bt$qf" -o t -- python "$R/tools/blocking_batch_timeline.py" --calls 60 --tune batch_qfrag=$qf > "$OUT/blocking_qfrag$qf.txt" 2>> "$OUT/blocking.err")
  f=$(find "$OUT/prof_bt$qf" -name "*kernel_stats.csv" | head -1)
  python - "$f" $qf >> "$OUT/qfrag_kernel_stats.txt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "batch_" in r["Name"] or "pick_tau" in r["Name"]:
        print("batch_qfrag", sys.argv[2], r["Name"].replace("void wax::", "")[:64], "calls", r["Calls"], "avg_us %.2f" % (float(r["AverageNs"]) / 1e3), "min_us %.2f" % (float(r["MinNs"]) / 1e3))
PY
  rm -rf "$OUT/prof_bt$qf"
done
for qf in 1 0 1 0; do
  timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 100 --warmup 10 --no-cpu-baseline --traffic off --tune batch_qfrag=$qf --secondary b1m_q256,b1m_q1024,c5_shard \
      --detail-out "$OUT/d.json" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch_qfrag $qf: ' + '  '.join('%s %.4f ms (gemm %.4f, blocking %.4f, ck %s)' % (s['name'], s['ms_per_step'], s['kernel_avg_ms'], s.get('blocking_ms', 0), s['ck'][:6]) for s in d['secondary']))" >> "$OUT/qfrag_ab.txt"
done
rm -f "$OUT/d.json"
tail -3 "$OUT/pytest_new.log"; cat "$OUT"/blocking_qfrag*.txt "$OUT/qfrag_kernel_stats.txt" "$OUT/qfrag_ab.txt"
