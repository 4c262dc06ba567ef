#!/bin/bash
# round 5, GPU session 20: floor of the expected survivors per query (batch_survivors x k', default 3) against batch time and fallbacks
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s20
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for sv in 3 2 3 2; do
  timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 40 --warmup 8 --no-cpu-baseline --traffic off --secondary b1m_q256,c5_shard,clustered_k10,detembed --tune batch_survivors=$sv --detail-out "$OUT/d.json" 2> /dev/null | tail -1 > "$OUT/line.json"
  python - "$OUT/d.json" $sv <<'PY' | tee -a "$OUT/summary.txt"
import json,sys
d=json.load(open(sys.argv[1]))
print('batch_survivors', sys.argv[2], ' | '.join('%s %.4f ms (gemm %.4f, blocking %.4f, fallbacks %s)' % (x['name'], x['ms_per_step'], x['roofline']['kernel_avg_ms'], x.get('ms_per_step_blocking_call', 0), x.get('certificate_fallbacks')) for x in d['secondary']))
PY
done
rm -f "$OUT/d.json" "$OUT/line.json"
