#!/bin/bash
# round 5, session 28: batched secondaries with the timed region in product mode (GEMMs timed in a calibration pass behind it), 2 / 3 / 4 batches in flight
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s28}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for d in 2 3 4 2 3; do
  timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 100 --warmup 10 --no-cpu-baseline --traffic off --batch-depth $d --secondary b1m_q256,b1m_q1024,c5_shard,clustered_k100 \
      --detail-out "$OUT/d.json" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch depth $d: ' + '  '.join('%s %.4f ms (gemm %.4f, bracketed %s, blocking %.4f, ck %s)' % (s['name'], s['ms_per_step'], s['kernel_avg_ms'], s.get('bracketed_ms'), s.get('blocking_ms', 0), s['ck'][:6]) for s in d['secondary']))" >> "$OUT/batch_depth.txt"
done
rm -f "$OUT/d.json"; cat "$OUT/batch_depth.txt"
