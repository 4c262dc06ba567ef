#!/bin/bash
# round 5, GPU session 15: sampled-tile budget of the one-pass planner (batch_sample_div; default 32) against batch time, pipelined and blocking
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s15
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for div in 32 64 128 256 16 32; do
  timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 40 --warmup 8 --no-cpu-baseline --traffic off --secondary b1m_q256,b1m_q1024,c5_shard --tune batch_sample_div=$div 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('sample_div $div', ' | '.join('%s %.4f ms (gemm %.4f, blocking %s)' % (x['name'], x['ms_per_step'], x.get('kernel_avg_ms') or 0, x.get('blocking_ms')) for x in d['secondary']))
" | tee -a "$OUT/summary.txt"
done
