#!/bin/bash
# round 5, GPU session 14: where the 1.25M-row scan's ~25 us of fixed cost per launch go: grid size x in-kernel merge on / off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s14
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for cfg in "grid_blocks=0" "grid_blocks=256" "grid_blocks=384" "grid_blocks=512 --tune fuse_merge=0" "grid_blocks=768 --tune fuse_merge=0" "grid_blocks=1024 --tune fuse_merge=0" "grid_blocks=2048 --tune fuse_merge=0" "grid_blocks=0 --tune query_args=0" "grid_blocks=0"; do
  timeout 300 python bench.py --gpus 1 --rows 2500000 --steps 100 --warmup 10 --no-cpu-baseline --traffic off --secondary s1250k,s1m --tune $cfg 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', '| 2.5M: %.4f ms kernel %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac']), ' | '.join('%s %.4f ms (kernel %.4f frac %.4f)' % (x['name'], x['ms_per_step'], x.get('kernel_avg_ms') or 0, x.get('frac') or 0) for x in d['secondary']))
" | tee -a "$OUT/summary.txt"
done
