#!/bin/bash
# round 5, session 26: the two-launch form of a pipelined scan with the query uploaded (default) or in the kernel arguments ("query_args" = 2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r05_s26}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for rep in 1 2; do
for rows in 1000000 1250000 10000000; do
  steps=2000; [ $rows = 10000000 ] && steps=300
  for qa in 1 2; do
    timeout 200 python bench.py --gpus 1 --rows $rows --steps $steps --warmup 100 --no-cpu-baseline --no-secondary --traffic off --tune query_args=$qa \
        --detail-out "$OUT/d.json" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rows $rows query_args $qa  %.1f q/s  %.2f us/step  kernel %.2f us (%s; bracketed %s)  frac %.4f' % (d['value'], d['ms_per_step']*1e3, r['kernel_avg_ms']*1e3, r.get('events'), r.get('kernel_avg_ms_bracketed'), r['frac']))" >> "$OUT/query_args_two_launch.txt"
  done
done
done
rm -f "$OUT/d.json"; cat "$OUT/query_args_two_launch.txt"
