#!/bin/bash
# round 6, session 9: general selection by number of streams; placement probe (several slabs in one process)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s9}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for st in 2 3 4; do
timeout 600 python tools/general_select_bench.py --rows 10000000 --dims 384 --topk 10 300 1000 --steps 40 --streams $st --depth 6 --out "$OUT/general_select_streams.jsonl" > /dev/null 2>> "$OUT/gs.err"
timeout 400 python tools/general_select_bench.py --rows 1000000 --dims 384 --topk 10 300 1000 --steps 200 --streams $st --depth 6 --out "$OUT/general_select_streams.jsonl" > /dev/null 2>> "$OUT/gs.err"
done
python -c "
import json
for l in open('$OUT/general_select_streams.jsonl'):
    d=json.loads(l); print(d['rows'],'streams',d['streams'],'k',d['top_k'],'pip %.4f blk %.4f'%(d['ms_pipelined'],d['ms_blocking']))"
for i in 1 2 3; do
timeout 300 python tools/placement_probe.py --engines 6 --out "$OUT/placement.jsonl" > /dev/null 2>> "$OUT/placement.err"
done
python -c "
import json
for l in open('$OUT/placement.jsonl'):
    d=json.loads(l)
    print('pid',d['pid'],[ (s['engine'],s['us_scan_kernel'],s['us_stream_read']) for s in d['slabs'] if s['rep']==1])"
