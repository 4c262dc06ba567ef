#!/bin/bash
# round 5, GPU session 7: the pp kernel with the split barrier (rega 12) in the PIPELINED bench setting (two batches in flight) against
# the round-4 default (rega 5); at 384-d also with two LDS tile buffers (debug 512: 100 KB instead of 150 KB per workgroup)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s7
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for cfg in "batch_rega=5" "batch_rega=12" "batch_rega=12 --tune batch_debug=512" "batch_rega=5" "batch_rega=12" "batch_rega=12 --tune batch_debug=512"; do
  tag=$(echo "$cfg" | tr -c 'a-z0-9=' '_')
  timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 40 --warmup 10 --no-cpu-baseline --traffic off --secondary b1m_q256,b1m_q1024,c5_shard,clustered_k100 --tune $cfg --detail-out "$OUT/detail_$tag.json" 2> "$OUT/bench_$tag.err" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', ' | '.join('%s %.4f ms (gemm %.4f, blocking %s)' % (x['name'], x['ms_per_step'], x.get('kernel_avg_ms') or 0, x.get('blocking_ms')) for x in d['secondary']))
" | tee -a "$OUT/summary.txt"
done
timeout 300 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 1024 --reps 5 --rega 12 --debug 0 256 512 0 256 512 > "$OUT/bench384.jsonl" 2> "$OUT/bench384.err"
python - "$OUT/bench384.jsonl" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
