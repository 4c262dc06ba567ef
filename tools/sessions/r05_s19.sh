#!/bin/bash
# round 5, GPU session 19: pace gate only on long launches (> 1024 tiles per workgroup): FETCH_SIZE and time at 1.25M x 768 (gate now off there)
# and at 10M x 768 (gate on); eight 768-d shards on one GPU again
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s19
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for rows in 1250000 10000000; do
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_f" -o p -- python "$R/tools/batch_bench.py" --dims 768 --rows $rows --nq 1024 --reps 3 > "$OUT/fetch_$rows.out" 2> "$OUT/fetch_$rows.err")
python tools/pmc_summary.py "$OUT/prof_f" > "$OUT/fetch_$rows.json"; rm -rf "$OUT/prof_f"
python - "$OUT/fetch_$rows.json" $rows <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'batch_gemm_rq' in k and 'FETCH_SIZE' in v and 'false, true' in k: print(sys.argv[2], k[-48:], v['FETCH_SIZE']['launches'], v['FETCH_SIZE']['hbm_bytes_per_launch_corrected'], v['FETCH_SIZE']['hbm_bytes_per_launch_corrected']/(int(sys.argv[2])*768*2))
PY
done
timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 40 --warmup 8 --no-cpu-baseline --traffic off --secondary c5_shard,c5_full 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' | '.join('%s %.4f ms (gemm %.4f, frac %.4f, blocking %s)' % (x['name'], x['ms_per_step'], x.get('kernel_avg_ms') or 0, x.get('frac') or 0, x.get('blocking_ms')) for x in d['secondary']))"
timeout 600 python tools/sharded_handle_bench.py --parts C 2>/dev/null
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 300 -k "pace_gate or variants_agree or config5" 2>&1 | tail -2
