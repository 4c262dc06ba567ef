#!/bin/bash
# round 5, GPU session 13: the pace gate counts only RUNNING workgroups (and polls for ~70 us at most): fan-out part C (eight 768-d shards on one
# GPU, 1 024 queries: 157 ms with the round-4 gate), the 768-d GEMM alone, FETCH_SIZE of config 5 whole
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s13
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 600 python tools/sharded_handle_bench.py --parts C > "$OUT/fanout_C.jsonl" 2> "$OUT/fanout_C.err"; cat "$OUT/fanout_C.jsonl"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --debug 0 4096 0 4096 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
python - "$OUT/bench768.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_f" -o p -- python "$R/tools/batch_bench.py" --dims 768 --rows 10000000 --nq 1024 --reps 2 > "$OUT/fetch_768_full.out" 2> "$OUT/fetch_768_full.err")
python tools/pmc_summary.py "$OUT/prof_f" > "$OUT/fetch_768_full.json"; rm -rf "$OUT/prof_f"
python - "$OUT/fetch_768_full.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'batch_gemm_rq' in k and 'FETCH_SIZE' in v: print(k[-60:], v['FETCH_SIZE']['launches'], v['FETCH_SIZE']['hbm_bytes_per_launch_corrected'])
PY
timeout 300 python bench.py --gpus 1 --rows 1000000 --steps 30 --warmup 6 --no-cpu-baseline --traffic off --secondary c5_shard,c5_full 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' | '.join('%s %.4f ms (gemm %.4f, blocking %s)' % (x['name'], x['ms_per_step'], x.get('kernel_avg_ms') or 0, x.get('blocking_ms')) for x in d['secondary']))"
