#!/bin/bash
# round 4, GPU session 24: the one-pass planner's gamma quantiles computed once instead of per batch — blocking batched calls
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s24
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for i in 1 2; do
  timeout 600 python bench.py --rows 1000000 --steps 50 --warmup 5 --no-cpu-baseline --traffic off --secondary b1m_q256,b1m_q1024,clustered_k100,c5_shard --detail-out "$OUT/bench_$i.detail.json" 2> /dev/null | tail -1 >> "$OUT/bench.jsonl"
done
python - "$OUT/bench.jsonl" <<'PY' | tee "$OUT/blocking.txt"
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(" | ".join(f"{s['name']} pipelined {s['ms_per_step']*1000:.1f} us blocking {s['blocking_ms']*1000:.1f} us gemm {s['kernel_avg_ms']*1000:.1f} ck {s['ck'][:8]}" for s in d["secondary"]))
PY
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "batch_randomised_soak or batched_full_size_parity or onepass" 2>&1 | tail -2
