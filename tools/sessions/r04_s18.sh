#!/bin/bash
# round 4, GPU session 18: what does the survivor path cost the 384-d filtering GEMM at k = 100 (~4 survivors per wave-tile)?
# batch_debug 0 = product, 8192 = cold path without the global stores, 64 = hot test only, 8 = no selection
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s18
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
for k in 10 100; do
  timeout 300 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 --topk $k --reps 6 --debug 0 8192 64 8 0 2>/dev/null | grep '^{' >> "$OUT/cold_path_k$k.jsonl"
done
python - "$OUT" <<'PY' | tee "$OUT/cold_path.txt"
import json, sys, os
for k in (10, 100):
    for l in open(os.path.join(sys.argv[1], f"cold_path_k{k}.jsonl")):
        d = json.loads(l)
        print("k", k, "debug", d.get("debug"), "gemm_us", d.get("gemm_kernel_us"), "ms_device_call", round(d.get("ms_device_call", 0), 4), {x: d[x] for x in d if "surviv" in x or "fallback" in x})
PY
