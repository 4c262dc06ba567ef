#!/bin/bash
# round 5, GPU session 2: where the time of the pp kernel (MODE 0 = batch_rega 9) goes at 768-d: batch_debug 8 = no selection,
# 64 = hot test only, 1 = no corpus stream, 2 = no K loop (timing only: results are garbage and fall back)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s2
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 3 --rega 9 --debug 0 8 64 1 2 3 9 0 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
timeout 600 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 1024 --reps 3 --rega 9 --debug 0 8 64 1 2 3 9 0 > "$OUT/bench384.jsonl" 2> "$OUT/bench384.err"
python - "$OUT/bench768.jsonl" "$OUT/bench384.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
