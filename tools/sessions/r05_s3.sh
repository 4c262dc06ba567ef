#!/bin/bash
# round 5, GPU session 3: the pace gate's cost in the pp kernel (batch_debug 4096 = gate off), 768-d
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s3
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 9 --debug 0 4096 0 4096 4104 4099 4098 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 5 --debug 0 4096 >> "$OUT/bench768.jsonl" 2>> "$OUT/bench768.err"
python - "$OUT/bench768.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
