#!/bin/bash
# round 5, GPU session 11: litmus test; non-temporal tile requests in the rq kernel (batch_debug 8192) at Q = 256 / 1024, 384-d and 768-d
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s11
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x -rf -s -p no:cacheprovider --timeout 200 -k "litmus" > "$OUT/pytest_litmus.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_litmus.log"; grep -a "litmus\]\|passed\|failed" "$OUT/pytest_litmus.log" | tail -4
timeout 600 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 1024 --reps 5 --debug 0 8192 0 8192 0 8192 > "$OUT/bench384.jsonl" 2> "$OUT/bench384.err"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 256 1024 --reps 5 --debug 0 8192 0 8192 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
python - "$OUT/bench384.jsonl" "$OUT/bench768.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
