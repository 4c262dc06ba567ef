#!/bin/bash
# round 5, GPU session 6: pp kernel with the split barrier (rega 12) against MODE 0 (rega 9) and round 4 (rega 5); cached DMA offsets at 384-d;
# survivor-count sweep at 768-d
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s6
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "variants_agree" > "$OUT/pytest_variants.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_variants.log"; tail -3 "$OUT/pytest_variants.log"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 5 9 12 5 9 12 9 12 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 9 --survivors 130 200 400 800 >> "$OUT/bench768.jsonl" 2>> "$OUT/bench768.err"
timeout 600 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 1024 --reps 5 --rega 5 9 12 5 9 12 > "$OUT/bench384.jsonl" 2> "$OUT/bench384.err"
timeout 600 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 1024 --reps 5 --rega 12 --debug 256 >> "$OUT/bench384.jsonl" 2>> "$OUT/bench384.err"
python - "$OUT/bench768.jsonl" "$OUT/bench384.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "surv", d["survivors"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
