#!/bin/bash
# round 5, GPU session 5: pp kernel with the lean DMA addressing (saddr form, no clamps) against the round-4 wide kernel, 768-d
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s5
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "variants_agree or edge_shapes or randomised_soak" > "$OUT/pytest_variants.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_variants.log"; tail -3 "$OUT/pytest_variants.log"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 5 9 5 9 5 9 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
timeout 600 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 9 --debug 8 64 9 >> "$OUT/bench768.jsonl" 2>> "$OUT/bench768.err"
timeout 600 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 1024 --reps 5 --rega 5 9 5 9 > "$OUT/bench384.jsonl" 2> "$OUT/bench384.err"
python - "$OUT/bench768.jsonl" "$OUT/bench384.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
