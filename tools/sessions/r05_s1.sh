#!/bin/bash
# round 5, GPU session 1: the ping-pong filtering GEMM (batch_gemm_pp_kernel) against the round-4 kernels.
# batch_rega 5 = round-4 default (wide at 768, register staging + split barrier at 384), 9 = one barrier per tile + asm reads,
# 10 = plain ping-pong, 8 = ping-pong with late-half DMA and primed fragment rings (batch_debug 256 / 512: read-ahead variants).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s1
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "variants_agree" > "$OUT/pytest_variants.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_variants.log"
tail -5 "$OUT/pytest_variants.log"
timeout 400 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 5 9 10 8 5 9 10 8 > "$OUT/bench768.jsonl" 2> "$OUT/bench768.err"
timeout 200 python tools/batch_bench.py --rows 1250000 --dims 768 --nq 1024 --reps 5 --rega 8 --debug 256 512 0 >> "$OUT/bench768.jsonl" 2>> "$OUT/bench768.err"
timeout 400 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 1024 --reps 5 --rega 5 9 10 8 5 9 10 8 > "$OUT/bench384.jsonl" 2> "$OUT/bench384.err"
timeout 200 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 1024 --reps 5 --rega 8 --debug 256 512 0 >> "$OUT/bench384.jsonl" 2>> "$OUT/bench384.err"
python - "$OUT/bench768.jsonl" "$OUT/bench384.jsonl" <<'PY' | tee "$OUT/summary.txt"
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(d["dims"], d["rows"], "nq", d["nq"], "rega", d["rega"], "dbg", d["debug"], "gemm_us %.1f" % d["gemm_kernel_us"], "dev_call_ms %.4f" % d["ms_device_call"], "fb", d["fallbacks_rank0"], d["result_checksum"])
PY
