#!/usr/bin/env bash
# The north star's reporting matrix on an 8-GPU node: queries/s for N in {10K, 1M, 10M} x 384 at 1, 2, 4 and 8 GPUs
# (one JSON line each, appended to $OUT). The driver's own scaling run uses the default workload (10M) only.
#   bash tools/scale_matrix.sh [out.jsonl]
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$R/gpurun_out/scale_matrix.jsonl}"
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
port=29700
for rows in 10000 1000000 10000000; do
  steps=200; [ "$rows" -le 1000000 ] && steps=1000
  for g in 1 2 4 8; do
    port=$((port + 1))
    if [ "$g" -eq 1 ]; then
      python "$R/bench.py" --gpus 1 --rows "$rows" --steps "$steps" --warmup 20 --no-cpu-baseline >> "$OUT" 2>> "$OUT.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$g" --master-addr 127.0.0.1 --master-port "$port" \
        "$R/bench.py" --gpus "$g" --rows "$rows" --steps "$steps" --warmup 20 --no-cpu-baseline >> "$OUT" 2>> "$OUT.err"
    fi
  done
done
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except Exception:
        continue
    print(f"rows {d['config']['rows']:>9}  gpus {d['n_gpus']}  {d['value']:10.1f} q/s  {d['ms_per_step']:.4f} ms/step  "
          f"scan {d['roofline']['achieved']:.0f} GB/s = {d['roofline']['frac']:.3f} of peak per GPU")
PY
