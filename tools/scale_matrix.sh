#!/usr/bin/env bash
# The north star's reporting matrix on an 8-GPU node: queries/s for N in {10K, 1M, 10M} x 384 (single query) at 1, 2, 4 and 8
# GPUs, plus BASELINE config 5 (10M x 768, 1024 queries per step) at each GPU count. bench.py's compact contract line of every
# run is appended to $OUT; one summary row per (rows, N) is printed at the end — value, ms/step, scan roofline fraction,
# how many ranks RCCL actually joined, and the result checksum, which must be the same at every N for one corpus.
#   bash tools/scale_matrix.sh [out.jsonl] [max_gpus]
# WAX_SCALE_SAME_DEVICE=1: rehearsal on a 1-GPU box (all ranks on GPU 0, host exchange) — checks the plumbing, not xGMI.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$R/gpurun_out/scale_matrix.jsonl}"
MAXG="${2:-8}"
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
: > "$OUT.err"
port=29700
extra=()
if [ -n "${WAX_SCALE_SAME_DEVICE:-}" ]; then export WAX_BENCH_SAME_DEVICE=1; extra=(--exchange host); fi
for rows in ${WAX_SCALE_ROWS:-10000 1000000 10000000}; do
  steps=200; [ "$rows" -le 1000000 ] && steps=1000
  sec=(--no-secondary)
  for g in 1 2 4 8; do
    [ "$g" -gt "$MAXG" ] && continue
    port=$((port + 1))
    # config 5 rides on the 10M-row runs: N = 1 as the c5_full secondary, N > 1 as the sharded c5 secondary
    if [ "$rows" -eq 10000000 ]; then
      if [ "$g" -eq 1 ]; then sec=(--secondary c5_full); else sec=(--secondary c5); fi
    fi
    common=(--gpus "$g" --rows "$rows" --steps "$steps" --warmup 20 --no-cpu-baseline --detail-out "$OUT.detail_${rows}_${g}.json" "${sec[@]}")
    if [ "$g" -eq 1 ]; then
      python "$R/bench.py" "${common[@]}" >> "$OUT" 2>> "$OUT.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$g" --master-addr 127.0.0.1 --master-port "$port" \
        "$R/bench.py" "${common[@]}" "${extra[@]}" >> "$OUT" 2>> "$OUT.err"
    fi
  done
done
python - "$OUT" <<'PY'
import json, sys
seen = {}
for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    c, r = d["config"], d["roofline"]
    same = seen.setdefault(c["rows"], c["checksum"]) == c["checksum"]
    print(f"rows {c['rows']:>9}  gpus {d['n_gpus']}  {d['value']:10.1f} q/s  {d['ms_per_step']:.4f} ms/step  "
          f"scan {r['achieved']:.0f} GB/s = {r['frac']:.3f} of peak per GPU  rccl_ranks {c.get('rccl_ranks')}  "
          f"checksum {c['checksum']} {'ok' if same else 'DIFFERS'}")
    for s in d.get("secondary", []):
        if "error" in s:
            print(f"    {s['name']}: ERROR {s['error']}")
        else:
            k5 = seen.setdefault("c5", s.get("ck")) == s.get("ck")
            print(f"    config 5 ({s['name']}) gpus {s.get('n_gpus', 1)}  {s['value']:12.0f} q/s  {s['ms_per_step']:.3f} ms/step  "
                  f"GEMM frac {s['frac']}  checksum {s.get('ck')} {'ok' if k5 else 'DIFFERS'}")
PY
