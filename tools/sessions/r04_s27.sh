#!/bin/bash
# round 4, GPU session 27: survivors parked in LDS and flushed after the tile loop (new build, 252 VGPRs) against the committed build (230),
# alternating in one session; then the GEMM variants / parity tests on the new build
# (a record of the experiment: it needs the experimental kernel — since removed, profiles/HISTORY.md — built as wax_amd/lib/libwaxhip.so and
# the committed build saved beside it as wax_amd/lib/libwaxhip_head.so.keep)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s27
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
cp wax_amd/lib/libwaxhip.so /tmp/lib_new.so
run() {  # run <tag>
  timeout 300 python bench.py --rows 1000000 --steps 50 --warmup 5 --no-cpu-baseline --traffic off --secondary b1m_q256,b1m_q1024,clustered_k100 --detail-out "$OUT/d.json" 2> /dev/null | tail -1 | sed "s/^/$1 /" >> "$OUT/ab.txt"
}
for i in 1 2; do
  cp /tmp/lib_new.so wax_amd/lib/libwaxhip.so; run parked
  cp wax_amd/lib/libwaxhip_head.so.keep wax_amd/lib/libwaxhip.so; run head
done
cp /tmp/lib_new.so wax_amd/lib/libwaxhip.so
python - "$OUT/ab.txt" <<'PY' | tee "$OUT/ab_table.txt"
import json, sys
for l in open(sys.argv[1]):
    tag, js = l.split(" ", 1)
    d = json.loads(js)
    print(tag, " | ".join(f"{s['name']} pipelined {s['ms_per_step']*1000:.1f} blocking {s['blocking_ms']*1000:.1f} gemm {s['kernel_avg_ms']*1000:.1f} ck {s['ck'][:8]}" for s in d["secondary"]))
PY
timeout 300 python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256 --topk 10 --reps 6 --debug 0 131072 0 131072 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('alone: debug', d.get('debug'), 'gemm_us', round(d.get('gemm_kernel_us', 0), 1), 'ms_device_call', round(d.get('ms_device_call', 0), 4))" | tee -a "$OUT/ab_table.txt"
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "variants_agree or split_barrier or batch_randomised_soak or device_side_full_retry or batched_full_size_parity" 2>&1 | tail -2 | tee -a "$OUT/ab_table.txt"
rm -f "$OUT/d.json"
