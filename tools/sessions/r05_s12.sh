#!/bin/bash
# round 5, GPU session 12: 384-d geometry A/B (two builds, alternating): 64-row tiles x 2 buffers (head) against 32-row tiles x 4 buffers (t32)
# (a record of the experiment: needs the two builds saved as wax_amd/lib/libwaxhip_head.so.keep and libwaxhip_t32.so.keep)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s12
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --rows 1000000 --steps 50 --warmup 5 --no-cpu-baseline --traffic off --secondary b1m_q256,b1m_q1024,clustered_k100 --detail-out "$OUT/d.json" 2> /dev/null | tail -1 | sed "s/^/$1 /" >> "$OUT/ab.txt"
}
for i in 1 2; do
  cp wax_amd/lib/libwaxhip_t32.so.keep wax_amd/lib/libwaxhip.so; run t32
  cp wax_amd/lib/libwaxhip_head.so.keep wax_amd/lib/libwaxhip.so; run head
done
python - "$OUT/ab.txt" <<'PY' | tee "$OUT/ab_table.txt"
import json, sys
for l in open(sys.argv[1]):
    tag, js = l.split(" ", 1)
    d = json.loads(js)
    print(tag, " | ".join(f"{s['name']} pipelined {s['ms_per_step']*1000:.1f} blocking {s['blocking_ms']*1000:.1f} gemm {s['kernel_avg_ms']*1000:.1f} ck {s['ck'][:8]}" for s in d["secondary"]))
PY
rm -f "$OUT/d.json"
