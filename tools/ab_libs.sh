#!/bin/bash
# Interleaved A/B of experiment builds (tools/build_variants.sh) of the filtering GEMM: every library in every round, one process each.
#   bash tools/ab_libs.sh <out.jsonl> <rounds> <rows> <dims> <nq> <topk> name1 name2 ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; ROUNDS=$2; ROWS=$3; DIMS=$4; NQ=$5; TOPK=$6; shift 6
cd "$R"; export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
  for name in "$@"; do
    WAX_HIP_LIB=$R/wax_amd/lib/exp/libwaxhip_$name.so timeout 300 python tools/gemm_phase_budget.py --rows $ROWS --dims $DIMS --nq $NQ --topk $TOPK \
        --opts 1 --ab-rounds 5 --reps 10 --tune "time_kernels=0" --out "$OUT.tmp" > /dev/null 2>> "$OUT.err"
    python - "$OUT.tmp" "$OUT" "$name" "$r" <<'PY'
import json, sys
src, dst, name, rnd = sys.argv[1:5]
try:
    d = json.loads(open(src).read().strip().splitlines()[-1])
    d["lib"], d["round"] = name, int(rnd)
    open(dst, "a").write(json.dumps(d) + "\n")
except Exception as ex:
    print("no record for", name, ex)
PY
    rm -f "$OUT.tmp"
  done
done
python - "$OUT" <<'PY'
import json, sys, collections
by = collections.defaultdict(list)
cyc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l)
    by[d["lib"]].append(d["product_kernel_us_ab_median"])
    if d.get("phases"):
        cyc[d["lib"]].append(d["phases"]["loop"]["early"]["mean_cycles"])
for k, v in by.items():
    print(k, "us per process:", [round(x, 1) for x in v], "loop cycles (prof build):", [round(x) for x in cyc[k]])
PY
