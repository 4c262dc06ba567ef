set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r06_fetch; mkdir -p $OUT; export TMPDIR=/tmp
for rega in 1 5; do
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_$rega" -o p -- python $R/tools/batch_bench.py --dims 768 --rows 10000000 --nq 1024 --reps 2 --rega $rega > "$OUT/out_$rega.txt" 2> "$OUT/err_$rega.txt")
python tools/pmc_summary.py "$OUT/prof_$rega" > "$OUT/fetch_$rega.json" 2>> "$OUT/err_$rega.txt"
rm -rf "$OUT/prof_$rega"
python - <<PY
import json
d=json.load(open("$OUT/fetch_$rega.json"))
for k,v in d.items():
    if "batch_gemm_rq" in k and "FETCH_SIZE" in v: print("rega $rega", k[:90], v["FETCH_SIZE"]["launches"], round(v["FETCH_SIZE"]["hbm_bytes_per_launch_corrected"]/1e9,3), "GB")
PY
grep -o '"gemm_kernel_us": [0-9.]*' "$OUT/out_$rega.txt" | head -2
done
