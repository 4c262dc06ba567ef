set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r06_ab5; mkdir -p $OUT
for cfg in "1000000 384 256 10" "1000000 384 256 100" "1000000 384 1024 10" "1250000 768 1024 10"; do set -- $cfg
timeout 600 python tools/gemm_phase_budget.py --rows $1 --dims $2 --nq $3 --topk $4 --opts 5 1 --ab-rounds 7 --reps 10 --out $OUT/rega.jsonl > /dev/null 2>> $OUT/err
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_ab5/rega.jsonl"):
    d=json.loads(l); print(d["dims"],d["nq"],"k",d["topk"],"batch_rega",d["batch_opt"],"median",round(d["product_kernel_us_ab_median"],1),d["product_kernel_us_ab_rounds"])
PY
for r in 1 2; do for rega in 5 1; do
timeout 600 python bench.py --gpus 1 --rows 1000000 --steps 60 --warmup 10 --no-cpu-baseline --traffic off --tune batch_rega=$rega --secondary b1m_q256,b1m_q1024,c5_shard,clustered_k100,dups17 --detail-out $OUT/d_${rega}_$r.json > $OUT/l_${rega}_$r.json 2> /dev/null
python - <<PY
import json
d=json.load(open("$OUT/l_${rega}_$r.json"))
print("rega", $rega, $r, [(s["name"], s["ms_per_step"], s["kernel_avg_ms"], s.get("blocking_ms"), s["ck"]) for s in d["secondary"]])
PY
done; done
