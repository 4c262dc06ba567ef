set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_ab2
for r in 1 2; do for name in ${WAX_AB_LIBS:-base lanes}; do
WAX_HIP_LIB=$R/wax_amd/lib/exp/libwaxhip_$name.so timeout 600 python bench.py --gpus 1 --rows 1000000 --steps 60 --warmup 10 --no-cpu-baseline --traffic off --secondary ${WAX_AB_SEC:-b1m_q256,b1m_q1024,c5_shard,clustered_k100,dups17} --detail-out gpurun_out/r06_ab2/d_${name}_$r.json > gpurun_out/r06_ab2/l_${name}_$r.json 2> /dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r06_ab2/l_${name}_$r.json"))
print("$name", $r, [(s["name"], s["ms_per_step"], s["kernel_avg_ms"], s.get("blocking_ms"), s["ck"]) for s in d["secondary"]])
PY
done; done
