#!/usr/bin/env python3
"""Copy the summaries of a tools/sessions/final.sh run (gpurun_out/<tag>/) into profiles/<round>/ under stable names and refresh
profiles/latest_traffic.json (the counters-only FETCH_SIZE passes bench.py replays as `roofline.traffic`).
    python tools/collect_profiles.py gpurun_out/r05_final profiles/r05"""
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
rnd = os.path.basename(dst.rstrip("/"))


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


cp("bench_n1.json", "z_bench_n1_final.json")
cp("bench_n1_detail.json", "z_bench_n1_final_detail.json")
cp("bench_driver_style.json", "z_bench_n1_driver_style_steps20.json")
cp("headline_chained_kernel_stats.csv", "z_headline_chained_kernel_stats.csv")
cp("headline_chained.out", "z_bench_headline_chained_under_rocprof.json")
cp("default_cmd_kernel_stats.csv", "z_default_cmd_kernel_stats.csv")
cp("default_cmd.out", "z_bench_n1_under_rocprof.json")
cp("scale_rehearsal.txt", "z_scale_matrix_rehearsal_one_gpu.txt")
for a, b in (("sq_768", "z_pmc_sq_counters_768_rq_final_build.json"), ("sq2_768", "z_pmc_sq2_counters_768_rq_final_build.json"),
             ("sq_384", "z_pmc_sq_counters_384_rq_final_build.json"), ("sq_768_wide", "z_pmc_sq_counters_768_wide_final_build.json"),
             ("sq2_768_wide", "z_pmc_sq2_counters_768_wide_final_build.json")):
    cp(a + ".json", b)
for a in ("one_process_2_shards.json", "one_process_n1.json", "pytest_gpu.log", "fuzz.txt", "latency_c.jsonl", "smoke.log", "phase_budget.jsonl", "phase_table.txt", "tail_pool_ab.jsonl",
          "fanout_ABC.jsonl", "gemm_s10m_k300_kernel_stats.csv", "gemm_s1m_k1000_kernel_stats.csv"):
    cp(a, "z_" + a)

# the fuzz log: the first trials and the summary line (every trial is one line: 300 KB per run)
fz = os.path.join(dst, "z_fuzz.txt")
if os.path.exists(fz):
    lines = open(fz).read().splitlines()
    if len(lines) > 60:
        open(fz, "w").write("\n".join(lines[:40] + [f"... {len(lines) - 41} more trials, one line each ..."] + lines[-1:]) + "\n")

# one rocprofv3 row per GEMM workload
rows = []
for w, label in (("b1m_q256", "config 3: 1M x 384, Q = 256"), ("b1m_q1024", "1M x 384, Q = 1024"), ("c5_shard", "config 5 per-GPU part: 1.25M x 768, Q = 1024"),
                 ("c5_full", "config 5 whole: 10M x 768, Q = 1024"), ("clustered_k100", "clustered corpus, 1M x 384, Q = 256, k = 100")):
    path = os.path.join(src, f"gemm_{w}_kernel_stats.csv")
    if not os.path.exists(path):
        continue
    det = {}
    try:
        det = json.load(open(os.path.join(src, f"gemm_{w}_detail.json")))
    except (OSError, ValueError):
        pass
    sec = next((x for x in det.get("secondary", []) if x.get("name") == w), {})
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if ("batch_gemm" in n or "batch_retry" in n or "batch_finish" in n or "pick_tau" in n or "batch_prep" in n) and "scan_kernel" not in n:
            rows.append({"workload": label, "kernel": n.split("(")[0].replace("void ", ""), "calls": r["Calls"], "avg_ns": r["AverageNs"], "min_ns": r["MinNs"],
                         "max_ns": r["MaxNs"], "bench_hip_event_gemm_avg_ms": sec.get("roofline", {}).get("kernel_avg_ms", ""),
                         "bench_ms_per_step": sec.get("ms_per_step", "")})
if rows:
    with open(os.path.join(dst, "z_gemm_rows_per_workload.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def fetch(name, pred):
    try:
        d = json.load(open(os.path.join(src, name + ".json")))
    except (OSError, ValueError):
        return None
    best = None
    for k, v in d.items():
        if "FETCH_SIZE" in v and pred(k):
            ent = {"kernel": k.replace("void ", ""), "launches": v["FETCH_SIZE"]["launches"], "hbm_bytes_per_launch": v["FETCH_SIZE"]["hbm_bytes_per_launch_corrected"],
                   "fetch_size_kb_mean": v["FETCH_SIZE"]["mean"]}
            if best is None or ent["hbm_bytes_per_launch"] > best["hbm_bytes_per_launch"]:
                best = ent
    return best


tpath = os.path.join(os.path.dirname(dst.rstrip("/")), "latest_traffic.json")
t = json.load(open(tpath))
out = {"fetch": {}}
h = fetch("fetch_headline", lambda k: "scan_kernel" in k)
if h:
    t.update({"source": f"profiles/{rnd}/z_pmc_fetch_size.json (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary; counters-only pass; timed region + calibration pass)",
              "rows_per_launch": 10000000, "dims": 384, "fetch_size_kb_mean": h["fetch_size_kb_mean"], "launches": h["launches"], "hbm_bytes_per_launch": h["hbm_bytes_per_launch"]})
    out["fetch"]["headline 10M x 384 scan"] = h
b = t.setdefault("batched", {}).setdefault("configs", {})
for name, key in (("fetch_768_shard", "1250000x768xq1024"), ("fetch_768_full", "10000000x768xq1024")):
    e = fetch(name, lambda k: "batch_gemm_rq" in k or "batch_gemm_wide" in k)
    if e:
        b[key] = {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "launches": e["launches"], "kernel": e["kernel"]}
        out["fetch"][key] = e
e = fetch("fetch_384_q256", lambda k: ("batch_gemm_rq" in k or "batch_gemm_rega" in k))
if e:
    out["fetch"]["1M x 384 (Q = 256 and 1 024 launches averaged)"] = e
t["batched"]["source"] = (f"profiles/{rnd}/z_pmc_fetch_size.json (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/batch_bench.py --dims D --rows N --nq Q --reps 2; "
                          "counters-only passes; the instantiation of the filtering GEMM with the largest mean)")
json.dump(t, open(tpath, "w"), indent=1)
out["correction"] = t.get("correction")
json.dump(out, open(os.path.join(dst, "z_pmc_fetch_size.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"]) for k, v in out["fetch"].items()}))
