#!/bin/bash
# round 4, GPU session 17: k in 81..128 keeps the fused finish kernel (k' capped at 192) with the device retry behind it — A/B against k' = 2k + 32
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s17
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf -x -p no:cacheprovider --timeout 600 -k "device_side_full_retry or certificate_bound or batch_parity or batched_k or batch_randomised_soak or large_k" > "$OUT/pytest_sel.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_sel.log"
python - "$OUT" > "$OUT/kp_ab.txt" 2> "$OUT/kp_ab.err" <<'PY'
import importlib.util, json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("wax_bench", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
dev = torch.device("cuda", 0)
for rep in (1, 2):
    for fused in (1, 0):
        bench.TUNES[:] = [f"batch_kp_fused={fused}"]
        for corpus, k in (("gaussian", 100), ("clustered", 100), ("gaussian", 128), ("clustered", 81), ("dups", 100)):
            x = bench.secondary_batched(torch, dev, 1_000_000, 384, 256, k, 40, 8, f"{corpus} k={k}", corpus=corpus)
            print(f"kp_fused {fused} {corpus:10s} k {k:3d}: {x['ms_per_step']*1000:7.1f} us pipelined, {x['ms_per_step_blocking_call']*1000:7.1f} us blocking, "
                  f"kernel {x['roofline']['kernel_avg_ms']*1000:6.1f} us, fallbacks/step {x.get('certificate_fallbacks_per_step')}, device retries {x.get('full_retries_on_device')}, "
                  f"host retries {x.get('full_retries')}, ck {x['last_result_checksum'][:10]}", flush=True)
PY
cat "$OUT/kp_ab.txt"; tail -3 "$OUT/kp_ab.err" | cut -c1-300; tail -4 "$OUT/pytest_sel.log"
