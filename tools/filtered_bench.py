#!/usr/bin/env python3
"""Latency of wax_hip_search_filtered (allow-list pre-filter on the device) vs the unfiltered scan, 1M x 384."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import wax_amd as wax  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rows, dims = 1_000_000, 384
eng = wax.HIPVectorEngine(dimensions=dims)
eng.reserve(rows)
for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
    eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
q = bench.unit_queries(4, dims)
rng = np.random.default_rng(0)
out = {"rows": rows, "dims": dims, "topk": 10}
eng.searchArrays(q[0], 10)
t0 = time.perf_counter()
for i in range(50):
    eng.searchArrays(q[i % 4], 10)
out["unfiltered_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
for n_allow in (100, 10_000, 100_000, 1_000_000):
    allow = rng.permutation(rows)[:n_allow].astype(np.uint64)
    for label, device_min in (("host_probe", -1), ("device_probe", 0)):   # id -> row on the host / by the table in HBM
        eng.setTuning("filter_device_min", device_min)
        eng.searchFiltered(q[0], 10, frameIds=allow)
        t0 = time.perf_counter()
        reps = 20 if n_allow <= 100_000 else 5
        for i in range(reps):
            ids, scores = eng.searchFiltered(q[i % 4], 10, frameIds=allow)
        out[f"allow_{n_allow}_{label}_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        assert len(ids) == 10 and np.all(np.isin(ids, allow))
print(json.dumps(out))
