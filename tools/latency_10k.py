#!/usr/bin/env python3
"""Warm single-query latency at 10 000 x 384, top-24 (run under `rocprofv3 --kernel-trace` to see the per-query launch
timeline: scan 6 us, merge 16 us, the rest is host time between the merge and the next submit)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wax_amd as wax
eng = wax.HIPVectorEngine(dimensions=384)
rows = np.random.default_rng(0).standard_normal((10000, 384)).astype(np.float32)
eng.addBatch(np.arange(10000, dtype=np.uint64), rows)
q = rows[5]
for _ in range(20):
    eng.searchArrays(q, 24)
t0 = time.perf_counter()
for _ in range(200):
    eng.searchArrays(q, 24)
print("latency_us", (time.perf_counter() - t0) / 200 * 1e6)
