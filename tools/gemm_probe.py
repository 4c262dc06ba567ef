#!/usr/bin/env python3
"""Timing experiments on the batched GEMM (results are NOT checked): which of {corpus loads, MFMAs,
epilogue} bounds a tile. Run under rocprofv3 --kernel-trace to read per-launch durations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import wax_amd as wax
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rows, dims = 1_000_000, 384
eng = wax.HIPVectorEngine(dimensions=dims)
eng.reserve(rows)
for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
    eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
q = bench.unit_queries(256, dims)
eng.searchBatch(q, 10)
for rega in (1, 0):
    for dbg in (0, 8, 9, 13, 2, 3):
        if rega == 0 and dbg:
            continue
        eng.setTuning("batch_rega", rega)
        eng.setTuning("batch_debug", dbg)
        eng.searchBatch(q, 10)
        t0 = time.perf_counter()
        for _ in range(2):
            eng.searchBatch(q[:256], 10)
        print(f"rega={rega} debug={dbg}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/batch", flush=True)
