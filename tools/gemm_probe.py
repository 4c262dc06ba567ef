#!/usr/bin/env python3
"""Timing experiments on the register-resident batched GEMM (results are NOT checked): which of {corpus loads,
MFMAs, selection, barrier} bounds a tile. Run under `rocprofv3 --kernel-trace`; the batch_gemm_rega_kernel launches
appear in the order printed here (3 batches per configuration, 3 slab launches per batch at growth 8, the last of
each batch being the ~850 K-row slab). Debug bits (GemmArgs::debug): 1 no corpus loads, 2 no MFMAs, 4 no per-tile
barrier (only with 1), 8 no selection, 16 both waves of a SIMD in the same MFMA/select order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import wax_amd as wax  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rows, dims = 1_000_000, 384
nq = int(os.environ.get("WAX_PROBE_NQ", "256"))
eng = wax.HIPVectorEngine(dimensions=dims)
eng.reserve(rows)
for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
    eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
q = bench.unit_queries(nq, dims)
eng.searchBatch(q, 10)
CONFIGS = [(1, 0), (1, 8), (1, 9), (1, 13), (1, 16), (1, 24), (1, 10), (1, 11), (2, 0), (2, 8), (2, 10)]
for rega, dbg in CONFIGS:
    eng.setTuning("batch_rega", rega)
    eng.setTuning("batch_debug", dbg)
    for _ in range(3):
        eng.searchBatch(q, 10)
    print(f"rega={rega} debug={dbg}", flush=True)
