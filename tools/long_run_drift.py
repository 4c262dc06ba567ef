#!/usr/bin/env python3
"""Does the pipelined single-query path slow down over a long run? 1M x 384, depth-4 submit/collect, per-250-query
averages, with and without per-kernel timing events."""
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
rows, dims = int(os.environ.get("WAX_ROWS", "1000000")), 384
eng = wax.HIPVectorEngine(dimensions=dims)
eng.reserve(rows)
for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
    eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
eng.setTuning("streams", 2)
eng.setTuning("slots", 4)
nq_distinct = int(os.environ.get("WAX_DISTINCT", "64"))
q = bench.unit_queries(nq_distinct, dims)
if os.environ.get("WAX_PREWARM"):
    st = torch.cuda.Stream(device=dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
    t0 = time.perf_counter()
    for i in range(int(os.environ["WAX_PREWARM"])):
        evs[i % 64].record(st)
    torch.cuda.synchronize()
    print(f"prewarm: {int(os.environ['WAX_PREWARM'])} event records in {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
if os.environ.get("WAX_NOGC"):
    import gc
    gc.collect()
    gc.disable()
    print("python gc disabled", flush=True)
W = int(os.environ.get("WAX_WINDOW", "500"))
for timed in (0, 1, 0):
    eng.setTuning("time_kernels", timed)
    pend = []
    out = []
    t0 = time.perf_counter()
    for i in range(int(os.environ.get('WAX_QUERIES', '3000'))):
        if len(pend) >= 4:
            eng.collect(pend.pop(0), 10)
        pend.append(eng.submit(q[i % nq_distinct], 10))
        if (i + 1) % W == 0:
            t1 = time.perf_counter()
            out.append(round((t1 - t0) / W * 1e6, 1))
            t0 = t1
    while pend:
        eng.collect(pend.pop(0), 10)
    print(f"time_kernels={timed}: us/query per 500-query window: {out}", flush=True)
