#!/usr/bin/env python3
"""Do several single-query scans of the same store, running at the same time on different streams, share HBM traffic
(the second .. fourth hit in L2 / the memory-side cache while they walk the same window)? Throughput with the scans chained
(one at a time) against 2 / 4 streams without the chain. Decides whether the exact-path fallbacks of a batch should be
issued concurrently."""
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
for rows, dims in ((1_000_000, 384), (400_000, 512), (4_000_000, 384)):
    eng = wax.HIPVectorEngine(dimensions=dims)
    eng.reserve(rows)
    for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    q = bench.unit_queries(64, dims)
    for chain, streams, depth in ((1, 2, 4), (0, 2, 4), (0, 4, 8), (0, 4, 16), (1, 2, 4)):
        eng.setTuning("scan_chain", chain)
        eng.setTuning("streams", streams)
        eng.setTuning("slots", depth)
        def run(n):
            pend = []
            for i in range(n):
                if len(pend) >= depth:
                    eng.collect(pend.pop(0), 10)
                pend.append(eng.submit(q[i % 64], 10))
            while pend:
                eng.collect(pend.pop(0), 10)
        run(40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        run(n)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps({"rows": rows, "dims": dims, "chain": chain, "streams": streams, "depth": depth, "qps": round(n / el, 1),
                          "us_per_query": round(el / n * 1e6, 1), "eff_TBps": round(rows * dims * 4 * n / el / 1e12, 2)}), flush=True)
    eng.close()
