#!/usr/bin/env python3
"""Host-side cost per query of the sharded pipeline (ShardedSearcher at world=1 on a tiny corpus, so the
GPU work is negligible): what the Python/torch/ctypes layer must stay under for 8-GPU scaling
(per-GPU scan of 1.25M x 384 rows is ~0.28 ms)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import wax_amd as wax
    from wax_amd import sharded
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for rows in (10_000, 1_250_000):
        eng = wax.HIPVectorEngine(dimensions=384)
        eng.reserve(rows)
        for r0, x in bench.device_rows(torch, 0, rows, 384, dev):
            eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
        q = bench.unit_queries(64, 384)
        for depth in (1, 4):
            s = sharded.ShardedSearcher(eng, 0, 1, 10, depth=depth, n_streams=2)
            for it in range(2):
                n = 2000 if rows <= 10_000 else 500
                t0 = time.perf_counter()
                for i in range(n):
                    if len(s.inflight) >= depth:
                        s.collect()
                    s.submit(q[i % 64])
                while s.inflight:
                    s.collect()
                dt = (time.perf_counter() - t0) / n
            print(f"rows={rows} depth={depth}: {dt * 1e6:.1f} us/query ({1 / dt:.0f} q/s) through ShardedSearcher (world=1)")
        # the engine's own submit/collect for comparison
        eng.setTuning("streams", 2)
        eng.setTuning("slots", 4)
        pend = []
        n = 2000 if rows <= 10_000 else 500
        t0 = time.perf_counter()
        for i in range(n):
            if len(pend) >= 4:
                eng.collect(pend.pop(0), 10)
            pend.append(eng.submit(q[i % 64], 10))
        while pend:
            eng.collect(pend.pop(0), 10)
        dt = (time.perf_counter() - t0) / n
        print(f"rows={rows} engine submit/collect depth 4: {dt * 1e6:.1f} us/query ({1 / dt:.0f} q/s)")
        del eng


if __name__ == "__main__":
    main()
