#!/usr/bin/env python3
"""Why does the one-pass pipeline lose below 64 units of 64 rows?  Per store size: ms per blocking 256-query batch and, per batch, the
certificate fallbacks, full retries (device / host) and shared exact passes, with the floor at 32 units and at the default."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import gc
    import torch
    import wax_amd as wax
    dev = torch.device("cuda", 0)
    gc.collect(); gc.freeze(); gc.disable()      # (a generation-2 pass with torch imported is a 40 ms call inside some 30-batch loop)
    keys = ("onepass_queries", "batch_fallbacks", "batch_retries", "batch_inline_retries", "batch_multi_passes")
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 3000, 4096, 8192]
    floors = (32,) if len(sys.argv) > 1 else (32, 1024)
    for n in sizes:
        for floor in floors:
            for k in ((10,) if len(sys.argv) > 1 else (10, 100)):
                eng = wax.HIPVectorEngine(dimensions=384)
                for r0, x in bench.device_rows(torch, 0, n, 384, dev):
                    eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
                eng.setTuning("batch_onepass_tiles", floor)
                q = bench.unit_queries(256, 384)
                for _ in range(3):
                    eng.searchBatchHits(q, k)
                c0 = {key: eng.getTuning(key) for key in keys}
                reps = 30
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.searchBatchHits(q, k)
                dt = (time.perf_counter() - t0) / reps * 1e6
                d = {key: (eng.getTuning(key) - c0[key]) / reps for key in keys}
                print(f"rows {n} floor {floor} top_k {k}: {dt:.0f} us per batch; per batch: " + ", ".join(f"{key} {v:.1f}" for key, v in d.items()), flush=True)
                eng.close()


if __name__ == "__main__":
    main()
