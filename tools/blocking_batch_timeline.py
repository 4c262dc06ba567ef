#!/usr/bin/env python3
"""One blocking batched call (config 3: 1M x 384, 256 queries, queries and hits in HBM) at a time: host wall time per call, to be laid
beside the kernel timeline of the same calls (rocprofv3 --kernel-trace of this process + tools/trace_timeline.py): where the time
between the filtering GEMM's duration and the call's duration goes.  --tune KEY=VALUE as in bench.py."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dims", type=int, default=384)
    p.add_argument("--nq", type=int, default=256)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--calls", type=int, default=200)
    p.add_argument("--tune", action="append", default=[])
    p.add_argument("--pinned", action="store_true", help="with --host: the queries come from pinned host memory")
    p.add_argument("--host", action="store_true", help="host-pointer form (searchBatchHits: queries uploaded, hits downloaded by the library)")
    a = p.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    eng = bench._load_engine(torch, dev, a.rows, a.dims)
    for kv in a.tune:
        k, v = kv.split("=", 1)
        eng.setTuning(k, int(v))
    dq = bench.batch_queries(torch, dev, a.nq, a.dims, "gaussian")
    out = torch.empty((a.nq, a.k, 2), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(10):
        eng.searchBatchHitsDevice(dq.data_ptr(), a.nq, a.k, out.data_ptr(), a.k, st)
    torch.cuda.synchronize()
    per = []
    hq = dq.cpu().numpy() if a.host else None
    if a.host and a.pinned:
        keep = dq.cpu().pin_memory()
        hq = keep.numpy()
    if a.host:
        for _ in range(5):
            eng.searchBatchHits(hq, a.k)
    for _ in range(a.calls):
        t0 = time.perf_counter()
        if a.host:
            eng.searchBatchHits(hq, a.k)
        else:
            eng.searchBatchHitsDevice(dq.data_ptr(), a.nq, a.k, out.data_ptr(), a.k, st)
        per.append((time.perf_counter() - t0) * 1e6)
    per = np.array(per)
    worst = np.argsort(per)[::-1][:5]
    print("slowest calls (index: us): " + ", ".join(f"{int(i)}: {per[i]:.0f}" for i in worst), flush=True)
    print(f"blocking {'host-pointer' if a.host else 'device-resident'} call, {a.rows} x {a.dims}, {a.nq} queries, tunes {a.tune}: mean {per.mean():.1f} us  median {np.median(per):.1f}  p10 {np.percentile(per, 10):.1f}  min {per.min():.1f}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
