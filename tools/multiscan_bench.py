#!/usr/bin/env python3
"""The shared exact pass (multiscan.hip) against one scan per query: milliseconds per call of the device-resident batch
entry point with the MFMA pipelines off ("batch_mode" = 0), i.e. the batch is answered by the exact path alone.
  python tools/multiscan_bench.py [--rows 1000000] [--dims 384 768] [--nq 1 2 8 16 32 48] [--k 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import wax_amd as wax  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dims", type=int, nargs="+", default=[384])
ap.add_argument("--nq", type=int, nargs="+", default=[1, 2, 8, 16, 32, 48])
ap.add_argument("--k", type=int, nargs="+", default=[10])
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
st = torch.cuda.current_stream(dev).cuda_stream
for dims in args.dims:
    eng = wax.HIPVectorEngine(dimensions=dims)
    eng.reserve(args.rows)
    for r0, x in bench.device_rows(torch, 0, args.rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    eng.setTuning("batch_mode", 0)
    scan_ms = eng.timeScanKernel(bench.unit_queries(1, dims)[0], args.k[0], 20)
    for nq, k in [(nq, k) for k in args.k for nq in args.nq]:
        args_k = k
        dq = torch.from_numpy(bench.unit_queries(nq, dims)).to(dev)
        out = torch.empty((nq, args_k, 2), dtype=torch.int64, device=dev)
        res = {}
        for multi in (1, 0):
            eng.setTuning("batch_multi", multi)
            for _ in range(3):
                eng.searchBatchHitsDevice(dq.data_ptr(), nq, args_k, out.data_ptr(), args_k, st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                eng.searchBatchHitsDevice(dq.data_ptr(), nq, args_k, out.data_ptr(), args_k, st)
            torch.cuda.synchronize()
            res[multi] = (time.perf_counter() - t0) / args.reps * 1e3
            res[f"hits{multi}"] = out.cpu().numpy().copy()
        assert np.array_equal(res["hits1"], res["hits0"])
        print(json.dumps({"rows": args.rows, "dims": dims, "nq": nq, "k": args_k, "scan_kernel_ms": scan_ms,
                          "shared_pass_ms_per_call": res[1], "one_scan_per_query_ms_per_call": res[0],
                          "speedup": res[0] / res[1], "passes": -(-nq // 16), "ms_per_pass": res[1] / (-(-nq // 16)),
                          "pass_vs_single_scan": res[1] / (-(-nq // 16)) / scan_ms}), flush=True)
    eng.close()
