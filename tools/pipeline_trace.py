#!/usr/bin/env python3
"""Pipelined device-resident batches (wax_hip_search_batch_submit_device / _collect_device, 2 in flight) for a kernel
trace: `rocprofv3 --kernel-trace -- python tools/pipeline_trace.py`; tools/trace_timeline.py prints the overlap."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import wax_amd as wax  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dims", type=int, default=384)
ap.add_argument("--nq", type=int, default=256)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--timed", type=int, default=1)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = wax.HIPVectorEngine(dimensions=args.dims)
eng.reserve(args.rows)
for r0, x in bench.device_rows(torch, 0, args.rows, args.dims, dev):
    eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
for kv in args.tune:
    k, v = kv.split("=", 1)
    eng.setTuning(k, int(v))
eng.setTuning("time_kernels", args.timed)
k = 10
dq = torch.from_numpy(bench.unit_queries(args.nq, args.dims)).to(dev)
outs = [torch.empty((args.nq, k, 2), dtype=torch.int64, device=dev) for _ in range(args.depth)]
st = torch.cuda.current_stream(dev).cuda_stream
eng.searchBatchHitsDevice(dq.data_ptr(), args.nq, k, outs[0].data_ptr(), k, st)


def run(n):
    tickets = []
    for i in range(n):
        if len(tickets) == args.depth:
            eng.searchBatchCollectDevice(tickets.pop(0))
        tickets.append(eng.searchBatchSubmitDevice(dq.data_ptr(), args.nq, k, outs[i % args.depth].data_ptr(), k, st))
    for t in tickets:
        eng.searchBatchCollectDevice(t)


run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(args.steps)
torch.cuda.synchronize()
print(f"ms_per_batch {(time.perf_counter() - t0) / args.steps * 1e3:.4f}")
