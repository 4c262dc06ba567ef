#!/usr/bin/env python3
"""Per-launch scan / filtering-GEMM durations under the two HIP-event modes of the library ("time_kernels" = 1: hipEventRecord in front of
and behind the launch; 2: the pair bound to the dispatch by hipExtLaunchKernel), printed launch by launch so that a rocprofv3 --kernel-trace
of THIS process can be laid beside them (same dispatches, same order):
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python tools/event_modes.py --rows 1250000
Blocking calls only (one kernel at a time, nothing chained), so an interval never holds another kernel."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=1_250_000)
    p.add_argument("--dims", type=int, default=384)
    p.add_argument("--n", type=int, default=40)
    p.add_argument("--nq", type=int, default=0, help="> 0: also time the filtering GEMM of batches of this many queries")
    a = p.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    eng = bench._load_engine(torch, dev, a.rows, a.dims)
    qs = bench.unit_queries(a.n + 4, a.dims)
    for q in qs[:4]:
        eng.searchArrays(q, 10)
    out = {"rows": a.rows, "dims": a.dims}
    for mode in (1, 2, 1, 2):
        eng.setTuning("time_kernels", mode)
        per = []
        for q in qs[4:]:
            eng.setTuning("reset_stats", 1)
            eng.searchArrays(q, 10)
            st = eng.stats()
            per.append(st.scan_kernel_ms_total * 1e3)
        out.setdefault(f"scan_mode{mode}_us", []).append([round(x, 2) for x in per])
        print(f"scan mode {mode}: mean {np.mean(per):.2f} us  median {np.median(per):.2f}  min {np.min(per):.2f}  ({len(per)} launches)", flush=True)
    if a.nq:
        dq = bench.batch_queries(torch, dev, a.nq, a.dims, "gaussian")
        o = torch.empty((a.nq, 10, 2), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        eng.setTuning("time_kernels", 0)
        for _ in range(3):
            eng.searchBatchHitsDevice(dq.data_ptr(), a.nq, 10, o.data_ptr(), 10, stream)
        for mode in (1, 2, 1, 2):
            eng.setTuning("time_kernels", mode)
            per = []
            for _ in range(a.n):
                eng.setTuning("reset_stats", 1)
                eng.searchBatchHitsDevice(dq.data_ptr(), a.nq, 10, o.data_ptr(), 10, stream)
                st = eng.stats()
                per.append(st.batch_gemm_ms_total * 1e3)
            out.setdefault(f"gemm_mode{mode}_us", []).append([round(x, 2) for x in per])
            print(f"gemm nq {a.nq} mode {mode}: mean {np.mean(per):.2f} us  median {np.median(per):.2f}  min {np.min(per):.2f}", flush=True)
    eng.setTuning("time_kernels", 0)
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
