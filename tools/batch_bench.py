#!/usr/bin/env python3
"""Batched-query benchmark (BASELINE configs 3/5 shapes on one GPU): Q x D^T bf16 MFMA GEMM +
per-query select + exact f32 re-score. Prints one JSON line per configuration.

    python tools/batch_bench.py [--rows 1000000] [--dims 384] [--nq 256] [--topk 10] [--reps 5]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--nq", type=int, nargs="+", default=[256])
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--slab-mb", type=int, nargs="+", default=[64])
    args = ap.parse_args()
    import torch
    import wax_amd as wax
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = wax.HIPVectorEngine(dimensions=args.dims)
    eng.reserve(args.rows)
    for r0, x in bench.device_rows(torch, 0, args.rows, args.dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    for nq in args.nq:
        q = bench.unit_queries(nq, args.dims)
        for slab in args.slab_mb:
            eng.setTuning("batch_slab_mb", slab)
            eng.searchBatch(q, args.topk)  # warm-up (+ mirror build on the first call)
            fb0 = eng.getTuning("batch_fallbacks")
            t0 = time.perf_counter()
            for _ in range(args.reps):
                ids, scores, counts = eng.searchBatch(q, args.topk)
            dt = (time.perf_counter() - t0) / args.reps
            print(json.dumps({"rows": args.rows, "dims": args.dims, "nq": nq, "topk": args.topk, "slab_mb": slab,
                              "ms_per_batch": dt * 1e3, "qps": nq / dt,
                              "tflops_bf16": 2.0 * nq * args.rows * args.dims / dt / 1e12,
                              "fallbacks": eng.getTuning("batch_fallbacks") - fb0}), flush=True)


if __name__ == "__main__":
    main()
