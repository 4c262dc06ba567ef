#!/usr/bin/env python3
"""On-device tuning sweep (run on the MI355X box via gpurun): times every scan-kernel variant x
grid size with HIP events, the streaming-read microbenchmark (node roofline denominator), top-k
sensitivity and end-to-end queries/s at several pipeline depths. Writes gpurun_out/sweep_<tag>.json.

    python tools/sweep.py --tag r01 [--rows 10000000] [--dims 384] [--quick]
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

import bench  # noqa: E402  (device_rows / unit_queries)


def build_engine(torch, wax, rows, dims, dev):
    eng = wax.HIPVectorEngine(dimensions=dims)
    eng.reserve(rows)
    for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    torch.cuda.synchronize()
    return eng


def gbps(rows, dims, ms):
    return rows * dims * 4 / (ms * 1e-3) / 1e9


def e2e_qps(eng, queries, k, depth, n):
    pend = []
    t0 = time.perf_counter()
    for i in range(n):
        if len(pend) >= depth:
            eng.collect(pend.pop(0), k)
        pend.append(eng.submit(queries[i % len(queries)], k))
    while pend:
        eng.collect(pend.pop(0), k)
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()

    import torch
    import wax_amd as wax
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {"rows": args.rows, "dims": args.dims, "device": torch.cuda.get_device_name(0), "results": []}
    res = out["results"]

    def rec(**kw):
        res.append(kw)
        print(json.dumps(kw), flush=True)

    q = bench.unit_queries(64, args.dims)
    eng = build_engine(torch, wax, args.rows, args.dims, dev)
    nvar = eng.getTuning("variant_count")
    grids = [512, 1024, 2048, 4096, 8192] if not args.quick else [2048]

    # 1. streaming read roofline denominator
    for nt in (1, 0):
        for grid in grids:
            eng.setTuning("stream_nt", nt)
            eng.setTuning("grid_blocks", grid)
            ms = eng.timeStreamRead(args.iters)
            rec(kind="stream_read", nt=nt, grid=grid, ms=ms, gbps=gbps(args.rows, args.dims, ms))

    # 2. scan kernel variants x grid
    for variant in range(nvar):
        for grid in grids:
            eng.setTuning("variant", variant)
            eng.setTuning("grid_blocks", grid)
            ms = eng.timeScanKernel(q[0], 10, args.iters)
            rec(kind="scan", variant=variant, grid=grid, k=10, ms=ms, gbps=gbps(args.rows, args.dims, ms))
    best = max((r for r in res if r["kind"] == "scan"), key=lambda r: r["gbps"])
    rec(kind="best_scan", **{k: best[k] for k in ("variant", "grid", "ms", "gbps")})
    eng.setTuning("variant", best["variant"])
    eng.setTuning("grid_blocks", best["grid"])

    # 3. top-k sensitivity of the fused kernel
    for k in (1, 10, 30, 64, 65, 100, 192):
        ms = eng.timeScanKernel(q[1], k, args.iters)
        rec(kind="scan_k", k=k, ms=ms, gbps=gbps(args.rows, args.dims, ms))

    # 4. end-to-end queries/s through the C ABI (submit/collect), pipeline depth x streams
    for streams in (1, 2):
        for depth in (1, 2, 4):
            eng.setTuning("streams", streams)
            eng.setTuning("slots", max(depth, 1))
            e2e_qps(eng, q, 10, depth, 10)
            qps = e2e_qps(eng, q, 10, depth, 100 if args.rows > 2_000_000 else 400)
            rec(kind="e2e", streams=streams, depth=depth, qps=qps, gbps=qps * args.rows * args.dims * 4 / 1e9)
    # general path (k = 1000) end to end
    eng.setTuning("streams", 1)
    t0 = time.perf_counter()
    for i in range(10):
        eng.searchArrays(q[i], 1000)
    rec(kind="e2e_general_k1000", qps=10 / (time.perf_counter() - t0))
    del eng
    torch.cuda.empty_cache()

    # 5. smaller corpora (cache-resident / launch-bound regimes) and D=768
    if not args.quick:
        for rows, dims in ((1_000_000, 384), (10_000, 384), (5_000_000, 768), (4_000_000, 1536), (8_000_000, 128)):
            e = build_engine(torch, wax, rows, dims, dev)
            qq = bench.unit_queries(8, dims)
            for variant in range(e.getTuning("variant_count")):
                for grid in (1024, 2048, 4096):
                    e.setTuning("variant", variant)
                    e.setTuning("grid_blocks", grid)
                    ms = e.timeScanKernel(qq[0], 10, args.iters * (10 if rows <= 1_000_000 else 1))
                    rec(kind="scan_other", rows=rows, dims=dims, variant=variant, grid=grid, ms=ms,
                        gbps=gbps(rows, dims, ms))
            e.setTuning("variant", -1)
            e.setTuning("grid_blocks", 0)
            e.setTuning("slots", 4)
            e.setTuning("streams", 1)
            qps = e2e_qps(e, qq, 10, 4, 400)
            rec(kind="e2e_other", rows=rows, dims=dims, depth=4, qps=qps, gbps=qps * rows * dims * 4 / 1e9)
            lat = []
            for i in range(50):
                t0 = time.perf_counter()
                e.searchArrays(qq[i % 8], 10)
                lat.append(time.perf_counter() - t0)
            rec(kind="latency_other", rows=rows, dims=dims, p50_us=float(np.median(lat) * 1e6), min_us=float(min(lat) * 1e6))
            del e
            torch.cuda.empty_cache()

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"sweep_{args.tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
