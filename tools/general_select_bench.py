#!/usr/bin/env python3
"""Single-query scans by top_k across the fused / general-selection boundary (k <= 192 fused in the scan kernel, beyond it the
distance pass + radix select of DESIGN §4.2): pipelined (depth 4, two streams — the bench's shape) and blocking ms per query.

    python tools/general_select_bench.py --rows 10000000 --dims 384 --topk 10 192 195 300 1000
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
    ap.add_argument("--topk", type=int, nargs="+", default=[10, 100, 192, 195, 300, 1000])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = bench._load_engine(torch, dev, args.rows, args.dims)
    eng.setTuning("streams", args.streams)
    eng.setTuning("slots", max(args.depth, 2))
    for kv in args.tune:
        k_, v_ = kv.split("=")
        eng.setTuning(k_, int(v_))
    qs = bench.unit_queries(args.steps + 8, args.dims)
    import hashlib
    for k in args.topk:
        bench.run_pipelined(lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), qs[:8], args.depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = bench.run_pipelined(lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), qs[8:], args.depth)
        torch.cuda.synchronize()
        pip = (time.perf_counter() - t0) / args.steps
        t1 = time.perf_counter()
        nb = min(args.steps, 20)
        for q in qs[8:8 + nb]:
            eng.searchArrays(q, k)
        blk = (time.perf_counter() - t1) / nb
        chk = hashlib.sha256(np.asarray(last[0], dtype=np.uint64).tobytes() + np.asarray(last[1], dtype=np.float32).tobytes()).hexdigest()[:16]
        line = json.dumps({"rows": args.rows, "dims": args.dims, "top_k": k, "tune": args.tune, "streams": args.streams, "depth": args.depth, "ms_pipelined": pip * 1e3, "ms_blocking": blk * 1e3,
                           "frac_of_8TBps_pipelined": args.rows * args.dims * 4 / pip / 8e12, "returned": int(len(last[0])), "checksum": chk})
        print(line, flush=True)
        if args.out:
            with open(args.out, "a") as f:
                f.write(line + "\n")
    eng.close()


if __name__ == "__main__":
    main()
