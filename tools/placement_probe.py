#!/usr/bin/env python3
"""Is the 1M-row bimodality (214 or 222 us per query, decided per process) a property of the PROCESS or of where one
allocation landed? K engines with the same rows x dims store in ONE process, the scan kernel timed back to back on each (HIP events,
200 launches) and the library's stream-read microbenchmark beside it. Slabs that differ inside one process = physical placement.

    python tools/placement_probe.py --engines 6 --rows 1000000
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engines", type=int, default=6)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    engs = [bench._load_engine(torch, dev, args.rows, args.dims) for _ in range(args.engines)]
    q = bench.unit_queries(4, args.dims)[0]
    rows = []
    for rep in range(2):
        for i, e in enumerate(engs):
            us = e.timeScanKernel(q, 10, 200) * 1e3
            sr = e.timeStreamRead(20) * 1e3
            rows.append({"rep": rep, "engine": i, "store_ptr": hex(e.getTuning("store_ptr")), "us_scan_kernel": round(us, 2), "us_stream_read": round(sr, 2)})
    line = json.dumps({"rows": args.rows, "dims": args.dims, "pid": os.getpid(), "slabs": rows})
    print(line, flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
