#!/usr/bin/env python3
"""One fresh process = one line: where the store slab of a rows x dims engine landed and how fast it scans (the 1M-row
point runs in one of two modes, 214 or 222 us per query, decided per process). Run it N times from a shell loop; WAX_PROBE_PAD_KB
allocates (and keeps) a pad of that many KB before the engine so that the slab's base address moves between processes on purpose.

    for i in $(seq 0 15); do WAX_PROBE_PAD_KB=$((i * 832)) python tools/bimodal_probe.py --rows 1000000; done
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
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pad_kb = int(os.environ.get("WAX_PROBE_PAD_KB", "0"))
    pad = torch.empty(pad_kb * 1024, dtype=torch.uint8, device=dev) if pad_kb else None
    eng = bench._load_engine(torch, dev, args.rows, args.dims)
    eng.setTuning("streams", 2)
    eng.setTuning("slots", 4)
    for kv in args.tune:
        k_, v_ = kv.split("=")
        eng.setTuning(k_, int(v_))
    qs = bench.unit_queries(args.steps + 20, args.dims)
    k = 10
    bench.run_pipelined(lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), qs[:20], 4)
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_pipelined(lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), qs[20:], 4)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / args.steps * 1e6)
    kern = eng.timeScanKernel(qs[0], k, 200) * 1e3      # back-to-back launches of the scan kernel alone, us each
    ptr = eng.getTuning("store_ptr")
    line = json.dumps({"rows": args.rows, "dims": args.dims, "pad_kb": pad_kb, "store_ptr": hex(ptr), "ptr_mod_2MiB": ptr % (2 << 20), "ptr_mod_1GiB_MiB": (ptr % (1 << 30)) >> 20,
                       "scan_grid": eng.getTuning("scan_grid"), "us_per_query_pipelined": [round(x, 2) for x in res], "us_scan_kernel_back_to_back": round(kern, 2),
                       "tune": args.tune, "pid": os.getpid()})
    print(line, flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    del pad
    eng.close()


if __name__ == "__main__":
    main()
