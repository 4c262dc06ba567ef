#!/usr/bin/env python3
"""Randomised check of the short selection / short merge against the long path / wave-list merge: same engine, same query,
"select_short" 1 then 0 — ids and scores must be equal bit for bit, whatever the store looks like (iid, sorted by similarity, clustered,
runs of duplicates, scaled rows, NaN / zero rows), for k on both sides of every boundary (64 / 192 / viability).

    python tools/fuzz_select.py --seconds 120 --seed 1
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
import torch  # noqa: E402

import wax_amd as wax  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
t_end = time.time() + args.seconds
trials = checked = shorts = fails = 0
while time.time() < t_end:
    dims = int(rng.choice([128, 384, 384, 768, 100, 64]))
    n = int(rng.choice([3000, 20_000, 60_000, 150_000, 400_000, 1_100_000]))
    metric = int(rng.choice([0, 0, 1, 2]))
    g = torch.Generator(device=dev)
    g.manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, dims), generator=g, device=dev, dtype=torch.float32)
    q = torch.randn((3, dims), generator=g, device=dev)
    layout = int(rng.integers(0, 6))
    if layout in (0, 1, 2, 3):
        x = torch.nn.functional.normalize(x, dim=1)
    if layout == 1:                                   # sorted by similarity to the first query: the best rows are the first rows
        order = torch.argsort(x @ torch.nn.functional.normalize(q[0], dim=0), descending=True)
        x = x[order].contiguous()
    elif layout == 2:                                 # clustered: 12 centres, queries near centres
        c = torch.randn((12, dims), generator=g, device=dev)
        x = torch.nn.functional.normalize(c[torch.randint(0, 12, (n,), generator=g, device=dev)] + 0.25 * x * dims ** 0.5 / dims ** 0.5, dim=1)
        q[:2] = c[:2] + 0.05 * q[:2]
    elif layout == 3:                                 # runs of exact duplicates (equal distances: the row decides)
        reps = int(rng.choice([300, 700, 5000]))
        x[n // 3:n // 3 + reps] = x[5]
        x[: reps // 2] = x[9]
        q[0] = x[5] + 0.01 * q[0]
    elif layout == 4:                                 # scaled rows (dot / l2 orders differ from cosine), a few zero rows
        x = x * (0.25 + 2.0 * torch.rand((n, 1), generator=g, device=dev))
        x[torch.randint(0, n, (17,), generator=g, device=dev)] = 0.0
    else:                                             # periodic: every 4 040th chunk of 8 rows is a near-copy of the query (one workgroup's share)
        step = int(rng.choice([505, 2020, 4040, 512])) * 8
        idx = torch.arange(0, n, step, device=dev)
        idx = (idx[:, None] + torch.arange(8, device=dev)[None, :]).reshape(-1)
        idx = idx[idx < n]
        x = torch.nn.functional.normalize(x, dim=1)
        x[idx] = torch.nn.functional.normalize(q[0][None, :] + 0.02 * torch.randn((len(idx), dims), generator=g, device=dev), dim=1)
    eng = wax.HIPVectorEngine(metric=wax.VectorMetric(metric), dimensions=dims)
    eng.reserve(n)
    eng.addBatchDevice(np.arange(n, dtype=np.uint64) * 5 + 2, x.contiguous())
    if rng.random() < 0.3:
        eng.setRowBase(int(rng.integers(1, 1 << 22)))
    qh = q.cpu().numpy()
    if rng.random() < 0.05:
        qh[2, 0] = np.nan
    ks = [int(v) for v in rng.choice([64, 65, 100, 192, 193, 250, 300, 500, 1000, 2000, 4096, 7000, 10000], size=4, replace=False)]
    s0, f0 = eng.getTuning("short_selects"), eng.getTuning("short_select_failures")
    for k in ks:
        for qi in qh:
            eng.setTuning("select_short", 1)
            a = eng.searchArrays(qi, k)
            eng.setTuning("select_short", 0)
            b = eng.searchArrays(qi, k)
            ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
            if not ok:
                print(json.dumps({"FAIL": True, "n": n, "dims": dims, "metric": metric, "layout": layout, "k": k}), flush=True)
                sys.exit(1)
            checked += 1
    shorts += eng.getTuning("short_selects") - s0
    fails += eng.getTuning("short_select_failures") - f0
    trials += 1
    print(json.dumps({"n": n, "dims": dims, "metric": metric, "layout": layout, "ks": ks, "short": eng.getTuning("short_selects") - s0,
                      "failed_over": eng.getTuning("short_select_failures") - f0}), flush=True)
    eng.close()
    del x, q
print(json.dumps({"trials": trials, "answers_checked": checked, "short_selections": shorts, "failed_over_to_the_long_path": fails, "ok": True}))
