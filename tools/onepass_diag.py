#!/usr/bin/env python3
"""Diagnostic for the one-pass batched pipeline: which queries of a batch lose their certificate, and why (emulates the
sampled threshold with torch: survivors per query, per-workgroup segment fill, gaps)."""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--rega", type=int, default=1)
    ap.add_argument("--topk", type=int, default=10)
    args = ap.parse_args()
    import torch
    import wax_amd as wax
    dev = torch.device("cuda", 0)
    n, D, k = args.rows, args.dims, args.topk
    eng = wax.HIPVectorEngine(dimensions=D)
    eng.reserve(n)
    rows = []
    for r0, x in bench.device_rows(torch, 0, n, D, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
        rows.append(x)
    corpus = torch.cat(rows)
    eng.setTuning("batch_rega", args.rega)
    q = bench.unit_queries(args.nq, D)
    eng.searchBatchHits(q, k)
    fb0 = eng.getTuning("batch_fallbacks")
    eng.searchBatchHits(q, k)
    print("fallbacks per batch:", eng.getTuning("batch_fallbacks") - fb0)
    bad = []
    for i in range(args.nq):
        fb0 = eng.getTuning("batch_fallbacks")
        eng.searchBatchHits(np.repeat(q[i:i + 1], 16, axis=0), k)
        if eng.getTuning("batch_fallbacks") - fb0:
            bad.append(i)
    print("queries failing alone (x16):", bad)
    # emulate the plan (engine.hip plan_onepass, defaults)
    tr = 64
    nt = (n + tr - 1) // tr
    kp = min(max(2 * k + 32, 64), 960)
    target = 8 * kp
    prow = target / n
    nl = -tr * math.log1p(-prow)
    sp = min(max(nt // 64, 64), 192)
    for g in (4, 8, 16, 32):
        a = 1 - 2 ** (-1 / g)
        l = max(1, math.floor(-math.log(a) / nl + 0.5))
        if l * g > nt // 2:
            l = (nt // 2) // g
        G, L = g, l
        if g * l >= sp:
            break
    S = G * L
    print(f"plan: ntiles {nt} G {G} L {L} S {S} kp {kp}")
    cb = torch.nn.functional.normalize(corpus, dim=1).to(torch.bfloat16).to(torch.float32)
    qt = torch.from_numpy(q).to(dev)
    qb = torch.nn.functional.normalize(qt, dim=1).to(torch.bfloat16).to(torch.float32)
    phys = torch.tensor([(i * nt) // S for i in range(S)], device=dev)
    seg = (torch.arange(n, device=dev) // tr) % 256
    stats = []
    for i in list(range(min(args.nq, 64))) + [b for b in bad if b >= 64]:
        sims = cb @ qb[i]
        tm = sims[: nt * tr].view(nt, tr).max(dim=1).values if n % tr == 0 else None
        tmax = tm[phys]                                            # sampled tile maxima, index i
        grp = torch.arange(S, device=dev) % 32 % G
        gm = torch.stack([tmax[grp == g].max() for g in range(G)])
        tau_sim = gm.min()
        surv = sims >= tau_sim
        T = int(surv.sum())
        segfill = torch.bincount(seg[surv], minlength=256)
        top = torch.sort(sims, descending=True).values
        stats.append((i, T, int(segfill.max()), float(1 - tau_sim), float(top[k - 1] - top[min(kp, T) - 1]) if T >= k else -1.0))
    for s_ in stats:
        flag = " <-- fails" if s_[0] in bad else ""
        if s_[0] in bad or s_[0] < 12:
            print("q%d survivors %d max_segment %d tau %.5f gap(k..min(kp,T)) %.5f%s" % (*s_, flag))
    Ts = np.array([s_[1] for s_ in stats])
    print("survivors: median %d min %d max %d" % (np.median(Ts), Ts.min(), Ts.max()))


if __name__ == "__main__":
    main()
