#!/usr/bin/env python3
"""What a workgroup-LOCAL admission threshold would admit (VERDICT r05 #2 proposed one: each filtering-GEMM workgroup runs its first S
tiles with tau = -inf recording tile maxima, picks tau per query from those maxima, filters the rest, revisits the S tiles — three
launches instead of five, no cross-workgroup dependency). The filtering GEMM's geometry at config 3: 256 workgroups, tiles of 64
rows, workgroup b owns tiles b, b + 256, ... (~61 tiles = 3 900 rows). This script computes, on the bench's own corpus and queries,
how many rows per query such thresholds admit, next to the global sampler's (j-th largest of 488 tile maxima spread over the store).
Similarities in f32 on the GPU (torch); counting only — nothing of the product path is used.

    python tools/local_threshold_sim.py --rows 1000000 --dims 384 --nq 256
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
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    tr, nwg = 64, 256
    ntiles = (args.rows + tr - 1) // tr
    q = torch.from_numpy(bench.unit_queries(args.nq, args.dims)).to(dev)
    # per-(tile, query) maxima and the full similarity matrix tile by tile would be 1 GB: keep per-tile sorted top-8 instead
    TOP = 8
    tile_top = torch.full((ntiles, args.nq, TOP), -2.0, device=dev)
    sims_chunks = []
    for r0, x in bench.device_rows(torch, 0, args.rows, args.dims, dev):
        s = (x @ q.T).float()                                   # [rows, nq]
        sims_chunks.append((r0, s.half()))                      # (counting with fp16 copies is enough for thresholds far from the values)
        n = s.shape[0]
        nt = (n + tr - 1) // tr
        pad = nt * tr - n
        if pad:
            s = torch.cat([s, torch.full((pad, args.nq), -2.0, device=dev)])
        t = s.view(nt, tr, args.nq).permute(0, 2, 1)            # [tiles, nq, 64]
        tile_top[r0 // tr: r0 // tr + nt] = torch.topk(t, TOP, dim=2).values
    tile_max = tile_top[:, :, 0]                                # [ntiles, nq]

    def admitted(tau):                                          # rows with sim >= tau[query], whole store, per query
        tot = torch.zeros(args.nq, device=dev)
        for r0, s in sims_chunks:
            tot += (s.float() >= tau[None, :]).sum(dim=0)
        return tot

    out = {"rows": args.rows, "dims": args.dims, "nq": args.nq, "tile_rows": tr, "workgroups": nwg, "tiles_per_workgroup": ntiles / nwg}
    # the product's global sampler: j-th largest of S tile maxima spread evenly
    S, j = 488, 8
    idx = (torch.arange(S, device=dev) * ntiles // S)
    tau_g = torch.sort(tile_max[idx], dim=0, descending=True).values[j - 1]
    out["global_sampler_j8_S488_survivors_per_query"] = float(admitted(tau_g).mean())
    # workgroup-local: first S_l tiles of the workgroup's own sequence, j-th largest of their maxima; survivors = rows of the workgroup's
    # tiles at or above it (the S_l sampled tiles are revisited with the threshold in force, so they count the same way)
    res = {}
    for S_l in (2, 4, 8, 16):
        for jl in (1, 2, 4):
            if jl > S_l:
                continue
            tot = torch.zeros(args.nq, device=dev)
            for b in range(nwg):
                mine = torch.arange(b, ntiles, nwg, device=dev)
                tm = tile_max[mine[:S_l]]                       # [S_l, nq]
                tau = torch.sort(tm, dim=0, descending=True).values[jl - 1]
                # rows >= tau within the workgroup's tiles: count through the per-tile top-8 (exact while a tile holds < 8 survivors)
                tt = tile_top[mine]                             # [tiles, nq, 8]
                tot += (tt >= tau[None, :, None]).sum(dim=(0, 2))
            res[f"first_{S_l}_tiles_rank_{jl}"] = {"survivors_per_query": float(tot.mean()), "revisit_overhead": S_l / (ntiles / nwg)}
    out["workgroup_local"] = res
    # running maximum (the threshold tightens as the workgroup goes): survivors = tiles whose maximum beats every earlier one's, at least
    tot = torch.zeros(args.nq, device=dev)
    for b in range(nwg):
        mine = torch.arange(b, ntiles, nwg, device=dev)
        tm = tile_max[mine]
        run = torch.cummax(tm, dim=0).values
        prev = torch.cat([torch.full((1, args.nq), -2.0, device=dev), run[:-1]])
        tt = tile_top[mine]
        tot += (tt >= prev[:, :, None]).sum(dim=(0, 2))
    out["workgroup_local_running_maximum_survivors_per_query"] = float(tot.mean())
    line = json.dumps(out)
    print(line)
    if args.out:
        open(args.out, "a").write(line + "\n")


if __name__ == "__main__":
    main()
