#!/usr/bin/env python3
"""Host fan-out cost of the in-library multi-GPU handle, measured on ONE GPU: G shards on device 0 (devices=[0]*G) do the
same device work as one engine holding all the rows, so whatever the sharded handle is slower by is host-side fan-out
(per-shard submits, events, peer copies, merge launch) — the part of multi-GPU scaling that can be measured without a
multi-GPU node. (xGMI transfer time itself is NOT measured here: every "peer copy" is a same-device copy.)

  A  tiny corpus (2000 rows per shard): microseconds per query at pipeline depth 1 and 4, handle vs one engine
  B  8 x 1.25M x 384 against one 10M x 384 engine: single-query queries/s at depth 4 (the headline shape)
  C  8 x 1.25M x 768 against one 10M x 768 engine: 1024-query batches, blocking call and two tickets in flight (config 5)

Prints one JSON line per measurement.  python tools/sharded_handle_bench.py [--parts A,B,C] [--shards 8]"""
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
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--parts", default="A,B,C")
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
G = args.shards


def load(rows, dims, devices=None, pre=()):
    eng = wax.HIPVectorEngine(dimensions=dims) if devices is None else wax.HIPVectorEngine(dimensions=dims, devices=devices)
    for k, v in pre:                     # handle-level settings that must be in place before the layout is chosen
        eng.setTuning(k, v)
    eng.reserve(rows)
    for r0, x in bench.device_rows(torch, 0, rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    for kv in args.tune:
        k, v = kv.split("=", 1)
        eng.setTuning(k, int(v))
    return eng


def time_single(eng, queries, k, depth, steps):
    def run(qs):
        pend = []
        for q in qs:
            if len(pend) >= depth:
                eng.collect(pend.pop(0), k)
            pend.append(eng.submit(q, k))
        while pend:
            eng.collect(pend.pop(0), k)
    run(queries[:20])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(queries[:steps])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def time_batched(eng, dq, nq, k, depth, steps):
    outs = [torch.empty((nq, k, 2), dtype=torch.int64, device=dev) for _ in range(max(depth, 1))]
    st = torch.cuda.current_stream(dev).cuda_stream
    eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, st)      # mirrors
    eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, st)

    def run(n):
        if depth <= 1:
            for _ in range(n):
                eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, st)
            return
        tickets = []
        for i in range(n):
            if len(tickets) == depth:
                eng.searchBatchCollectDevice(tickets.pop(0))
            tickets.append(eng.searchBatchSubmitDevice(dq.data_ptr(), nq, k, outs[i % depth].data_ptr(), k, st))
        for t in tickets:
            eng.searchBatchCollectDevice(t)
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, outs[(steps - 1) % max(depth, 1)].cpu().numpy()


parts = args.parts.split(",")
if "A" in parts:
    dims, per = 384, 2000
    queries = bench.unit_queries(2000, dims)
    one = load(per * G, dims)
    one.setTuning("streams", 2)
    one.setTuning("slots", 4)
    # the fan-out proper (small-store rule off: the 24 MB store IS spread over the G shards), round-4 device gather against the
    # round-5 per-shard tickets submitted side by side; and the default (rule on: the store stays on the first shard)
    for label, pre, post in (("spread, device gather (round 4)", (("shard_min_mb", 0),), (("ticket_path", 0),)),
                             ("spread, per-shard tickets (round 5)", (("shard_min_mb", 0),), (("ticket_path", 1),)),
                             ("default (small-store rule: one shard holds it)", (), ())):
        many = load(per * G, dims, [0] * G, pre)
        many.setTuning("streams", 2)
        many.setTuning("slots", 4)
        for k, v in post:
            many.setTuning(k, v)
        for depth in (1, 4):
            a, b = time_single(one, queries, 10, depth, 2000), time_single(many, queries, 10, depth, 2000)
            print(json.dumps({"part": "A", "what": f"tiny corpus ({per} rows x {G} shards), single query, depth {depth}: {label}", "one_engine_us": a * 1e6,
                              "sharded_handle_us": b * 1e6, "fanout_us_per_query": (b - a) * 1e6, "per_shard_us": (b - a) * 1e6 / G,
                              "rows_per_shard": [many.shardInfo(g)[2] for g in range(G)]}), flush=True)
        many.close()
    one.close()
if "B" in parts:
    dims, n = 384, args.rows
    queries = bench.unit_queries(300, dims)
    res = {}
    for name, devs in (("one", None), ("sharded", [0] * G)):
        eng = load(n, dims, devs)
        eng.setTuning("streams", 2)
        eng.setTuning("slots", 4)
        res[name] = time_single(eng, queries, 10, 4, 200)
        res[name + "_ids"] = eng.searchArrays(queries[0], 10)[0].tolist()
        eng.close()
    assert res["one_ids"] == res["sharded_ids"]
    print(json.dumps({"part": "B", "what": f"{G} x {n // G} x {dims} on one GPU vs one {n} x {dims} engine, single query, depth 4",
                      "one_engine_ms": res["one"] * 1e3, "sharded_handle_ms": res["sharded"] * 1e3,
                      "ratio": res["sharded"] / res["one"], "one_qps": 1 / res["one"], "sharded_qps": 1 / res["sharded"]}), flush=True)
if "C" in parts:
    dims, n, nq, k = 768, args.rows, 1024, 10
    dq = torch.from_numpy(bench.unit_queries(nq, dims)).to(dev)
    res = {}
    for name, devs in (("one", None), ("sharded", [0] * G)):
        eng = load(n, dims, devs)
        res[name + "_blocking"], h1 = time_batched(eng, dq, nq, k, 1, 8)
        res[name + "_depth2"], h2 = time_batched(eng, dq, nq, k, 2, 12)
        res[name + "_hits"] = h2
        assert np.array_equal(h1, h2)
        res[name + "_fallbacks"] = eng.getTuning("batch_fallbacks")
        eng.close()
    assert np.array_equal(res["one_hits"], res["sharded_hits"])
    print(json.dumps({"part": "C", "what": f"{G} x {n // G} x {dims} on one GPU vs one {n} x {dims} engine, {nq} queries per batch",
                      "one_engine_ms_blocking": res["one_blocking"] * 1e3, "sharded_handle_ms_blocking": res["sharded_blocking"] * 1e3,
                      "one_engine_ms_depth2": res["one_depth2"] * 1e3, "sharded_handle_ms_depth2": res["sharded_depth2"] * 1e3,
                      "ratio_blocking": res["sharded_blocking"] / res["one_blocking"], "ratio_depth2": res["sharded_depth2"] / res["one_depth2"],
                      "fallbacks": [res["one_fallbacks"], res["sharded_fallbacks"]]}), flush=True)
