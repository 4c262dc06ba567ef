#!/usr/bin/env python3
"""Randomised cross-check of the batched (bf16 MFMA, one-pass / slab) path against the single-query f32 path: random store
size, dimension, metric, k, batch size, row_base, clustered / duplicated rows. Every checked answer must be identical
(ids and scores); prints the plan-relevant counters so that rare planner corners (small stores, large k) are visible.
--sharded P: with probability P a trial ALSO loads the rows into a sharded handle (2-5 shards, all on GPU 0) and demands that
its batched answer (host buffers, and two device-resident tickets in flight) equals the single engine's hit for hit."""
import argparse
import os
os.environ.setdefault("WAX_HIP_SHARD_MIN_MB", "0")   # sharded handles of the fuzz spread their (small) stores over every shard
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
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--sharded", type=float, default=0.0, help="probability of also checking a sharded handle (shards on GPU 0)")
ap.add_argument("--rows-min", type=int, default=20_000)
ap.add_argument("--rows-max", type=int, default=400_000, help="80 %% of the stores have rows-min .. rows-max rows, the others up to 3 x rows-max")
ap.add_argument("--tune", action="append", default=[], help="KEY=VALUE set on every engine")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
t_end = time.time() + args.seconds
trials = checked = fallbacks = onepass_q = retries = sharded_trials = 0
while time.time() < t_end:
    dims = int(rng.choice([128, 256, 384, 512, 768, 192]))
    n = int(rng.integers(args.rows_min, args.rows_max)) if rng.random() < 0.8 else int(rng.integers(args.rows_max, 3 * args.rows_max))   # (the larger ones: a tail pool at one query group)
    metric = int(rng.choice([0, 0, 1, 2]))
    k = int(rng.choice([1, 5, 10, 30, 64, 100, 200, 300, 460]))
    nq = int(rng.choice([16, 17, 64, 255, 256, 257, 300, 700, 1024, 1500]))
    g = torch.Generator(device=dev)
    g.manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, dims), generator=g, device=dev, dtype=torch.float32)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        x = torch.nn.functional.normalize(x, dim=1)
    elif mode == 1:                                  # clustered: 20 centres
        c = torch.randn((20, dims), generator=g, device=dev)
        x = torch.nn.functional.normalize(c[torch.randint(0, 20, (n,), generator=g, device=dev)] + 0.3 * x, dim=1)
    elif mode == 2:                                  # duplicates and scaled rows
        x = torch.nn.functional.normalize(x, dim=1)
        x[n // 2:n // 2 + 500] = x[7]
        x = x * (0.5 + torch.rand((n, 1), generator=g, device=dev))
    eng = wax.HIPVectorEngine(metric=wax.VectorMetric(metric), dimensions=dims)
    for kv in args.tune:
        eng.setTuning(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
    eng.reserve(n)
    eng.addBatchDevice(np.arange(n, dtype=np.uint64) * 3 + 1, x.contiguous())
    row_base = int(rng.integers(0, 1 << 20)) if rng.random() < 0.5 else 0
    if row_base:
        eng.setRowBase(row_base)
    q = torch.randn((nq, dims), generator=g, device=dev)
    if mode == 1:
        q[: nq // 2] = x[torch.randint(0, n, (nq // 2,), generator=g, device=dev)] + 0.05 * q[: nq // 2]
    qh = q.cpu().numpy()
    b0, f0, o0 = eng.getTuning("batch_queries"), eng.getTuning("batch_fallbacks"), eng.getTuning("onepass_queries")
    r0 = eng.getTuning("batch_retries")
    ids, scores, counts = eng.searchBatch(qh, k)
    sel = rng.permutation(nq)[:12]
    for i in sel:
        s_ids, s_scores = eng.searchArrays(qh[i], k)
        kk = len(s_ids)
        ok = counts[i] == kk and np.array_equal(ids[i, :kk], s_ids) and np.array_equal(scores[i, :kk], s_scores)
        if not ok:
            print(json.dumps({"FAIL": True, "n": n, "dims": dims, "metric": metric, "k": k, "nq": nq, "mode": mode, "query": int(i)}), flush=True)
            sys.exit(1)
        checked += 1
    if args.sharded > 0 and rng.random() < args.sharded and k <= 1000:
        shards = int(rng.choice([2, 3, 5]))
        many = wax.HIPVectorEngine(metric=wax.VectorMetric(metric), dimensions=dims, devices=[0] * shards)
        for kv in args.tune:
            many.setTuning(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
        many.reserve(n)
        many.addBatchDevice(np.arange(n, dtype=np.uint64) * 3 + 1, x.contiguous())
        h_one, c_one = eng.searchBatchHits(qh, k) if row_base == 0 else (None, None)
        if h_one is None:                      # the handle has no row_base: compare ids / scores instead of raw keys
            m_ids, m_scores, m_counts = many.searchBatch(qh, k)
            ok = np.array_equal(m_counts, counts) and np.array_equal(m_ids, ids) and np.array_equal(m_scores, scores)
        else:
            h_many, c_many = many.searchBatchHits(qh, k)
            ok = np.array_equal(h_one, h_many) and np.array_equal(c_one, c_many)
            kk = min(k, 192)
            dq = q.contiguous()
            st = torch.cuda.current_stream(dev).cuda_stream
            ref = torch.empty((nq, kk, 2), dtype=torch.int64, device=dev)
            eng.searchBatchHitsDevice(dq.data_ptr(), nq, kk, ref.data_ptr(), kk, st)
            outs = [torch.empty((nq, kk, 2), dtype=torch.int64, device=dev) for _ in range(2)]
            ts = [many.searchBatchSubmitDevice(dq.data_ptr(), nq, kk, outs[i].data_ptr(), kk, st) for i in range(2)]
            for t in ts:
                many.searchBatchCollectDevice(t)
            ok = ok and all(torch.equal(o, ref) for o in outs)
        many.close()
        if not ok:
            print(json.dumps({"FAIL": "sharded", "n": n, "dims": dims, "metric": metric, "k": k, "nq": nq, "mode": mode, "shards": shards}), flush=True)
            sys.exit(1)
        sharded_trials += 1
    trials += 1
    fallbacks += eng.getTuning("batch_fallbacks") - f0
    onepass_q += eng.getTuning("onepass_queries") - o0
    retries += eng.getTuning("batch_retries") - r0
    print(json.dumps({"n": n, "dims": dims, "metric": metric, "k": k, "nq": nq, "mode": mode,
                      "mfma_queries": eng.getTuning("batch_queries") - b0, "onepass": eng.getTuning("onepass_queries") - o0,
                      "fallbacks": eng.getTuning("batch_fallbacks") - f0, "retries": eng.getTuning("batch_retries") - r0}), flush=True)
    eng.close()
    del x, q
print(json.dumps({"trials": trials, "sharded_trials": sharded_trials, "answers_checked": checked, "fallbacks": fallbacks, "full_retries": retries, "onepass_queries": onepass_q, "ok": True}))
