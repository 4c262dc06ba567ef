#!/usr/bin/env python3
"""Batched-query benchmark (BASELINE configs 3 / 5): Q x D^T as a bf16 MFMA GEMM with fused selection
and exact f32 re-score, optionally row-sharded over N GPUs (one process per GPU, per-shard top-k hits
all-gathered over RCCL and merged per query). Prints one JSON line per configuration on rank 0.

    python tools/batch_bench.py --rows 1000000 --dims 384 --nq 256                     # config 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29501 tools/batch_bench.py --gpus 8 --rows 10000000 --dims 768 --nq 1024   # config 5
"""
from __future__ import annotations

import argparse
import hashlib
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
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--nq", type=int, nargs="+", default=[256])
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--metric", type=int, default=0, help="0 cosine, 1 dot, 2 l2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--slab-mb", type=int, nargs="+", default=[64])
    ap.add_argument("--growth", type=int, nargs="+", default=[0], help="slab growth factors to sweep (0 = library default)")
    ap.add_argument("--debug", type=int, nargs="+", default=[0], help="batch_debug bit masks to sweep (timing experiments)")
    ap.add_argument("--first", type=int, nargs="+", default=[0], help="rows of the dense first slab to sweep (0 = library default)")
    ap.add_argument("--batch-min", type=int, nargs="+", default=[-1], help="batch_min values to sweep (-1 = library default)")
    ap.add_argument("--rega", type=int, nargs="+", default=[-1], help="batch_rega modes to sweep (-1 = library default)")
    ap.add_argument("--onepass", type=int, nargs="+", default=[-1], help="batch_onepass values to sweep (-1 = library default)")
    ap.add_argument("--survivors", type=int, nargs="+", default=[0], help="batch_survivors values to sweep (0 = library default)")
    ap.add_argument("--sample-div", type=int, nargs="+", default=[0], help="batch_sample_div values to sweep (0 = library default)")
    ap.add_argument("--exchange", choices=["rccl", "host"], default="rccl")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if os.environ.get("WAX_BENCH_SAME_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.exchange == "rccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    import wax_amd as wax
    from wax_amd import sharded
    lo, hi = sharded.shard_bounds(args.rows, world, rank, align=128)
    eng = wax.HIPVectorEngine(metric=args.metric, dimensions=args.dims)
    eng.reserve(max(hi - lo, 1))
    for r0, x in bench.device_rows(torch, lo, hi, args.dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    eng.setRowBase(lo)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if args.exchange == "rccl" else dist.barrier()

    import gc
    gc.collect()
    gc.disable()    # a generation-2 collection with torch loaded pauses the thread for tens of ms (tools/long_run_drift.py)
    for nq in args.nq:
        q = bench.unit_queries(nq, args.dims)
        dq = torch.from_numpy(q).to(dev)
        dout = torch.empty((nq, args.topk, 2), dtype=torch.int64, device=dev)
        for slab, growth, dbg, rega, first, bmin, onepass, surv, sdiv in [
                (s_, g_, d_, r_, f_, m_, o_, v_, x_) for s_ in args.slab_mb for g_ in args.growth for d_ in args.debug
                for r_ in args.rega for f_ in args.first for m_ in args.batch_min for o_ in args.onepass
                for v_ in args.survivors for x_ in args.sample_div]:
            if onepass >= 0:
                eng.setTuning("batch_onepass", onepass)
            if surv:
                eng.setTuning("batch_survivors", surv)
            if sdiv:
                eng.setTuning("batch_sample_div", sdiv)
            if bmin >= 0:
                eng.setTuning("batch_min", bmin)
            eng.setTuning("batch_slab_mb", slab)
            if first:
                eng.setTuning("batch_first", first)
            if rega >= 0:
                eng.setTuning("batch_rega", rega)
            eng.setTuning("batch_debug", dbg)
            if growth:
                eng.setTuning("batch_growth", growth)
            sharded.sharded_search_batch(eng, q, args.topk, world, args.exchange)  # warm-up (+ mirror build)
            fb0 = eng.getTuning("batch_fallbacks")
            sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                ids, scores, valid = sharded.sharded_search_batch(eng, q, args.topk, world, args.exchange)
            sync()
            dt = (time.perf_counter() - t0) / args.reps
            t1 = time.perf_counter()                       # the C-ABI call alone (no all-gather, no numpy decode)
            for _ in range(args.reps):
                eng.searchBatchHits(q, args.topk)
            dt_call = (time.perf_counter() - t1) / args.reps
            st = torch.cuda.current_stream(dev).cuda_stream          # device-resident call: queries and hits stay in HBM
            eng.searchBatchHitsDevice(dq.data_ptr(), nq, args.topk, dout.data_ptr(), args.topk, st)
            torch.cuda.synchronize()
            eng.setTuning("time_kernels", 1)
            eng.setTuning("reset_stats", 1)
            t2 = time.perf_counter()
            for _ in range(args.reps):
                eng.searchBatchHitsDevice(dq.data_ptr(), nq, args.topk, dout.data_ptr(), args.topk, st)
            torch.cuda.synchronize()
            dt_dev = (time.perf_counter() - t2) / args.reps
            stt = eng.stats()
            gemm_us = stt.batch_gemm_ms_total / stt.batch_gemms_timed * 1e3 if stt.batch_gemms_timed else float("nan")
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev if args.exchange == "rccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if rank == 0:
                chk = hashlib.sha256(np.ascontiguousarray(ids).tobytes() + np.ascontiguousarray(scores).tobytes()).hexdigest()[:16]
                print(json.dumps({"n_gpus": world, "rows": args.rows, "dims": args.dims, "nq": nq, "topk": args.topk,
                                  "slab_mb": slab, "growth": eng.getTuning("batch_growth"), "debug": dbg, "batch_min": eng.getTuning("batch_min"), "first": eng.getTuning("batch_first"), "rega": eng.getTuning("batch_rega"), "ms_per_batch": dt * 1e3, "ms_c_call": dt_call * 1e3, "qps_c_call": nq / dt_call,
                                  "ms_device_call": dt_dev * 1e3, "gemm_kernel_us": gemm_us, "tflops_bf16_device_call": 2.0 * nq * (hi - lo) * args.dims / dt_dev / 1e12,
                                  "onepass": eng.getTuning("batch_onepass"), "survivors": eng.getTuning("batch_survivors"), "sample_div": eng.getTuning("batch_sample_div"),
                                  "tflops_bf16_c_call": 2.0 * nq * (hi - lo) * args.dims / dt_call / 1e12, "qps": nq / dt,
                                  "tflops_bf16": 2.0 * nq * args.rows * args.dims / dt / 1e12,
                                  "fallbacks_rank0": eng.getTuning("batch_fallbacks") - fb0,
                                  "exchange": args.exchange if world > 1 else "none", "result_checksum": chk}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
