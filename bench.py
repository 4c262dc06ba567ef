#!/usr/bin/env python3
"""bench.py — headline benchmark of the Wax vector scan + top-k path on MI355X.

Workload (BASELINE.json `metric`): 10M x 384-dim f32 corpus, cosine top-10, one query per step.
A "step" is one pass of the hot path: one query scanned against the whole corpus (row-sharded
over the N GPUs, per-shard top-k all-gathered over RCCL and merged), results back on the host.
The corpus is resident in HBM before the timed region; the only host<->device traffic inside it
is the 1.5 KB query in and the k results out (both included in `value`).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

Rank 0 prints ONE JSON line (contract in the task statement): `value` = whole-job queries/s,
`roofline` = achieved HBM GB/s of the scan kernel (HIP events recorded around every scan-kernel
launch of the timed region, on the stream the kernel runs on) against the 8 TB/s MI355X peak,
`cpu_baseline` = the oracle's CPU scan timed on this host (rank 0, N=1 only; one thread and all threads),
`secondary` (N=1 only) = the other single-GPU BASELINE configurations, timed after the headline with the same
barrier/synchronise bracket, each with its own roofline block: 1M x 384 single query (config 2), 1M x 384 with
256 queries per step (config 3: bf16 MFMA GEMM + fused top-k) and one GPU's share of config 5 (1.25M x 768,
1024 queries per step); the batched ones go through the device-resident entry point (queries already in HBM).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
CORPUS_SEED = 20260220
QUERY_SEED = 7
GRANULE = 65536


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--dims", type=int, default=384)
    p.add_argument("--topk", type=int, default=10)
    p.add_argument("--depth", type=int, default=4, help="queries in flight (software pipeline)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    p.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                   help="experiments: wax_hip_set_tuning(KEY, VALUE) on every engine the bench creates (repeatable)")
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (N=1 only)")
    p.add_argument("--host-merge", action="store_true", help="N>1: merge gathered hits on the host instead of the device")
    p.add_argument("--exchange", choices=["rccl", "host"], default="rccl",
                   help="N>1: rccl = all-gather device buffers over RCCL (default); host = download + gloo all-gather "
                        "(control path; also lets two test ranks share one GPU with WAX_BENCH_SAME_DEVICE=1)")
    return p.parse_args()


TUNES = []


def apply_tunes(eng):
    for kv in TUNES:
        k, v = kv.split("=", 1)
        eng.setTuning(k, int(v))


def device_rows(torch, lo, hi, dims, dev):
    """Rows [lo, hi) of the synthetic corpus: unit-norm Gaussian, granule g seeded with CORPUS_SEED+g,
    so every GPU count sees the same global corpus."""
    g = torch.Generator(device=dev)
    r = lo
    while r < hi:
        gi = r // GRANULE
        g0 = gi * GRANULE
        g.manual_seed(CORPUS_SEED + gi)
        block = torch.randn((GRANULE, dims), generator=g, device=dev, dtype=torch.float32)
        block = torch.nn.functional.normalize(block, dim=1)
        a, b = r - g0, min(hi - g0, GRANULE)
        yield r, block[a:b].contiguous()
        r = g0 + b


def _human_rows(n):
    if n % 1_000_000 == 0:
        return f"{n // 1_000_000}M"
    if n % 1000 == 0:
        return f"{n // 1000}K"
    return str(n)


def unit_queries(n, dims):
    rng = np.random.Generator(np.random.PCG64(QUERY_SEED))
    q = rng.standard_normal((n, dims))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def usable_host_threads(omp_max):
    """Threads this process can actually run at once: the OpenMP default capped by the affinity mask and by the cgroup
    CPU quota (the GPU box is a container slice: 256 logical CPUs visible, cpu.max = 16 CPUs; oversubscribing the quota
    gets the process throttled — 256 threads scan at 9.6 GB/s, 16 threads at 133 GB/s, profiles/r02/m_cpu_sweep.json)."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(torch, args, dev, queries):
    """The oracle's CPU scan (C restatement of the reference's arithmetic: a3 + a5) on a bounded sample of the same
    corpus, timed on this host as BASELINE.md §3 prescribes: (i) one thread, (ii) all host threads (static row
    partition, per-thread heap, merge; "all" = what the cgroup quota lets run at once). The inner loop is the metric-specialised FMA kernel (cosine: dot and |v|^2
    only), and the sample is first-touched by the threads that scan it (NUMA-local pages). Reported next to the GPU
    number; never the thing measured. Wax's actual CPU engine (USearch HNSW through Swift) cannot run here."""
    import oracle
    oracle.build()
    n_s = min(args.cpu_sample_rows, args.rows)
    threads = usable_host_threads(oracle.max_threads())
    sample = oracle.numa_sample(n_s, args.dims, threads)
    for lo, x in device_rows(torch, 0, n_s, args.dims, dev):
        blk = x.cpu().numpy()
        oracle.copy_rows(sample[lo:lo + blk.shape[0]], blk, max(1, min(threads, 16)))

    def timed(nthreads, budget_s, max_queries):
        oracle.scan_topk_fast(0, sample, queries[0], args.topk, nthreads)  # warm-up / page-in
        t0 = time.perf_counter()
        done = 0
        while True:
            oracle.scan_topk_fast(0, sample, queries[done % len(queries)], args.topk, nthreads)
            done += 1
            el = time.perf_counter() - t0
            if el >= budget_s or done >= max_queries:
                break
        return done, el

    variants = []
    for nthreads, budget, cap in ((1, args.cpu_baseline_seconds / 3.0, 40), (threads, args.cpu_baseline_seconds, 4000)):
        done, el = timed(nthreads, budget, cap)
        qps_sample = done / el
        variants.append({
            "threads": nthreads, "value": qps_sample * n_s / args.rows, "unit": "queries/s",
            "sample_qps": qps_sample, "sample_gbps": n_s * args.dims * 4 * qps_sample / 1e9,
            "sample": f"{done} queries over the first {n_s} rows in {el:.1f} s",
        })
    best = variants[-1]
    return {
        "value": best["value"], "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"{best['sample']} of the same corpus on {threads} threads ({best['sample_qps']:.2f} q/s on the sample = "
                  f"{best['sample_gbps']:.1f} GB/s), scaled by {n_s}/{args.rows} rows to the full workload; "
                  f"metric-specialised FMA inner loop, NUMA first-touch by the scanning threads",
        "variants": variants,
    }


# ---------------------------------------------------------------------------
# secondary configurations (world == 1): same bracket as the headline, each with its own roofline block

def _bracket(torch):
    torch.cuda.synchronize()


def secondary_single_query(torch, dev, rows, dims, k, steps, warmup, depth, label="BASELINE config 2"):
    """BASELINE config 2 (and the 10K-row point of the north star's N matrix): rows x dims f32, one query per step, the
    headline's code path at another size."""
    from wax_amd import HIPVectorEngine, VectorMetric
    eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims)
    eng.reserve(rows)
    for r0, x in device_rows(torch, 0, rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    queries = unit_queries(warmup + steps, dims)
    eng.setTuning("time_kernels", 1)
    eng.setTuning("streams", 2)
    eng.setTuning("slots", max(depth, 2))
    apply_tunes(eng)

    def run(qs):
        pending = []
        for q in qs:
            if len(pending) >= depth:
                eng.collect(pending.pop(0), k)
            pending.append(eng.submit(q, k))
        while pending:
            eng.collect(pending.pop(0), k)

    run(queries[:warmup])
    eng.setTuning("reset_stats", 1)
    _bracket(torch)
    t0 = time.perf_counter()
    run(queries[warmup:])
    _bracket(torch)
    el = time.perf_counter() - t0
    st = eng.stats()
    launches = int(st.scan_kernels_timed)
    kern_ms = st.scan_kernel_ms_total / launches if launches else float("nan")
    nbytes = rows * dims * 4
    achieved = nbytes / (kern_ms * 1e-3) / 1e9
    eng.close()
    return {
        "config": f"{rows} x {dims} f32 cosine top-{k}, one query per step, 1 GPU ({label})",
        "value": steps / el, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
        "dtype": "f32",
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "kernel": "wax::scan_kernel", "kernel_avg_ms": kern_ms, "kernel_launches_timed": launches,
                     "algorithmic_bytes_per_launch": nbytes},
    }


def secondary_batched(torch, dev, rows, dims, nq, k, steps, warmup, label, row_base=0):
    """BASELINE configs 3 / 5 (one GPU's share): nq queries per step as a bf16 MFMA GEMM + fused top-k + exact f32
    re-score. Queries and results stay in HBM (wax_hip_search_batch_submit_device / _collect_device): the timed region
    holds no host<->device traffic except nq certificate flags per step."""
    from wax_amd import HIPVectorEngine, VectorMetric
    eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims)
    eng.reserve(rows)
    for r0, x in device_rows(torch, 0, rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    eng.setRowBase(row_base)
    apply_tunes(eng)
    depth = 2                                # batches in flight, like the headline's --depth software pipeline
    dq = torch.from_numpy(unit_queries(nq, dims)).to(dev)
    outs = [torch.empty((nq, k, 2), dtype=torch.int64, device=dev) for _ in range(depth)]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(n_steps):
        """n_steps batches through wax_hip_search_batch_submit_device / _collect_device, `depth` tickets in flight:
        the next batch's launches and the host's wake-up hide under the running batch. Every batch is complete
        (certificates checked, fallbacks re-run) at its collect."""
        tickets = []
        for i in range(n_steps):
            if len(tickets) == depth:
                eng.searchBatchCollectDevice(tickets.pop(0))
            tickets.append(eng.searchBatchSubmitDevice(dq.data_ptr(), nq, k, outs[i % depth].data_ptr(), k, stream))
        for t in tickets:
            eng.searchBatchCollectDevice(t)

    eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, stream)   # builds the bf16 mirror (untimed, like the corpus upload)
    run(warmup)
    # one blocking call per step, for reference: what a caller that cannot pipeline sees
    _bracket(torch)
    tb = time.perf_counter()
    for _ in range(max(10, steps // 4)):
        eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, stream)
    _bracket(torch)
    blocking_ms = (time.perf_counter() - tb) / max(10, steps // 4) * 1e3
    fb0 = eng.getTuning("batch_fallbacks")
    eng.setTuning("time_kernels", 1)
    apply_tunes(eng)
    eng.setTuning("reset_stats", 1)
    _bracket(torch)
    t0 = time.perf_counter()
    run(steps)
    _bracket(torch)
    el = time.perf_counter() - t0
    st = eng.stats()
    launches = int(st.batch_gemms_timed)
    kern_ms = st.batch_gemm_ms_total / launches if launches else float("nan")
    flops = 2.0 * nq * rows * dims
    nbytes = rows * dims * 2                  # bf16 mirror, streamed once per launch
    t_hbm, t_mfma = nbytes / (HBM_PEAK_GBPS * 1e9), flops / (MFMA_BF16_PEAK_TFLOPS * 1e12)
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    if bound == "hbm":
        achieved, peak, unit = nbytes / (kern_ms * 1e-3) / 1e9, HBM_PEAK_GBPS, "GB/s"
    else:
        achieved, peak, unit = flops / (kern_ms * 1e-3) / 1e12, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s"
    traffic, traffic_source = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json"))).get("batched", {})
        ent = tj.get("configs", {}).get(f"{rows}x{dims}xq{nq}")
        if ent:
            traffic = ent["hbm_bytes_per_launch"]
            traffic_source = "replayed from profiles/latest_traffic.json (" + tj.get("source", "") + "), not measured in this run"
    except (OSError, ValueError, KeyError):
        pass
    res = {
        "config": label,
        "value": nq * steps / el, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
        "dtype": "bf16 GEMM, exact f32 re-score", "queries_per_step": nq, "batches_in_flight": depth,
        "ms_per_step_blocking_call": blocking_ms,
        "end_to_end_tflops_bf16": flops / (el / steps) / 1e12,
        "end_to_end_frac_of_roof": max(t_hbm, t_mfma) / (el / steps),
        "certificate_fallbacks": int(eng.getTuning("batch_fallbacks") - fb0),
        "pipeline": "one-pass" if eng.getTuning("onepass_queries") > 0 else "slab",
        "roofline": {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                     "kernel": "wax::batch_gemm_rega_kernel" if dims != 768 else "wax::batch_gemm_ksplit_kernel",
                     "kernel_avg_ms": kern_ms, "kernel_launches_timed": launches,
                     "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
                     "hbm_floor_ms": t_hbm * 1e3, "mfma_floor_ms": t_mfma * 1e3,
                     "traffic": traffic, "traffic_source": traffic_source},
    }
    eng.close()
    return res


def main():
    args = parse_args()
    TUNES.extend(args.tune)
    # RCCL / CUDA-tensor IPC on this driver stack needs dmabuf IPC (the image exports it; keep it if launched bare)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # `--gpus N` in ONE process (no torch.distributed launcher): the library's own multi-GPU engine
    # (wax_hip_engine_create_sharded) spreads the corpus over devices 0..N-1 behind the same handle.
    in_library = world == 1 and args.gpus > 1
    if world != args.gpus and not in_library:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run for N>1")
    if os.environ.get("WAX_BENCH_SAME_DEVICE"):  # testing only: several ranks on one GPU (needs --exchange host)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_rccl = args.exchange == "rccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_rccl:
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(seconds=300))
            probe_t = torch.ones(1, device=dev)
            dist.all_reduce(probe_t)  # fail here, loudly and at once, if RCCL cannot talk across the node
            assert int(probe_t.item()) == world
            # (A silent fall-back to the gloo host exchange was tried and removed: tearing down a half-initialised
            # RCCL group hangs instead of failing. `--exchange host` selects the host exchange explicitly.)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from wax_amd import HIPVectorEngine, VectorMetric, build, sharded
    build.build()
    if not HIPVectorEngine.isAvailable():
        raise SystemExit("no gfx950 device visible: bench.py measures the HIP path only")

    n, dims, k = args.rows, args.dims, args.topk
    lo, hi = sharded.shard_bounds(n, world, rank, align=64)
    t_build = time.perf_counter()
    if in_library:
        same = bool(os.environ.get("WAX_BENCH_SAME_DEVICE"))          # testing only: every shard on GPU 0
        devs = [0 if same else g for g in range(args.gpus)]
        eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims, devices=devs)
        eng.reserve(n)                                                 # block layout: ceil(n / N) rows per shard
        per = -(-n // args.gpus)
        per = -(-per // 64) * 64
        r = 0
        while r < n:                                                   # every granule is generated on the device that will hold it
            g_ = min(r // per, args.gpus - 1)
            gdev = torch.device("cuda", devs[g_])
            r_hi = min(n, (r // GRANULE + 1) * GRANULE, (g_ + 1) * per if g_ + 1 < args.gpus else n)
            for r0, x in device_rows(torch, r, r_hi, dims, gdev):
                eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
            r = r_hi
        for d_ in set(devs):
            torch.cuda.synchronize(d_)
        lo, hi = 0, -(-n // args.gpus)                                  # rows per launch of ONE shard's scan kernel (roofline)
    else:
        eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims)
        eng.reserve(max(hi - lo, 1))
        for r0, x in device_rows(torch, lo, hi, dims, dev):
            eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
        eng.setRowBase(lo)
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}/{world}: shard rows [{lo},{hi}) = {(hi - lo) * dims * 4 / 1e9:.2f} GB in HBM, "
        f"built in {time.perf_counter() - t_build:.1f} s")

    total = args.warmup + args.steps
    queries = unit_queries(max(total, 8), dims)

    # untimed sanity check: a stored row must retrieve itself
    probe_row = min(n - 1, 123457)
    probe = None
    for r0, x in device_rows(torch, probe_row, probe_row + 1, dims, dev):
        probe = x[0].cpu().numpy()

    eng.setTuning("time_kernels", 1)
    apply_tunes(eng)
    if in_library and args.exchange == "rccl" and not os.environ.get("WAX_BENCH_SAME_DEVICE"):
        eng.setTuning("exchange", 1)        # one ncclAllGather per query on the library's single-process communicator
    if world == 1:
        # two in-order streams; the library chains the scan kernels through an event so they never
        # overlap each other (per-kernel HIP-event times stay clean) while one query's merge / result
        # write / next query upload hide under the neighbouring scan
        eng.setTuning("streams", 2)
        eng.setTuning("slots", max(args.depth, 2))
        searcher = None

        def submit(q):
            return eng.submit(q, k)

        def collect(t):
            return eng.collect(t, k)
    else:
        searcher = sharded.ShardedSearcher(eng, rank, world, k, depth=args.depth, n_streams=2,
                                           host_merge=args.host_merge, exchange="rccl" if use_rccl else "host")

        def submit(q):
            searcher.submit(q)
            return None

        def collect(_):
            return searcher.collect()

    ids, scores = collect(submit(probe))
    assert int(ids[0]) == probe_row and abs(float(scores[0]) - 1.0) < 1e-5, (ids[:3], scores[:3])

    def run(qs):
        pending = []
        last = None
        for q in qs:
            if len(pending) >= args.depth:
                last = collect(pending.pop(0))
            pending.append(submit(q))
        while pending:
            last = collect(pending.pop(0))
        return last

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if use_rccl:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # Python's cyclic collector must not fire inside the timed region: with torch imported a full (generation-2)
    # collection walks millions of objects and pauses the submitting thread for 40-70 ms (seen as a one-off stall
    # around the 450th query of a run, tools/long_run_drift.py); nothing below creates reference cycles.
    import gc
    gc.collect()
    gc.disable()
    run(queries[:args.warmup])
    eng.setTuning("reset_stats", 1)
    barrier()
    t0 = time.perf_counter()
    last = run(queries[args.warmup:args.warmup + args.steps])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if use_rccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = eng.stats()
    assert len(last[0]) == min(k, n)
    import hashlib
    checksum = hashlib.sha256(np.asarray(last[0], dtype=np.uint64).tobytes()
                              + np.asarray(last[1], dtype=np.float32).tobytes()).hexdigest()[:16]

    if rank == 0:
        qps = args.steps / elapsed
        launches = int(st.scan_kernels_timed)
        kern_ms = (st.scan_kernel_ms_total / launches) if launches else float("nan")
        bytes_per_launch = (hi - lo) * dims * 4
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if launches else float("nan")
        # HBM traffic needs a PMC pass of its own (rocprofv3 --pmc FETCH_SIZE, never combined with tracing): it cannot be
        # measured inside this run. A figure REPLAYED from the committed counter pass of this same command is reported
        # with its source; without a matching pass the field is null.
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("rows_per_launch") == hi - lo and tj.get("dims") == dims:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "replayed from profiles/latest_traffic.json (" + str(tj.get("source", "rocprofv3 --pmc FETCH_SIZE pass of this command")) + "), not measured in this run"
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": f"queries/sec, {_human_rows(n)} x {dims}-dim f32 cosine top-{k} brute-force scan (single query per step)",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": args.gpus if in_library else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n} x {dims}-dim f32 unit-norm Gaussian corpus (seed {CORPUS_SEED}), cosine top-{k}, "
                            f"one query per step, corpus resident in HBM and row-sharded over {args.gpus if in_library else world} GPU(s)",
                "rows": n, "dims": dims, "top_k": k, "rows_per_gpu": hi - lo,
                "parallelism": (f"row-shard x{args.gpus}, ONE process: the library's multi-GPU engine (wax_hip_engine_create_sharded), "
                                + ("single-process RCCL all-gather" if eng.getTuning("exchange") == 1 else "peer-copy gather")
                                + " of per-shard top-k + merge on the first device") if in_library else
                               (f"row-shard x{world}" + ((" + RCCL all-gather of per-shard top-k" if use_rccl else
                                                          " + host (gloo) all-gather of per-shard top-k") if world > 1 else "")),
                "pipeline_depth": args.depth,
                "merge": "host" if (world > 1 and (args.host_merge or not use_rccl)) else "device",
                "exchange": ("in-library" if in_library else (("rccl all_gather" if use_rccl else "host (gloo)") if world > 1 else "none")),
                "last_result_checksum": checksum,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS if launches else None,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "wax::scan_kernel (fused scan + per-wave top-k)",
                "kernel_avg_ms": kern_ms,
                "kernel_launches_timed": launches,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": "rows_per_gpu*dims*4 bytes per launch / mean HIP-event duration of the scan kernel over the "
                        "timed region (rank 0's shard; scans are chained across the two pipeline streams, so they "
                        "never overlap each other)",
            },
        }
        if world == 1 and not in_library and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, args, dev, queries)
        elif world == 1:
            out["cpu_baseline"] = None
        if world == 1 and not in_library and not args.no_secondary:
            # the other single-GPU BASELINE configurations, timed after the headline (its engine is released first)
            gc.enable()
            eng.close()
            gc.collect()
            gc.disable()
            sec = []
            for fn in (lambda: secondary_single_query(torch, dev, 10_000, 384, k, max(args.steps, 2000), max(args.warmup, 100), args.depth,
                                                      "the 10K-row point of the N matrix: launch-latency-bound, 15 MB per query"),
                       lambda: secondary_single_query(torch, dev, 1_000_000, 384, k, max(args.steps, 100), max(args.warmup, 10), args.depth),
                       lambda: secondary_batched(torch, dev, 1_000_000, 384, 256, k, max(args.steps, 200), max(args.warmup, 20),
                                                 "1000000 x 384, 256 queries per step, cosine top-10, bf16 MFMA GEMM + fused top-k, 1 GPU "
                                                 "(BASELINE config 3), queries and results resident in HBM, 2 batches in flight"),
                       lambda: secondary_batched(torch, dev, 1_000_000, 384, 1024, k, max(args.steps // 2, 100), max(args.warmup, 20),
                                                 "1000000 x 384, 1024 queries per step, cosine top-10, bf16 MFMA GEMM + fused top-k, 1 GPU "
                                                 "(config 3 at four times the batch: the MFMA-bound shape), queries and results resident in "
                                                 "HBM, 2 batches in flight"),
                       lambda: secondary_batched(torch, dev, 1_250_000, 768, 1024, k, max(args.steps // 2, 60), max(args.warmup, 10),
                                                 "1250000 x 768 (one GPU's share of 10M x 768 over 8 GPUs), 1024 queries per step, cosine "
                                                 "top-10, bf16 MFMA GEMM + fused top-k (BASELINE config 5, per-GPU part), queries and "
                                                 "results resident in HBM, 2 batches in flight", row_base=3_750_000)):
                try:
                    sec.append(fn())
                except Exception as ex:  # noqa: BLE001 — a secondary failure must not lose the headline line
                    sec.append({"error": f"{type(ex).__name__}: {ex}"})
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        if use_rccl:
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
