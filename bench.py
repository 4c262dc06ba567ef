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
    python bench.py --gpus 8            # ONE process: the library's own multi-GPU engine (sharded handle)

Rank 0 prints ONE JSON line (contract in the task statement):
  `value`        whole-job queries/s of the timed region, run the way the product runs: queries software-pipelined over two
                 streams, scans free to overlap ("scan_chain" auto: no event chain while kernels are not being timed);
  `roofline`     achieved HBM GB/s of the scan kernel against the 8 TB/s MI355X peak. `frac` is PER LAUNCH: HIP events on
                 every scan-kernel launch, on the stream the kernel runs on, in calibration passes of the SAME run right
                 after the timed region (same engine, same queries, kernels chained so that a launch runs alone): the event
                 pair bound to the dispatch (hipExtLaunchKernel; `events: kernel-bound`), with the mean of the hipEventRecord
                 bracket around the same launches beside it (`kernel_avg_ms_bracketed`). `pipeline_frac` prices the timed
                 region itself: bytes per launch x steps / elapsed;
  `cpu_baseline` the oracle's CPU scan timed on this host (rank 0, N=1 only; one thread and all threads);
  `secondary`    the other BASELINE configurations, each with the same barrier/synchronise bracket and its own roofline
                 block. N=1: 10K / 1M x 384 single query (configs 1-2), 1M x 384 with 256 and 1024 queries per step
                 (config 3), config 5 at full size on one GPU (10M x 768, 1024 queries per step) and its 8-GPU share
                 (1.25M x 768), and the batched path on clustered / DeterministicEmbedder corpora (certificate fallbacks).
                 N>1: config 5 sharded over the N GPUs (the scaling curve BASELINE.json asks for), through the launch shape
                 in use — torchrun ranks + RCCL all-gather, or one process on the sharded handle.
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
SECONDARY_N1 = ["s10k", "s1m", "s1250k", "s10m_k300", "s1m_k1000", "b30k_k100", "b1m_q256", "b1m_q1024", "c5_shard", "c5_full", "clustered_k10", "clustered_k100", "dups17", "detembed"]
DUP_ROWS, DUP_AT, DUP_OF, DUP_QUERIES = 2048, 500_000, 7, 44    # the "dups17" corpus: 2048 copies of row 7; 44 of 256 queries (17 %) aim at it


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
    p.add_argument("--cpu-sample-rows", type=int, default=0,
                   help="rows of the corpus the CPU baseline scans; 0 (default) = ALL rows when the host has the memory (3 x the store "
                        "free), else the first 1M rows scaled")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                   help="experiments: wax_hip_set_tuning(KEY, VALUE) on every engine the bench creates (repeatable)")
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations")
    p.add_argument("--secondary", default="all",
                   help="comma-separated subset of the secondary configurations (N=1: " + ",".join(SECONDARY_N1) + "; N>1: s1m,s10k,c5); default all")
    p.add_argument("--c5-rows", type=int, default=10_000_000, help="rows of the config-5 corpus (tests shrink it)")
    p.add_argument("--chain-timed-region", action="store_true",
                   help="time the kernels INSIDE the timed region (scans chained, as in rounds 1-2): the run a rocprofv3 "
                        "--kernel-trace --stats summary is compared with — every launch of the region is one kernel alone")
    p.add_argument("--traffic", choices=["auto", "live", "replay", "off"], default="auto",
                   help="roofline.traffic of the headline kernel: live = a rocprofv3 --pmc FETCH_SIZE child pass inside this run (N = 1); "
                        "replay = the committed pass in profiles/latest_traffic.json; auto = live where rocprofv3 is on PATH, else replay")
    p.add_argument("--events", choices=["bound", "bracket"], default="bound",
                   help="per-launch kernel times: bound = HIP events bound to the dispatch (hipExtLaunchKernel pair: no trailing marker / chain wait in the interval, "
                        "default, with the bracketed figure beside it), bracket = hipEventRecord in front of and behind the launch only")
    p.add_argument("--batch-depth", type=int, default=0, help="batched secondaries: batches in flight (default 0 = auto: 3 for batches of up to 256 queries, 2 for larger ones)")
    p.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--traffic-child-batch", default=None, help=argparse.SUPPRESS)   # "nq:corpus:row_base": the batched form of the child
    p.add_argument("--detail-out", default=None, metavar="PATH",
                   help="where the verbose record goes (default bench_detail.json next to bench.py); stdout carries ONE compact line")
    p.add_argument("--host-merge", action="store_true", help="N>1: merge gathered hits on the host instead of the device")
    p.add_argument("--exchange", choices=["rccl", "host", "tickets"], default=None,
                   help="N>1, one rank per GPU (torchrun): rccl = all-gather device buffers over RCCL (default); host = download + gloo "
                        "all-gather (control path; also lets two test ranks share one GPU with WAX_BENCH_SAME_DEVICE=1). N>1 in ONE process "
                        "(the library's sharded handle): tickets = per-shard tickets + host merge, the library's default and the faster one "
                        "(default); rccl = one ncclAllGather per query on the library's communicator. Both are measured as secondaries "
                        "(h_tickets, h_rccl) in that shape whatever the headline uses")
    return p.parse_args()


TUNES = []


def apply_tunes(eng):
    for kv in TUNES:
        k, v = kv.split("=", 1)
        eng.setTuning(k, int(v))


# ---------------------------------------------------------------------------
# synthetic corpora, generated on the device that will hold them

def device_rows(torch, lo, hi, dims, dev):
    """Rows [lo, hi) of the primary corpus: unit-norm Gaussian, granule g seeded with CORPUS_SEED+g,
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


def clustered_rows(torch, lo, hi, dims, dev, centres=20, spread=0.3):
    """tools/fuzz_batch.py's "20 tight clusters": normalize(centre[c_i] + 0.3 * gaussian), c_i uniform — the kind of
    corpus a real embedding store is (topic clusters); rows of a cluster sit within a few bf16 error bands of each other."""
    g = torch.Generator(device=dev)
    g.manual_seed(CORPUS_SEED + 991)
    c = torch.randn((centres, dims), generator=g, device=dev, dtype=torch.float32)
    r = lo
    while r < hi:
        gi = r // GRANULE
        g0 = gi * GRANULE
        g.manual_seed(CORPUS_SEED + 100_000 + gi)
        noise = torch.randn((GRANULE, dims), generator=g, device=dev, dtype=torch.float32)
        which = torch.randint(0, centres, (GRANULE,), generator=g, device=dev)
        block = torch.nn.functional.normalize(c[which] + spread * noise, dim=1)
        a, b = r - g0, min(hi - g0, GRANULE)
        yield r, block[a:b].contiguous()
        r = g0 + b


def deterministic_embedder_rows(torch, lo, hi, dims, dev, prefix="doc-"):
    """The reference's DeterministicEmbedder (RAGBenchmarkSupport.swift:114-157) on the texts "doc-<i>", vectorised:
    seed = FNV-1a 64 of the UTF-8 text, a 64-bit LCG step per component, Float(Int64(bitPattern: state)) / Float(Int64.max),
    then L2 normalisation. int64 arithmetic wraps exactly like Swift's &* / &+ (checked against the oracle's C restatement
    in tests/test_host_cpu.py)."""
    def i64(x):  # a uint64 constant as the int64 with the same bits
        return x - (1 << 64) if x >= (1 << 63) else x
    fnv_prime, lcg_a, lcg_c = 1099511628211, i64(6364136223846793005), i64(1442695040888963407)
    h0 = 14695981039346656037
    for ch in prefix.encode("utf-8"):
        h0 = ((h0 ^ ch) * fnv_prime) & ((1 << 64) - 1)
    step = 1 << 18
    for r0 in range(lo, hi, step):
        r1 = min(hi, r0 + step)
        idx = torch.arange(r0, r1, device=dev, dtype=torch.int64)
        h = torch.full_like(idx, i64(h0))
        ndig = torch.ones_like(idx)
        for p in range(1, 19):
            ndig += (idx >= 10 ** p).to(torch.int64)
        for pos in range(19):                      # most significant digit first, like the decimal text
            active = pos < ndig
            div = torch.pow(torch.tensor(10, device=dev, dtype=torch.int64), torch.clamp(ndig - 1 - pos, min=0))
            digit = (idx // div) % 10 + 48
            h = torch.where(active, (h ^ digit) * fnv_prime, h)
        out = torch.empty((r1 - r0, dims), device=dev, dtype=torch.float32)
        state = h
        for j in range(dims):
            state = state * lcg_a + lcg_c
            out[:, j] = state.to(torch.float32) / float(2 ** 63)   # Float(Int64.max) rounds to 2^63
        yield r0, torch.nn.functional.normalize(out, dim=1).contiguous()


def duplicated_rows(torch, lo, hi, dims, dev):
    """The primary corpus with rows [DUP_AT, DUP_AT + DUP_ROWS) replaced by copies of row DUP_OF: exact ties that NO
    reduced-precision filter can order (and more of them than the widest re-score holds), so a query that aims at them can
    only be answered by the exact f32 path — the certificate refuses by construction."""
    src = None
    for _, x in device_rows(torch, DUP_OF, DUP_OF + 1, dims, dev):
        src = x[0].clone()
    for r0, x in device_rows(torch, lo, hi, dims, dev):
        a, b = max(r0, DUP_AT), min(r0 + x.shape[0], DUP_AT + DUP_ROWS)
        if a < b:
            x = x.clone()
            x[a - r0:b - r0] = src
        yield r0, x


CORPORA = {"gaussian": device_rows, "clustered": clustered_rows, "detembed": deterministic_embedder_rows, "dups": duplicated_rows}


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


def host_memory_free_bytes():
    """What this process may still allocate on the host: MemAvailable, capped by the cgroup's memory.max minus its current use."""
    free = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                free = int(line.split()[1]) * 1024
    except (OSError, ValueError):
        return 0
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            free = min(free, int(lim) - int(open("/sys/fs/cgroup/memory.current").read().strip()))
    except (OSError, ValueError):
        pass
    return max(0, free)


def cpu_baseline(torch, args, dev, queries):
    """The oracle's CPU scan (C restatement of the reference's arithmetic: a3 + a5) on a bounded sample of the same
    corpus, timed on this host as BASELINE.md §3 prescribes: (i) one thread, (ii) all host threads (static row
    partition, per-thread heap, merge; "all" = what the cgroup quota lets run at once). The inner loop is the metric-specialised FMA kernel (cosine: dot and |v|^2
    only), and the sample is first-touched by the threads that scan it (NUMA-local pages). Reported next to the GPU
    number; never the thing measured. Wax's actual CPU engine (USearch HNSW through Swift) cannot run here."""
    import oracle
    oracle.build()
    n_s = min(args.cpu_sample_rows, args.rows) if args.cpu_sample_rows > 0 else min(1_000_000, args.rows)
    if args.cpu_sample_rows <= 0 and host_memory_free_bytes() >= 3 * args.rows * args.dims * 4:
        n_s = args.rows                      # the whole workload on the host: no scaling (10M x 384: 15.4 GB; ~100 queries in the budget)
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
    for nthreads, budget, cap in ((1, args.cpu_baseline_seconds / 3.0, 40), (threads, args.cpu_baseline_seconds, 4000)):   # (a query that overruns its budget still completes)
        done, el = timed(nthreads, budget, cap)
        qps_sample = done / el
        variants.append({
            "threads": nthreads, "value": qps_sample * n_s / args.rows, "unit": "queries/s",
            "sample_qps": qps_sample, "sample_gbps": n_s * args.dims * 4 * qps_sample / 1e9,
            "sample": (f"{done} queries over all {n_s} rows in {el:.1f} s" if n_s == args.rows else
                       f"{done} queries over the first {n_s} rows in {el:.1f} s"),
        })
    best = variants[-1]
    return {
        "value": best["value"], "unit": "queries/s", "cores": threads, "kind": "port",
        "sample_short": (f"{best['sample']}; {best['sample_gbps']:.0f} GB/s" if n_s == args.rows else
                         f"{best['sample']}, x{n_s}/{args.rows} rows; {best['sample_gbps']:.0f} GB/s"),
        "sample": f"{best['sample']} of the same corpus on {threads} threads ({best['sample_qps']:.2f} q/s on the sample = "
                  f"{best['sample_gbps']:.1f} GB/s)" + ("" if n_s == args.rows else f", scaled by {n_s}/{args.rows} rows to the full workload")
                  + "; metric-specialised FMA inner loop, NUMA first-touch by the scanning threads",
        "variants": variants,
    }


def cpu_baseline_small(torch, dev, rows, dims, topk, queries, seconds=3.0):
    """`cpu_baseline` for a secondary single-query size (the 10K and 1M points of the north star's N matrix): the same oracle scan over
    ALL `rows` rows on one thread and on the usable host threads, `seconds` of budget in total. Compact: value, cores, the 1-thread value
    and what the sample was."""
    import types
    a = types.SimpleNamespace(rows=rows, dims=dims, topk=topk, cpu_sample_rows=rows, cpu_baseline_seconds=seconds)
    full = cpu_baseline(torch, a, dev, queries)
    v1 = next((v for v in full["variants"] if v["threads"] == 1), None)
    return {"value": full["value"], "unit": "queries/s", "cores": full["cores"], "kind": "port",
            "value_1_thread": v1["value"] if v1 else None, "sample": full["sample_short"]}


# ---------------------------------------------------------------------------
# the single-query measurement: product-mode timed region + per-launch calibration pass

CALIBRATION_STEPS = 60


def run_pipelined(submit, collect, qs, depth):
    pending = []
    last = None
    for q in qs:
        if len(pending) >= depth:
            last = collect(pending.pop(0))
        pending.append(submit(q))
    while pending:
        last = collect(pending.pop(0))
    return last


def measure_single_query(eng, submit, collect, queries, warmup, steps, depth, barrier, chain_timed_region=False, floor_ms=0.0):
    """warm-up, then EXACTLY `steps` timed steps bracketed by barrier(); then the per-launch calibration pass.
    Returns (elapsed_s, last_result, kernel_avg_ms, launches_timed, calibration dict)."""
    eng.setTuning("time_kernels", 1 if chain_timed_region else 0)
    run_pipelined(submit, collect, queries[:warmup], depth)
    eng.setTuning("reset_stats", 1)
    barrier()
    t0 = time.perf_counter()
    last = run_pipelined(submit, collect, queries[warmup:warmup + steps], depth)
    barrier()
    elapsed = time.perf_counter() - t0
    if chain_timed_region:
        st = eng.stats()
        launches = int(st.scan_kernels_timed)
        kern_ms = st.scan_kernel_ms_total / launches if launches else float("nan")
        return elapsed, last, kern_ms, launches, {"steps": steps, "ms_per_step": elapsed / steps * 1e3,
                                                  "mode": "the timed region itself (--chain-timed-region: kernels timed and chained inside it)"}
    # calibration: the same queries again with HIP events on every scan launch, on the stream the kernel runs on, scans chained
    # through an event so that a launch runs alone. Two passes:
    #   "time_kernels" = 1 (bracketed): hipEventRecord in front of and behind the launch — the interval holds the kernel AND the
    #       packets around it (the two markers, the chain wait, the dispatch latency: ~10 us whatever the kernel's length);
    #   "time_kernels" = 2 (kernel-bound, the figure `frac` uses): hipExtLaunchKernel binds the event pair to the dispatch itself —
    #       from the marker the runtime puts in front of the dispatch to the dispatch's completion: no trailing marker, no chain wait.
    #       Against rocprofv3's own duration of the SAME dispatch (profiles/r05/k_event_modes_vs_rocprofv3_same_launches.txt) it is
    #       still 4-8 us long (the dispatch latency); the bracket is 6-9 us long on a blocking call, 15-19 us in this chained pass.
    # A kernel-bound mean that is not plausible next to the bracketed one (runtime without the binding) falls back to the bracket.
    n_cal = min(steps, CALIBRATION_STEPS)

    def cal_pass(mode):
        eng.setTuning("time_kernels", mode)
        run_pipelined(submit, collect, queries[warmup:warmup + min(4, n_cal)], depth)
        eng.setTuning("reset_stats", 1)
        barrier()
        t1 = time.perf_counter()
        run_pipelined(submit, collect, queries[warmup:warmup + n_cal], depth)
        barrier()
        el = time.perf_counter() - t1
        st = eng.stats()
        n_ = int(st.scan_kernels_timed)
        return (st.scan_kernel_ms_total / n_ if n_ else float("nan")), n_, el

    br_ms, br_n, br_el = cal_pass(1)
    kern_ms, launches, cal_el, events = br_ms, br_n, br_el, "bracketed"
    kb_ms = None
    if EVENT_MODE == "bound":
        kb_ms, kb_n, kb_el = cal_pass(2)
        if kb_n and br_n and kernel_bound_plausible(kb_ms, br_ms, floor_ms):
            kern_ms, launches, cal_el, events = kb_ms, kb_n, kb_el, "kernel-bound"
    eng.setTuning("time_kernels", 0)
    return elapsed, last, kern_ms, launches, {
        "steps": n_cal, "ms_per_step": cal_el / n_cal * 1e3, "events": events,
        "kernel_avg_ms_bracketed": br_ms, "kernel_avg_ms_kernel_bound": kb_ms,
        "mode": "same run, same engine, the first queries of the timed region again, scans chained (never two at once), HIP events on "
                "the scan's own stream: " + ("bound to the scan's dispatch (hipExtLaunchKernel start / stop pair: from the marker the runtime puts in front of "
                                             "the dispatch to the dispatch's completion; 4-8 us above rocprofv3's duration of the same dispatch, "
                                             "profiles/r05/k_event_modes_vs_rocprofv3_same_launches.txt); the bracketed mean (hipEventRecord around the "
                                             "launch: kernel + both markers + the chain wait) is kernel_avg_ms_bracketed"
                                             if events == "kernel-bound" else "recorded in front of and behind the launch (\"time_kernels\" = 1)")}


RCCL_FAILURE = [None]    # why RCCL was abandoned in this run (None = not abandoned)
BATCH_DEPTH = 0          # --batch-depth: batches in flight of the batched secondaries (0 = auto: 3 up to 256 queries per batch, else 2)
EVENT_MODE = "bound"     # --events: "bound" (default) = frac from kernel-bound HIP events, "bracket" = rounds 1-4's hipEventRecord bracket


def kernel_bound_plausible(kb_ms, br_ms, floor_ms=0.0):
    """A kernel-bound interval is the bracketed one minus the packets around the kernel: never longer than it (2 % of jitter allowed),
    not shorter by more than 40 us + 10 % (the packets cost ~10 us) — unless it still is at least `floor_ms`, the time the launch's
    algorithmic bytes take at the peak rate (an unbound pair reports next to nothing, not a physically possible scan: the k > 192
    queries' bracket carries 0.5 ms that neither rocprofv3's duration of the same dispatch nor the pipelined rate shows) —, never below a
    quarter of the bracket (short kernels: 0.9 x - 40 us is negative under 44 us, which used to accept ANY positive interval —
    advisor, round 5) and never below `floor_ms`. Anything else is a runtime that did not bind the pair."""
    near_bracket = kb_ms >= br_ms * 0.9 - 0.04 or (floor_ms > 0.0 and kb_ms >= floor_ms)
    return (kb_ms == kb_ms and br_ms == br_ms and kb_ms > 0 and kb_ms <= br_ms * 1.02 + 0.002 and near_bracket
            and kb_ms >= 0.25 * br_ms and kb_ms >= floor_ms)


TRAFFIC_CHILD_WARM, TRAFFIC_CHILD_LAUNCHES = 2, 6


def traffic_child(torch, dev, rows, dims, k):
    """`--traffic-child` (run under rocprofv3 --pmc by live_traffic): the headline corpus, a few blocking single-query scans, nothing else."""
    eng = _load_engine(torch, dev, rows, dims)
    apply_tunes(eng)
    torch.cuda.synchronize()
    for q in unit_queries(TRAFFIC_CHILD_WARM + TRAFFIC_CHILD_LAUNCHES, dims):
        eng.searchArrays(q, k)
    eng.close()


def traffic_child_batch(torch, dev, rows, dims, k, spec):
    """`--traffic-child --traffic-child-batch nq:corpus:row_base`: the batched workload, a mirror build + one warm batch + three counted ones."""
    nq, corpus, row_base = spec.split(":")
    nq = int(nq)
    eng = _load_engine(torch, dev, rows, dims, corpus)
    eng.setRowBase(int(row_base))
    apply_tunes(eng)
    dq = batch_queries(torch, dev, nq, dims, corpus)
    out = torch.empty((nq, k, 2), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2 + 3):
        eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, out.data_ptr(), k, stream)
    torch.cuda.synchronize()
    eng.close()


TRAFFIC_MODE = "replay"      # set by main(): what secondary_batched may do for its roofline.traffic


def fetch_size_per_launch(out_dir, batched):
    """FETCH_SIZE [KiB] of the counted launches in a rocprofv3 `--pmc FETCH_SIZE --output-format csv` directory, in dispatch order,
    warm-ups dropped. Single-query child: the scan kernel's launches. Batched child: the `batch_gemm_*` instantiation with the largest
    mean — the filtering GEMM streams the whole mirror, the sampling launch of the same template reads a few hundred tiles."""
    import csv
    import glob
    vals, by_name = [], {}
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name") or ""
                if r.get("Counter_Name") != "FETCH_SIZE":
                    continue
                ent = (int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"]))
                if not batched and "scan_kernel" in name:
                    vals.append(ent)
                elif batched and "batch_gemm" in name:
                    by_name.setdefault(name, []).append(ent)
    if batched and by_name:
        vals = max(by_name.values(), key=lambda v: sum(x for _, x in v) / len(v))
    return [v for _, v in sorted(vals)][TRAFFIC_CHILD_WARM:]


# The counter passes are an optional leg of a run that must finish within minutes whatever the box does: each child has its own limit, all
# of them together a budget, and the first pass that times out switches the rest of the run to the replayed figures.
LIVE_CHILD_LIMIT_S, LIVE_TOTAL_BUDGET_S = 150.0, 300.0
LIVE_STATE = {"disabled": None, "spent_s": 0.0, "passes": 0}


def live_allowed():
    """None if another counter pass may start, else the reason why not."""
    if LIVE_STATE["disabled"]:
        return "counter passes disabled for the rest of this run: " + LIVE_STATE["disabled"]
    if LIVE_STATE["spent_s"] >= LIVE_TOTAL_BUDGET_S:
        return f"counter-pass budget of {LIVE_TOTAL_BUDGET_S:.0f} s spent ({LIVE_STATE['passes']} passes, {LIVE_STATE['spent_s']:.0f} s)"
    return None


def live_account(seconds, timed_out=False):
    LIVE_STATE["spent_s"] += seconds
    LIVE_STATE["passes"] += 1
    if timed_out:
        LIVE_STATE["disabled"] = f"a pass hit its {LIVE_CHILD_LIMIT_S:.0f} s limit"


def live_traffic(rows, dims, k, timeout_s=None, batch=None):
    """HBM bytes per launch of the headline scan kernel, measured IN this run: a counters-only child pass
    (`rocprofv3 --pmc FETCH_SIZE --kernel-trace`, no other tracing: MI355X_MICROARCH.md's HBM recipe) over the same corpus
    and kernel, TRAFFIC_CHILD_LAUNCHES launches after TRAFFIC_CHILD_WARM warm-ups. bytes = FETCH_SIZE (KiB) * 1024 * 2 (the guide's
    gfx950 correction: the counter sees a 128-B request of a wide coalesced stream as 64 B). Returns (bytes or None, source text)."""
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(v.startswith(("ROCPROF", "ROCP_")) for v in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under rocprofv3: no nested counter pass"
    why_not = live_allowed()
    if why_not:
        return None, why_not
    if timeout_s is None:
        timeout_s = min(LIVE_CHILD_LIMIT_S, max(30.0, LIVE_TOTAL_BUDGET_S - LIVE_STATE["spent_s"]))
    tmp = tempfile.mkdtemp(prefix="wax_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--",
           sys.executable, os.path.abspath(__file__), "--traffic-child", "--rows", str(rows), "--dims", str(dims), "--topk", str(k)]
    if batch is not None:
        cmd += ["--traffic-child-batch", batch]
    for t in TUNES:
        cmd += ["--tune", t]
    env = dict(os.environ, TMPDIR="/tmp")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(v, None)
    t0 = time.perf_counter()
    try:
        with open(os.path.join(tmp, "child.log"), "w") as lf:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=lf, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)      # exactly the process group started above
                proc.wait()
                live_account(time.perf_counter() - t0, timed_out=True)
                return None, f"counter pass timed out after {timeout_s:.0f} s"
        live_account(time.perf_counter() - t0)
        vals = fetch_size_per_launch(tmp, batch is not None)
        if rc != 0 or not vals:
            tail = ""
            try:
                tail = open(os.path.join(tmp, "child.log")).read()[-300:].replace("\n", " | ")
            except OSError:
                pass
            return None, f"counter pass failed (rc {rc}, {len(vals)} launches seen): {tail}"
        mean_kib = sum(vals) / len(vals)
        return mean_kib * 1024.0 * 2.0, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace child pass (counters only) over the same "
                                         f"corpus{' and query block' if batch else ''}, {len(vals)} launches after {TRAFFIC_CHILD_WARM} warm-ups, {time.perf_counter() - t0:.0f} s; "
                                         f"bytes = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 correction); min/max over launches "
                                         f"{min(vals) * 2048:.6g} / {max(vals) * 2048:.6g}")
    except Exception as ex:  # noqa: BLE001 - the bench line must not die on its optional leg
        return None, f"counter pass raised {type(ex).__name__}: {ex}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan_roofline(bytes_per_launch, kern_ms, launches, elapsed, steps, cal, traffic=None, traffic_source=None):
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if launches else float("nan")
    pipeline = bytes_per_launch * steps / elapsed / 1e9
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS if launches else None,
        "pipeline_achieved": pipeline, "pipeline_frac": pipeline / HBM_PEAK_GBPS,
        "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "wax::scan_kernel (fused scan + per-wave top-k)",
        "kernel_avg_ms": kern_ms, "kernel_launches_timed": launches,
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "calibration": cal,
        "note": "frac = rows_per_gpu*dims*4 bytes per launch / mean HIP-event duration of the scan kernel, per launch, from the "
                "calibration pass of this run (kernels chained: one at a time; calibration.events says which event pair); pipeline_frac = the same bytes x steps / the "
                "timed region's elapsed time (rank 0's shard), i.e. what the overlapped product pipeline sustains end to end",
    }


# ---------------------------------------------------------------------------
# secondary configurations: same bracket as the headline, each with its own roofline block

def _bracket(torch):
    torch.cuda.synchronize()


def _load_engine(torch, dev, rows, dims, corpus="gaussian", devices=None, lo=0):
    from wax_amd import HIPVectorEngine, VectorMetric
    if devices is None:
        eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims)
    else:
        eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims, devices=devices)
    eng.reserve(rows)
    for r0, x in CORPORA[corpus](torch, lo, lo + rows, dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    return eng


def secondary_single_query(torch, dev, rows, dims, k, steps, warmup, depth, label="BASELINE config 2", cpu_seconds=0.0):
    """BASELINE config 2 (and the 10K-row point of the north star's N matrix): rows x dims f32, one query per step, the
    headline's code path at another size. cpu_seconds > 0: the oracle's CPU scan of the same rows beside it (north star: "next to
    Wax's own CPU scan ... in the same run")."""
    eng = _load_engine(torch, dev, rows, dims)
    queries = unit_queries(warmup + steps, dims)
    eng.setTuning("streams", 2)
    eng.setTuning("slots", max(depth, 2))
    apply_tunes(eng)
    elapsed, last, kern_ms, launches, cal = measure_single_query(
        eng, lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), queries, warmup, steps, depth, lambda: _bracket(torch),
        floor_ms=rows * dims * 4 / (HBM_PEAK_GBPS * 1e9) * 1e3)
    import hashlib
    checksum = hashlib.sha256(np.asarray(last[0], dtype=np.uint64).tobytes() + np.asarray(last[1], dtype=np.float32).tobytes()).hexdigest()[:16]
    nbytes = rows * dims * 4
    grid = eng.getTuning("scan_grid")
    merged, overlapped = eng.getTuning("merged_scans"), eng.getTuning("overlap_scans")
    short_stats = (int(eng.getTuning("short_selects")), int(eng.getTuning("short_select_failures")))
    eng.close()
    traffic, traffic_source = None, None
    if TRAFFIC_MODE in ("auto", "live"):
        traffic, traffic_source = live_traffic(rows, dims, k)      # (a store that lives in the L2s reads far less than its size from HBM)
        if traffic is None:
            log(f"[bench] live traffic pass unavailable ({rows}x{dims}): {traffic_source}")
            traffic_source = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json"))).get("single", {})
        ent = tj.get("configs", {}).get(f"{rows}x{dims}")
        if ent and traffic is None and TRAFFIC_MODE in ("auto", "replay"):
            traffic = ent["hbm_bytes_per_launch"]
            traffic_source = "replayed from profiles/latest_traffic.json (" + tj.get("source", "") + "), not measured in this run"
    except (OSError, ValueError, KeyError):
        pass
    rf = scan_roofline(nbytes, kern_ms, launches, elapsed, steps, cal, traffic, traffic_source)
    if k > 192:
        rf["note_general_selection"] = ("top_k > 192: the fused scan leaves every workgroup's 192 best and one workgroup selects and certifies the "
                                        "top_k among them (DESIGN 4.2); the distance pass + radix selection behind it return at once unless the "
                                        "certificate fails; kernel_avg_ms is the scan kernel, ms_per_step the whole query")
        rf["short_selects"], rf["short_select_failures"] = short_stats
    rf["scan_grid"] = grid
    # 1: the scan kernel's last-arriving workgroup did the final merge; 2: a merge launch behind the scan (stores beyond 2 GiB; and, with
    # other scans in flight as here, stores from "merge_overlap_mb" up: the merge overlaps the next scan)
    rf["launches_per_query"] = 1 if merged > overlapped else 2
    res = {
        "config": f"{rows} x {dims} f32 cosine top-{k}, one query per step, 1 GPU ({label})",
        "value": steps / elapsed, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "dtype": "f32", "last_result_checksum": checksum, "roofline": rf, "top_k": k,
    }
    if cpu_seconds > 0:
        try:
            res["cpu_baseline"] = cpu_baseline_small(torch, dev, rows, dims, k, queries, cpu_seconds)
        except Exception as ex:  # noqa: BLE001 — the baseline beside a secondary must not lose the secondary
            res["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"}
    return res


def _load_handle(torch, devs, rows, dims):
    """A sharded handle (one process, the library's multi-GPU engine) holding `rows` rows of the bench corpus: the library
    chooses the block layout (even spread, but never less than "shard_min_mb" of rows per block: a small store stays on the
    first device); every granule is generated on the device whose shard takes it."""
    from wax_amd import HIPVectorEngine, VectorMetric
    eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims, devices=devs)
    eng.reserve(rows)
    per = int(eng.getTuning("block_rows"))
    r = 0
    while r < rows:
        g_ = min(r // per, len(devs) - 1)
        gdev = torch.device("cuda", devs[g_])
        r_hi = min(rows, (r // GRANULE + 1) * GRANULE, (g_ + 1) * per if g_ + 1 < len(devs) else rows)
        for r0, x in device_rows(torch, r, r_hi, dims, gdev):
            eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
        r = r_hi
    for d_ in set(devs):
        torch.cuda.synchronize(d_)
    return eng


def handle_preflight(torch, devs):
    """Untimed first-contact check of the one-process multi-GPU shape (`python bench.py --gpus N`), before anything is measured: peer
    access pair by pair as the handle found it, a small store FORCED over every device ("shard_min_mb" 0) answering exactly like one
    engine on device 0, and one engine per non-zero ordinal answering on its own device. Never fatal: what fails is printed and
    carried in config.preflight, and the run goes on (per-shard tickets need no peer access at all)."""
    from wax_amd import HIPVectorEngine, VectorMetric
    info = {"devices": list(devs)}
    try:
        rows, dims, k = 32768, 64, 10
        g = torch.Generator(device="cpu").manual_seed(1234)
        x = torch.randn((rows, dims), generator=g, dtype=torch.float32)
        x = (x / x.norm(dim=1, keepdim=True)).numpy()
        ids = np.arange(rows, dtype=np.uint64)
        qs = x[[5, 4096, 20000, rows - 1]] * np.float32(1.0)
        one = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims, device=devs[0])
        one.addBatch(ids, x)
        ref = [one.searchArrays(q, k) for q in qs]
        one.close()
        h = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims, devices=list(devs))
        h.setTuning("shard_min_mb", 0)
        h.reserve(rows)
        h.addBatch(ids, x)
        info["peer_pairs"], info["peer_enabled"] = int(h.getTuning("peer_pairs")), int(h.getTuning("peer_enabled"))
        info["rows_per_device"] = [int(h.shardInfo(i)[2]) for i in range(len(devs))]
        got = [h.searchArrays(q, k) for q in qs]
        info["spread_equals_one_engine"] = all(list(a[0]) == list(b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(got, ref))
        h.close()
        per = []
        for d in sorted(set(devs) - {devs[0]}):
            e = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims, device=d)
            e.addBatch(ids[:4096], x[:4096])
            r = e.searchArrays(x[7], k)
            per.append(bool(int(r[0][0]) == 7 and int(e.device) == d))
            e.close()
        info["engine_on_each_nonzero_ordinal"] = all(per) if per else None
        info["ok"] = bool(info["spread_equals_one_engine"] and (info["engine_on_each_nonzero_ordinal"] in (True, None)))
    except Exception as ex:  # noqa: BLE001
        info["ok"] = False
        info["error"] = f"{type(ex).__name__}: {ex}".replace("\n", " ")[:200]
    log(f"[bench] preflight (one process, {len(devs)} devices): {json.dumps(info)}")
    return info


def handle_exchange(eng, args, n_devices, want_rccl=None):
    """One-process shape: which exchange the handle uses. `--exchange rccl` (or want_rccl) asks for one ncclAllGather per query on the
    library's single-process communicator; if the library cannot load RCCL, or its communicator does not span the N devices, the run
    DEGRADES to per-shard tickets (the library's default exchange), says so (RCCL_FAILURE -> config.exchange, rccl_ranks = 0) and
    goes on: the first contact with a multi-GPU node must not end without a line (round 5 raised SystemExit here)."""
    want = (args.exchange == "rccl") if want_rccl is None else want_rccl
    if want and not os.environ.get("WAX_BENCH_SAME_DEVICE"):
        try:
            if os.environ.get("WAX_BENCH_FAKE_RCCL_FAILURE"):
                raise RuntimeError("WAX_BENCH_FAKE_RCCL_FAILURE")
            eng.setTuning("exchange", 1)
            ranks = int(eng.getTuning("rccl_ranks"))
            if ranks != n_devices:
                raise RuntimeError(f"the library's communicator has {ranks} ranks for {n_devices} devices")
            return 1, ranks
        except Exception as ex:  # noqa: BLE001
            RCCL_FAILURE[0] = f"{type(ex).__name__}: {ex}".replace("\n", " ")[:160]
            log(f"[bench] one-process RCCL exchange unavailable ({RCCL_FAILURE[0]}): per-shard tickets + host merge instead")
            try:
                eng.setTuning("exchange", 0)
                eng.setTuning("ticket_path", 1)
            except Exception:  # noqa: BLE001
                pass
    return 0, 0


def handle_single_query(torch, args, devs, rows, dims, k, steps, warmup, label, want_rccl=None):
    """The N-matrix points (N in {10K, 1M} x 384) in the ONE-PROCESS shape (`python bench.py --gpus N`): the same sharded handle
    as the headline at another corpus size. A store below the small-store threshold is not spread (rows_per_gpu shows it)."""
    eng = _load_handle(torch, devs, rows, dims)
    apply_tunes(eng)
    exchange_mode, rccl_ranks = handle_exchange(eng, args, len(devs), want_rccl)
    eng.setTuning("streams", 2)
    eng.setTuning("slots", max(args.depth, 2))
    queries = unit_queries(warmup + steps, dims)
    per_gpu = [int(eng.shardInfo(g)[2]) for g in range(len(devs))]
    elapsed, last, kern_ms, launches, cal = measure_single_query(
        eng, lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), queries, warmup, steps, args.depth, lambda: _bracket(torch))
    import hashlib
    checksum = hashlib.sha256(np.asarray(last[0], dtype=np.uint64).tobytes() + np.asarray(last[1], dtype=np.float32).tobytes()).hexdigest()[:16]
    grid = eng.getTuning("scan_grid")
    tickets, single = int(eng.getTuning("ticket_searches")), int(eng.getTuning("single_shard_searches"))
    eng.close()
    rf = scan_roofline(max(per_gpu) * dims * 4, kern_ms, launches, elapsed, steps, cal)
    rf["scan_grid"] = grid
    return {
        "config": f"{rows} x {dims} f32 cosine top-{k}, one query per step, ONE process, sharded handle over {len(devs)} device(s), "
                  f"rows per device {per_gpu} ({label})",
        "value": steps / elapsed, "unit": "queries/s", "n_gpus": len(devs), "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "dtype": "f32", "rows_per_gpu": per_gpu, "last_result_checksum": checksum,
        "exchange": "rccl" if exchange_mode == 1 else ("one device" if single else "per-shard tickets, host merge"),
        "rccl_ranks": rccl_ranks, "ticket_searches": tickets, "single_shard_searches": single,
        "roofline": rf,
    }


def init_distributed(torch, dist, rank, world, dev, use_rccl):
    """The default process group of the one-rank-per-GPU shape. Returns whether RCCL is the exchange.
    use_rccl: ONE group with both backends — CPU tensors travel through gloo, CUDA tensors through RCCL (created lazily, at the probe
    below). If RCCL cannot talk across the node and SAYS so (an exception on any rank, at group creation or at the probe), every rank
    learns it through gloo and the run goes on with the host exchange, labelled (RCCL_FAILURE -> config.exchange): the first contact
    with an 8-GPU node then still prints its line. Nothing is torn down (tearing down a half-initialised RCCL group
    hangs): the dead CUDA backend is simply never used again. A probe that HANGS is ended by the group's timeout, as before."""
    import datetime
    if not use_rccl:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return False
    ok, why = 1, ""
    try:
        dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    except Exception as ex:  # noqa: BLE001 — no usable NCCL backend at all (every rank sees the same): plain gloo
        RCCL_FAILURE[0] = f"{type(ex).__name__}: {ex}".replace("\n", " ")[:160]
        log(f"[bench] rank {rank}: no RCCL backend ({RCCL_FAILURE[0]}): host (gloo) exchange")
        if dist.is_initialized():
            dist.destroy_process_group()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return False
    if os.environ.get("WAX_BENCH_FAKE_RCCL_FAILURE"):            # tests: pretend the collective library refused
        ok, why = 0, "WAX_BENCH_FAKE_RCCL_FAILURE"
    else:
        try:
            probe_t = torch.ones(1, device=dev)
            dist.all_reduce(probe_t)
            torch.cuda.synchronize()
            if int(probe_t.item()) != world:
                ok, why = 0, f"probe all-reduce returned {int(probe_t.item())}, expected {world}"
        except Exception as ex:  # noqa: BLE001
            ok, why = 0, f"{type(ex).__name__}: {ex}"
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                   # gloo
    if int(flag.item()) == 0:
        RCCL_FAILURE[0] = (why or "another rank's probe failed").replace("\n", " ")[:160]
        log(f"[bench] rank {rank}: RCCL is not usable ({RCCL_FAILURE[0]}): continuing with the host (gloo) exchange")
        return False
    return True


def dist_barrier(torch, dist, use_rccl, device_index):
    """Barrier over the default group. RCCL: the device barrier. Host exchange: an all-reduce of a CPU tensor — in a pure gloo group
    and in the mixed "cpu:gloo,cuda:nccl" group alike it travels through gloo (dist.barrier() on the mixed group would pick the CUDA
    backend, i.e. the very RCCL that the fall-back exists to avoid)."""
    if use_rccl:
        dist.barrier(device_ids=[device_index])
    else:
        dist.all_reduce(torch.zeros(1))


SMALL_STORE_BYTES = 64 << 20     # = the library's "shard_min_mb" default: a store below it is not spread over GPUs


def small_store(rows, dims, world):
    """The small-store rule in the one-rank-per-GPU launch shape: below SMALL_STORE_BYTES the whole store lives on rank 0's GPU and
    rank 0 answers alone (no exchange); the other ranks hold an empty engine and only meet the barriers. Sharding a 15 MB store over
    N GPUs buys N launches and a collective per query for 2 MB of scan each (round-4 rehearsal: 7 882 q/s at 2 ranks against 94 843 at 1)."""
    return world > 1 and rows * dims * 4 < SMALL_STORE_BYTES


def sharded_single_query(torch, dist, args, rank, world, local_rank, use_rccl, rows, dims, k, steps, warmup, label):
    """The N-matrix points (N in {10K, 1M} x 384) at world > 1: the headline's sharded single-query path — every rank scans
    its row shard, per-shard top-k all-gathered (RCCL) and merged per query — at another corpus size. Same bracket as the
    headline (barrier + synchronize on both sides, max over ranks). Called by EVERY rank (collectives inside)."""
    from wax_amd import sharded
    dev = torch.device("cuda", torch.cuda.current_device())
    solo = small_store(rows, dims, world)
    lo, hi = ((0, rows) if rank == 0 else (0, 0)) if solo else sharded.shard_bounds(rows, world, rank, align=64)
    eng = _load_engine(torch, dev, max(hi - lo, 0), dims, lo=lo)
    eng.setRowBase(lo)
    apply_tunes(eng)
    queries = unit_queries(warmup + steps, dims)

    def barrier():
        torch.cuda.synchronize()
        dist_barrier(torch, dist, use_rccl, local_rank)
        torch.cuda.synchronize()

    if solo:
        eng.setTuning("streams", 2)
        eng.setTuning("slots", max(args.depth, 2))
        elapsed, last, kern_ms, launches, cal = measure_single_query(eng, lambda q: eng.submit(q, k), lambda t: eng.collect(t, k), queries,
                                                                     warmup, steps, args.depth, barrier)
    else:
        searcher = sharded.ShardedSearcher(eng, rank, world, k, depth=args.depth, n_streams=2, host_merge=args.host_merge,
                                           exchange="rccl" if use_rccl else "host")

        def submit(q):
            searcher.submit(q)

        elapsed, last, kern_ms, launches, cal = measure_single_query(eng, submit, lambda _: searcher.collect(), queries, warmup, steps,
                                                                     args.depth, barrier)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if use_rccl else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    import hashlib
    checksum = hashlib.sha256(np.asarray(last[0], dtype=np.uint64).tobytes() + np.asarray(last[1], dtype=np.float32).tobytes()).hexdigest()[:16]
    grid = eng.getTuning("scan_grid")
    eng.close()
    rf = scan_roofline(max(hi - lo, 1) * dims * 4, kern_ms, launches, elapsed, steps, cal)
    rf["scan_grid"] = grid
    return {
        "config": (f"{rows} x {dims} f32 cosine top-{k}, one query per step, {world} ranks, the store on rank 0's GPU alone "
                   f"(small-store rule: below {SMALL_STORE_BYTES >> 20} MB nothing is sharded, no exchange) ({label})" if solo else
                   f"{rows} x {dims} f32 cosine top-{k}, one query per step, row-sharded over {world} GPUs ({label})"),
        "value": steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "dtype": "f32", "rows_per_gpu": ([rows] + [0] * (world - 1)) if solo else hi - lo,
        "last_result_checksum": checksum,
        "exchange": "none (small-store rule)" if solo else ("rccl" if use_rccl else "host" + (" (rccl failed)" if RCCL_FAILURE[0] else "")),
        "roofline": rf,
    }


def batched_roofline(rows, dims, nq, kern_ms, launches, rega=5, live=None):
    flops = 2.0 * nq * rows * dims
    nbytes = rows * dims * 2                  # bf16 mirror, streamed once per launch
    t_hbm, t_mfma = nbytes / (HBM_PEAK_GBPS * 1e9), flops / (MFMA_BF16_PEAK_TFLOPS * 1e12)
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    if bound == "hbm":
        achieved, peak, unit = nbytes / (kern_ms * 1e-3) / 1e9, HBM_PEAK_GBPS, "GB/s"
    else:
        achieved, peak, unit = flops / (kern_ms * 1e-3) / 1e12, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s"
    traffic, traffic_source = None, None
    live_note = None
    if live is not None and TRAFFIC_MODE in ("auto", "live"):
        traffic, live_note = live_traffic(rows, dims, live["k"], batch=f"{nq}:{live['corpus']}:{live['row_base']}")
        traffic_source = live_note if traffic is not None else None
        if traffic is None:
            log(f"[bench] live traffic pass unavailable ({rows}x{dims}xq{nq}): {live_note}")
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json"))).get("batched", {})
        ent = tj.get("configs", {}).get(f"{rows}x{dims}xq{nq}")
        if ent and traffic is None and TRAFFIC_MODE in ("auto", "replay"):
            traffic = ent["hbm_bytes_per_launch"]
            traffic_source = "replayed from profiles/latest_traffic.json (" + tj.get("source", "") + "), not measured in this run"
    except (OSError, ValueError, KeyError):
        pass
    # the register-resident-queries GEMM where it exists ("batch_rega" 0 = the LDS-tiled kernel everywhere)
    kernel = "wax::batch_gemm_rq_kernel" if (dims in (128, 256, 384, 512, 768) and rega != 0) else "wax::batch_gemm_kernel"
    # what the kernel's K loop alone (no HBM stream, no selection) sustains on this part with embedding-like operands: the matrix
    # clock is power-limited (tools/mfma_probe.hip; DESIGN.md "The matrix roof"). Informational: `peak` stays the nominal figure.
    sustained = None
    try:
        for line in open(os.path.join(ROOT, "profiles", "r03", "q_mfma_k_loop_probe.jsonl")):
            pj = json.loads(line)
            if (pj.get("data") == "gaussian embedding" and pj.get("mode") == "reads+mfma" and pj.get("waves_per_simd") == 2
                    and pj.get("ahead") == 3 and not pj.get("tile_fence")):
                sustained = pj["tflops_bf16"]
    except (OSError, ValueError, KeyError):
        pass
    extra = {}
    if sustained:
        extra = {"mfma_sustained_tflops_k_loop_alone": sustained,
                 "mfma_frac_of_sustained": flops / (kern_ms * 1e-3) / 1e12 / sustained,
                 "mfma_sustained_source": "profiles/r03/q_mfma_k_loop_probe.jsonl (tools/mfma_probe.hip: the kernel's K loop alone on all 256 CUs, "
                                          "N(0, 1/384) bf16 operands, 2 waves per SIMD) — committed measurement, not taken in this run"}
    return flops, max(t_hbm, t_mfma), {
        **extra,
        "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
        "kernel": kernel, "kernel_avg_ms": kern_ms, "kernel_launches_timed": launches,
        "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
        "hbm_floor_ms": t_hbm * 1e3, "mfma_floor_ms": t_mfma * 1e3,
        "traffic": traffic, "traffic_source": traffic_source}


def batch_queries(torch, dev, nq, dims, corpus, eng_rows=None):
    """Query block in HBM. gaussian / detembed corpora: unit Gaussian queries. clustered: the fuzz tool's mix — half of
    the queries sit inside a cluster (a stored row + 0.05 gaussian, not normalised), half are plain Gaussian."""
    q = torch.from_numpy(unit_queries(nq, dims)).to(dev)
    if corpus == "clustered":
        g = torch.Generator(device=dev)
        g.manual_seed(QUERY_SEED + 5)
        raw = torch.randn((nq, dims), generator=g, device=dev, dtype=torch.float32)
        rows = []
        for _, x in clustered_rows(torch, 0, GRANULE, dims, dev):
            rows.append(x)
        x0 = torch.cat(rows)[: nq // 2]
        raw[: nq // 2] = x0 + 0.05 * raw[: nq // 2]
        q = raw.contiguous()
    if corpus == "dups":
        g = torch.Generator(device=dev)
        g.manual_seed(QUERY_SEED + 9)
        for _, x in device_rows(torch, DUP_OF, DUP_OF + 1, dims, dev):
            m = min(DUP_QUERIES, nq)
            q[:m] = torch.nn.functional.normalize(x[0][None, :] + 0.02 * torch.randn((m, dims), generator=g, device=dev), dim=1)
    return q


def secondary_batched(torch, dev, rows, dims, nq, k, steps, warmup, label, row_base=0, corpus="gaussian", live_traffic=True):
    """BASELINE configs 3 / 5: nq queries per step as a bf16 MFMA GEMM + fused top-k + exact f32 re-score. Queries and
    results stay in HBM (wax_hip_search_batch_submit_device / _collect_device): the timed region holds no host<->device
    traffic except nq certificate flags per step."""
    eng = _load_engine(torch, dev, rows, dims, corpus)
    eng.setRowBase(row_base)
    apply_tunes(eng)
    # batches in flight, like the headline's --depth software pipeline. Auto: three for batches of up to 256 queries, two for larger
    # ones — with a third ticket out, the next batch's threshold kernel and the stream gaps around it run under the current filtering
    # GEMM instead of between two GEMMs (1M x 384 x 256 queries: 0.216 -> 0.210 ms per batch, clustered k = 100 0.323 -> 0.284), while
    # 1 024-query batches lose 3 % to the extra small kernels in front of their long GEMMs (profiles/r05/k_batch_depth_2_3_4.txt)
    depth = BATCH_DEPTH if BATCH_DEPTH > 0 else (3 if nq <= 256 else 2)
    dq = batch_queries(torch, dev, nq, dims, corpus)
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
    nblk = max(5, steps // 4)
    _bracket(torch)
    tb = time.perf_counter()
    for _ in range(nblk):
        eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, stream)
    _bracket(torch)
    blocking_ms = (time.perf_counter() - tb) / nblk * 1e3
    fb0, rt0, mp0 = eng.getTuning("batch_fallbacks"), eng.getTuning("batch_retries"), eng.getTuning("batch_multi_passes")
    ir0 = eng.getTuning("batch_inline_retries")
    # Timed region = the product path, like the headline's: nothing is timed inside it ("time_kernels" = 0 — round 5, second session;
    # until then the filtering GEMMs were timed and chained INSIDE the region, and the chain wait and the event packets sat between
    # the threshold kernel and the GEMM of every batch: ~12 us of a 0.22 ms batch, profiles/r05/k_pipelined_batch_timeline.csv).
    eng.setTuning("time_kernels", 0)
    apply_tunes(eng)
    eng.setTuning("reset_stats", 1)
    _bracket(torch)
    t0 = time.perf_counter()
    run(steps)
    _bracket(torch)
    el = time.perf_counter() - t0
    fb1, rt1, mp1, ir1 = (eng.getTuning("batch_fallbacks"), eng.getTuning("batch_retries"), eng.getTuning("batch_multi_passes"),
                          eng.getTuning("batch_inline_retries"))
    last_ck = _hits_checksum(outs[(steps - 1) % depth])

    # calibration passes right behind it (same engine, same batches, still two in flight): the filtering GEMMs timed and chained so
    # that an interval is one GEMM — kernel-bound HIP events ("time_kernels" = 2, see measure_single_query), then the hipEventRecord
    # bracket beside it (--events bracket: the bracket only)
    def cal_pass(mode, n_steps):
        eng.setTuning("time_kernels", mode)
        run(2)
        eng.setTuning("reset_stats", 1)
        run(n_steps)
        _bracket(torch)
        st_ = eng.stats()
        n_ = int(st_.batch_gemms_timed)
        return (st_.batch_gemm_ms_total / n_ if n_ else float("nan")), n_

    n_cal = max(4, min(steps, 60))
    br_ms, br_n = cal_pass(1, n_cal if EVENT_MODE != "bound" else max(4, min(steps, 20)))
    kern_ms, launches, events = br_ms, br_n, "bracketed"
    if EVENT_MODE == "bound":
        kb_ms, kb_n = cal_pass(2, n_cal)
        if kb_n and br_n and kernel_bound_plausible(kb_ms, br_ms):
            kern_ms, launches, events = kb_ms, kb_n, "kernel-bound"
    eng.setTuning("time_kernels", 0)
    flops, floor_s, rf = batched_roofline(rows, dims, nq, kern_ms, launches, int(eng.getTuning("batch_rega")))
    rf["events"], rf["kernel_avg_ms_bracketed"] = events, br_ms
    want_live = live_traffic                  # roofline.traffic of the filtering GEMM is measured below (child counter pass)
    res = {
        "config": label,
        "value": nq * steps / el, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
        "dtype": "bf16 GEMM, exact f32 re-score", "queries_per_step": nq, "batches_in_flight": depth, "corpus": corpus,
        "ms_per_step_blocking_call": blocking_ms,
        "end_to_end_tflops_bf16": flops / (el / steps) / 1e12,
        "end_to_end_frac_of_roof": floor_s / (el / steps),
        "certificate_fallbacks": int(fb1 - fb0),
        "certificate_fallbacks_per_step": (fb1 - fb0) / (steps + 0.0),
        "full_retries": int(rt1 - rt0),
        "full_retries_on_device": int(ir1 - ir0),
        "shared_exact_passes": int(mp1 - mp0),
        "pipeline": "one-pass" if eng.getTuning("onepass_queries") > 0 else "slab",
        "last_result_checksum": last_ck,
        "roofline": rf,
    }
    rega = int(eng.getTuning("batch_rega"))
    eng.close()
    if want_live and TRAFFIC_MODE in ("auto", "live"):
        # HBM bytes per launch of the filtering GEMM, measured now (the engine above is released first): a counters-only child pass
        del dq, outs
        torch.cuda.empty_cache()
        _, _, res["roofline"] = batched_roofline(rows, dims, nq, kern_ms, launches, rega, live={"k": k, "corpus": corpus, "row_base": row_base})
        res["roofline"]["events"], res["roofline"]["kernel_avg_ms_bracketed"] = events, br_ms
    return res


def _hits_checksum(hits):
    """sha256 over the [nq][k] (key, frame id) hits of a batch: equal at every shard count / launch shape."""
    import hashlib
    a = hits.cpu().numpy() if hasattr(hits, "cpu") else np.asarray(hits)
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()[:16]


def config5_sharded(torch, dist, args, rank, world, in_library, use_rccl, k=10, nq=1024, dims=768):
    """BASELINE config 5 over the N GPUs of this run: 10M x 768 row-sharded, 1024 queries per step (bf16 MFMA GEMM + fused
    top-k on every shard, exact re-score), per-shard [nq][k] hits exchanged and merged per query by key. Both launch shapes:
      * torchrun ranks: every rank answers the batch on its shard (device-resident), one RCCL all-gather of nq*k hits per
        rank, merge on every rank (wax_amd.sharded.ShardedBatchSearcher, two batches in flight);
      * one process: the sharded handle's own device-resident submit / collect (peer copies of the hits to the first
        device, merge there).
    `value` = queries/s of the whole job (max over ranks of the elapsed time)."""
    from wax_amd import sharded
    rows = args.c5_rows
    steps, warmup, depth = max(10, min(args.steps // 4, 40)), max(3, min(args.warmup, 6)), 2
    same = bool(os.environ.get("WAX_BENCH_SAME_DEVICE"))
    if in_library:
        n_sh = args.gpus
        devs = [0 if same else g for g in range(n_sh)]
        dev0 = torch.device("cuda", devs[0])
        eng = _load_handle(torch, devs, rows, dims)
        per = max(int(eng.shardInfo(g)[2]) for g in range(n_sh))
        rows_per_gpu = per
        dev = dev0
    else:
        lo, hi = sharded.shard_bounds(rows, world, rank, align=64)
        dev = torch.device("cuda", torch.cuda.current_device())
        eng = _load_engine(torch, dev, hi - lo, dims, lo=lo)
        eng.setRowBase(lo)
        rows_per_gpu = hi - lo
    apply_tunes(eng)
    dq = torch.from_numpy(unit_queries(nq, dims)).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist_barrier(torch, dist, use_rccl, torch.cuda.current_device())
        torch.cuda.synchronize()

    if in_library:
        outs = [torch.empty((nq, k, 2), dtype=torch.int64, device=dev) for _ in range(depth)]

        def run(n_steps):
            tickets = []
            for i in range(n_steps):
                if len(tickets) == depth:
                    eng.searchBatchCollectDevice(tickets.pop(0))
                tickets.append(eng.searchBatchSubmitDevice(dq.data_ptr(), nq, k, outs[i % depth].data_ptr(), k, stream))
            for t in tickets:
                eng.searchBatchCollectDevice(t)
            return outs[(n_steps - 1) % depth]
        eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, outs[0].data_ptr(), k, stream)     # mirrors (untimed)
    else:
        searcher = sharded.ShardedBatchSearcher(eng, rank, world, k, nq, depth=depth, exchange="rccl" if use_rccl else "host")

        def run(n_steps):
            last = None
            for _ in range(n_steps):
                if searcher.pending() == depth:
                    last = searcher.collect()
                searcher.submit(dq)
            while searcher.pending():
                last = searcher.collect()
            return last
        searcher.submit(dq)
        searcher.collect()                                                                   # mirror (untimed)
    run(warmup)
    eng.setTuning("time_kernels", 0)                 # timed region = the product path (nothing timed or chained inside it)
    eng.setTuning("reset_stats", 1)
    barrier()
    t0 = time.perf_counter()
    last = run(steps)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if use_rccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    last_ck = _hits_checksum(last)
    fallbacks = int(eng.getTuning("batch_fallbacks"))
    # calibration pass (every rank, the same number of batches): the filtering GEMMs timed and chained, HIP events recorded around them
    eng.setTuning("time_kernels", 1)
    eng.setTuning("reset_stats", 1)
    run(max(4, min(steps, 60)))
    barrier()
    eng.setTuning("time_kernels", 0)
    st = eng.stats()
    launches = int(st.batch_gemms_timed)
    kern_ms = st.batch_gemm_ms_total / launches if launches else float("nan")
    flops, floor_s, rf = batched_roofline(rows_per_gpu, dims, nq, kern_ms, launches, int(eng.getTuning("batch_rega")))
    n_gpus = args.gpus if in_library else world
    res = {
        "config": f"{rows} x {dims} row-sharded over {n_gpus} GPU(s) ({rows_per_gpu} rows each), {nq} queries per step, cosine top-{k}, "
                  f"bf16 MFMA GEMM + fused top-k per shard + exact f32 re-score, per-shard hits "
                  + ("peer-copied to the first device and merged there (ONE process, sharded handle)" if in_library else
                     ("all-gathered over RCCL and merged on every rank" if use_rccl else "all-gathered on the host (gloo) and merged"))
                  + " (BASELINE config 5), queries and results resident in HBM, 2 batches in flight",
        "value": nq * steps / el, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
        "n_gpus": n_gpus, "scaling": "strong", "dtype": "bf16 GEMM, exact f32 re-score", "queries_per_step": nq,
        "batches_in_flight": depth, "rows_per_gpu": rows_per_gpu,
        "end_to_end_tflops_bf16": 2.0 * nq * rows * dims / (el / steps) / 1e12,
        "end_to_end_frac_of_roof_per_gpu": floor_s / (el / steps),
        "certificate_fallbacks": fallbacks,
        "last_result_checksum": last_ck,
        "roofline": rf,
    }
    eng.close()
    return res


# ---------------------------------------------------------------------------
# output: ONE compact JSON line on stdout (the driver keeps a bounded tail of stdout: round 3's 22 KB line was cut and the
# round went unrecorded), the verbose record in a file and on stderr

LINE_BUDGET = 4096     # bytes; tests/test_host_cpu.py builds a worst-case record and asserts the line stays under it


def _r(x, sig=6):
    """Floats to `sig` significant digits (the line is a report, not a checkpoint); everything else unchanged."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d}


def compact_line(full):
    """The contract line from the full record: the contract's keys, `roofline` and `cpu_baseline` with numbers only, and one
    five-number entry per secondary configuration. Prose (`config` paragraphs, `traffic_source`, `note`, `calibration`,
    sustained-roof fields, per-variant CPU samples) stays in the detail file."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    cfg = full.get("config") or {}
    c = _pick(cfg, ("rows", "dims", "top_k", "rows_per_gpu", "pipeline_depth", "merge", "exchange", "rccl_ranks", "shards",
                    "devices"))
    c["workload"] = cfg.get("workload_short") or str(cfg.get("workload", ""))[:120]
    c["parallelism"] = cfg.get("parallelism_short") or str(cfg.get("parallelism", ""))[:60]
    c["checksum"] = cfg.get("last_result_checksum")
    if isinstance(cfg.get("preflight"), dict):       # one-process multi-GPU shape: first-contact check, numbers only
        pf = cfg["preflight"]
        c["preflight"] = {k_: pf.get(k_) for k_ in ("ok", "peer_pairs", "peer_enabled") if k_ in pf}
        if pf.get("error"):
            c["preflight"]["error"] = str(pf["error"])[:80]
    if cfg.get("exchange"):
        c["exchange"] = str(cfg["exchange"])[:80]
    line["config"] = c
    rf = full.get("roofline")
    if rf is not None:
        r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "pipeline_frac", "kernel_avg_ms", "kernel_launches_timed",
                       "algorithmic_bytes_per_launch", "traffic", "peak_measured", "frac_of_measured"))
        r["kernel"] = str(rf.get("kernel", "")).split(" ")[0]
        cal = rf.get("calibration") or {}
        if cal.get("events"):
            r["events"] = cal["events"]                  # "kernel-bound" (hipExtLaunchKernel pair) or "bracketed" (hipEventRecord around the launch)
            r["kernel_avg_ms_bracketed"] = _r(cal.get("kernel_avg_ms_bracketed"))
        if rf.get("traffic") is not None:
            r["traffic_from"] = "live-pmc" if str(rf.get("traffic_source", "")).startswith("measured in this run") else "replayed-pmc"
        line["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb is not None:
        b = _pick(cb, ("value", "unit", "cores", "kind"))
        v1 = next((v for v in cb.get("variants", []) if v.get("threads") == 1), None)
        if v1:
            b["value_1_thread"] = _r(v1["value"])
        b["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:100]
        line["cpu_baseline"] = b
    elif "cpu_baseline" in full:
        line["cpu_baseline"] = None
    sec = []
    for x in full.get("secondary") or []:
        if "error" in x:
            sec.append({"name": x.get("name"), "error": str(x["error"])[:80]})
            continue
        e = {"name": x.get("name"), "value": _r(x.get("value")), "ms_per_step": _r(x.get("ms_per_step"))}
        xr = x.get("roofline") or {}
        e["frac"] = _r(xr.get("frac"), 4)
        e["kernel_avg_ms"] = _r(xr.get("kernel_avg_ms"), 4)
        br = (xr.get("calibration") or {}).get("kernel_avg_ms_bracketed") if "calibration" in xr else xr.get("kernel_avg_ms_bracketed")
        if br is not None:
            e["bracketed_ms"] = _r(br, 4)                # the same launches under hipEventRecord brackets (kernel + packets around it)
        e["ck"] = x.get("last_result_checksum")          # equal at every N / launch shape for the same workload
        if x.get("n_gpus", 1) != 1:
            e["n_gpus"] = x["n_gpus"]
            if "rows_per_gpu" in x:
                e["rows_per_gpu"] = x["rows_per_gpu"]
            if x.get("rccl_ranks"):
                e["rccl_ranks"] = x["rccl_ranks"]
            if x.get("exchange"):
                e["exchange"] = str(x["exchange"])[:60]
        if "ms_per_step_blocking_call" in x:
            e["blocking_ms"] = _r(x["ms_per_step_blocking_call"], 4)
        xc = x.get("cpu_baseline")
        if isinstance(xc, dict) and "value" in xc:       # the oracle's CPU scan of the same rows, same run: q/s on `cores` threads / on one
            e["cpu"] = {"qps": _r(xc["value"], 4), "cores": xc.get("cores"), "qps_1t": _r(xc.get("value_1_thread"), 4)}
        sec.append(e)
    if "secondary" in full:
        line["secondary"] = sec
    if full.get("detail"):
        line["detail"] = full["detail"]
    return line


def emit(full, detail_out=None):
    """Write the full record (file + stderr), print the compact line (stdout, last)."""
    path = detail_out or os.path.join(ROOT, "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["detail"] = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as ex:
        log(f"[bench] could not write {path}: {ex}")
    log("[bench detail] " + json.dumps(full))
    text = json.dumps(compact_line(full), separators=(",", ":"))
    if len(text) >= LINE_BUDGET:       # cannot happen with the fields above (tested); never print an oversized line
        slim = compact_line({k: v for k, v in full.items() if k != "secondary"})
        slim["secondary_dropped"] = f"line was {len(text)} bytes; see the detail file"
        text = json.dumps(slim, separators=(",", ":"))
    sys.stdout.flush()
    print(text, flush=True)


import threading as _threading

_EMIT_ONCE = {"lock": _threading.Lock(), "done": False}


def emit_once(full, detail_out=None):
    """emit(), at most once per process: the main thread and the secondaries' watchdog may both get here."""
    with _EMIT_ONCE["lock"]:
        if _EMIT_ONCE["done"]:
            return
        _EMIT_ONCE["done"] = True
        emit(full, detail_out)


SECONDARY_LIMIT_S = float(os.environ.get("WAX_BENCH_SECONDARY_LIMIT_S", "600"))


def arm_secondary_watchdog(out, args, rank, limit_s=None):
    """The headline is measured before the secondaries and printed after them (ONE line): a secondary that never returns — at N > 1 a
    collective one rank never joins, code that has only ever run on one physical GPU — would lose it. After `limit_s` seconds in the
    secondaries rank 0 prints the line with what has finished (plus an entry that says so) and every rank leaves (exit code 0, the
    other ranks a few seconds behind rank 0 so that its line is out first). Cancelled when the secondaries return."""
    import threading
    limit_s = SECONDARY_LIMIT_S if limit_s is None else limit_s

    def fire():
        try:
            if rank == 0 and out is not None:
                out.setdefault("secondary", []).append(
                    {"name": "watchdog", "error": f"secondaries still running after {limit_s:.0f} s: line emitted without the rest"})
                emit_once(out, args.detail_out)
                sys.stdout.flush()
        finally:
            os._exit(0)

    t = threading.Timer(limit_s + (0.0 if rank == 0 else 5.0), fire)
    t.daemon = True
    t.start()
    return t


def main():
    args = parse_args()
    TUNES.extend(args.tune)
    global EVENT_MODE, BATCH_DEPTH
    EVENT_MODE = args.events
    BATCH_DEPTH = max(0, min(4, args.batch_depth))
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
    if args.exchange is None:
        args.exchange = "tickets" if in_library else "rccl"
    if args.exchange == "tickets" and not in_library:
        args.exchange = "host"
    use_rccl = args.exchange == "rccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_rccl = init_distributed(torch, dist, rank, world, dev, use_rccl)

    from wax_amd import HIPVectorEngine, VectorMetric, build, sharded
    build.build()
    if not HIPVectorEngine.isAvailable():
        raise SystemExit("no gfx950 device visible: bench.py measures the HIP path only")

    n, dims, k = args.rows, args.dims, args.topk
    if args.traffic_child:
        if args.traffic_child_batch:
            traffic_child_batch(torch, dev, n, dims, k, args.traffic_child_batch)
        else:
            traffic_child(torch, dev, n, dims, k)
        return
    global TRAFFIC_MODE
    TRAFFIC_MODE = args.traffic if (world == 1 and not in_library) else ("replay" if args.traffic in ("auto", "replay") else "off")
    lo, hi = sharded.shard_bounds(n, world, rank, align=64)
    solo = (not in_library) and small_store(n, dims, world)   # one rank per GPU, a store too small to shard: rank 0 holds and answers it
    if solo:
        lo, hi = (0, n) if rank == 0 else (0, 0)
    t_build = time.perf_counter()
    preflight = None
    if in_library:
        same = bool(os.environ.get("WAX_BENCH_SAME_DEVICE"))          # testing only: every shard on GPU 0
        devs = [0 if same else g for g in range(args.gpus)]
        preflight = handle_preflight(torch, devs)
        eng = _load_handle(torch, devs, n, dims)                       # block layout chosen by the library (ceil(n / N) rows per shard at this size)
        shard_rows = [int(eng.shardInfo(g)[2]) for g in range(args.gpus)]
        lo, hi = 0, max(shard_rows)                                     # rows per launch of ONE shard's scan kernel (roofline)
    else:
        eng = HIPVectorEngine(metric=VectorMetric.cosine, dimensions=dims)
        eng.reserve(max(hi - lo, 1))
        for r0, x in (device_rows(torch, lo, hi, dims, dev) if hi > lo else ()):
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

    apply_tunes(eng)
    if in_library:
        handle_exchange(eng, args, args.gpus)   # --exchange rccl: one ncclAllGather per query on the library's communicator, N ranks or no run
    elif world > 1 and use_rccl and dist.get_world_size() != world:
        raise SystemExit(f"[bench] --exchange rccl: the process group has {dist.get_world_size()} ranks, expected {world}")
    if world == 1 or solo:
        # two in-order streams: one query's merge / result write / next query upload hide under the neighbouring scan,
        # and (product mode) neighbouring scans overlap each other's ramp and tail
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
    if not (solo and rank != 0):
        assert int(ids[0]) == probe_row and abs(float(scores[0]) - 1.0) < 1e-5, (ids[:3], scores[:3])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist_barrier(torch, dist, use_rccl, local_rank)
        torch.cuda.synchronize()

    # Python's cyclic collector must not fire inside the timed region: with torch imported a full (generation-2)
    # collection walks millions of objects and pauses the submitting thread for 40-70 ms (seen as a one-off stall
    # around the 450th query of a run, tools/long_run_drift.py); nothing below creates reference cycles.
    import gc
    gc.collect()
    gc.disable()
    elapsed, last, kern_ms, launches, cal = measure_single_query(eng, submit, collect, queries, args.warmup, args.steps, args.depth,
                                                                 barrier, chain_timed_region=args.chain_timed_region,
                                                                 floor_ms=max(hi - lo, 0) * dims * 4 / (HBM_PEAK_GBPS * 1e9) * 1e3)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if use_rccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert len(last[0]) == (min(k, n) if not (solo and rank != 0) else 0)
    # the node's own streaming-read rate over the same slab ("both denominators"): the library's microbenchmark —
    # every float4 of the store summed, no selection — timed with HIP events right behind the timed region, on this engine
    stream_read_gbps = None
    if not in_library and hi > lo:
        try:
            ms_sr = eng.timeStreamRead(10)
            if ms_sr and ms_sr > 0:
                stream_read_gbps = (hi - lo) * dims * 4 / (ms_sr * 1e-3) / 1e9
        except Exception as ex:  # noqa: BLE001
            log(f"[bench] stream-read microbenchmark unavailable: {type(ex).__name__}: {ex}")
    import hashlib
    checksum = hashlib.sha256(np.asarray(last[0], dtype=np.uint64).tobytes()
                              + np.asarray(last[1], dtype=np.float32).tobytes()).hexdigest()[:16]
    exchange_mode = eng.getTuning("exchange") if in_library else 0
    want = SECONDARY_N1 if args.secondary == "all" else [s for s in args.secondary.split(",") if s]
    out = None
    if rank == 0:
        qps = args.steps / elapsed
        bytes_per_launch = (hi - lo) * dims * 4
        # HBM traffic needs a PMC pass of its own (rocprofv3 --pmc FETCH_SIZE, never combined with tracing): at N = 1 a child process
        # does that pass inside this run (live_traffic). Where it cannot (N > 1, no rocprofv3, a failed pass) the figure is REPLAYED
        # from the committed counter pass of this same command and says so; without a matching pass the field is null.
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        live_note = None
        if args.traffic in ("auto", "live") and world == 1 and not in_library:
            traffic, live_note = live_traffic(n, dims, k)
            if traffic is not None:
                traffic_source = live_note
            else:
                log(f"[bench] live traffic pass unavailable: {live_note}")
        if traffic is None and args.traffic in ("auto", "replay") and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("rows_per_launch") == hi - lo and tj.get("dims") == dims:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "replayed from profiles/latest_traffic.json (" + str(tj.get("source", "rocprofv3 --pmc FETCH_SIZE pass of this command")) + "), not measured in this run"
                    if live_note:
                        traffic_source += "; live pass: " + live_note
            except Exception:  # noqa: BLE001
                traffic = None
        n_gpus = args.gpus if in_library else world
        if in_library:
            rccl_ranks = int(eng.getTuning("rccl_ranks")) if exchange_mode == 1 else 0
        else:
            rccl_ranks = (dist.get_world_size() if (world > 1 and use_rccl) else 0)
        out = {
            "metric": f"queries/sec, {_human_rows(n)} x {dims}-dim f32 cosine top-{k} brute-force scan (single query per step)",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": n_gpus,
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
                            f"one query per step, corpus resident in HBM and row-sharded over {n_gpus} GPU(s)",
                "workload_short": f"{_human_rows(n)} x {dims} f32 unit-Gaussian corpus in HBM, cosine top-{k}, 1 query/step",
                "rows": n, "dims": dims, "top_k": k, "rows_per_gpu": (shard_rows if in_library else ([n] + [0] * (world - 1)) if solo else hi - lo),
                # how many ranks the collective library actually joined (torchrun shape: dist world size after the all_reduce
                # probe; one-process shape: the library's ncclCommCount when its RCCL exchange is on, else 1 = peer copies)
                "rccl_ranks": rccl_ranks, "shards": n_gpus,
                "parallelism_short": (f"row-shard x{n_gpus} one-process " + ("rccl" if exchange_mode == 1 else "tickets+host-merge")) if in_library
                                     else (f"x{world} ranks, store on rank 0 (small-store rule)" if solo else
                                           f"row-shard x{world}" + ((" rccl all_gather" if use_rccl else " gloo all_gather") if world > 1 else "")),
                "parallelism": (f"row-shard x{args.gpus}, ONE process: the library's multi-GPU engine (wax_hip_engine_create_sharded), "
                                + ("single-process RCCL all-gather of per-shard top-k + merge on the first device" if exchange_mode == 1 else
                                   "per-shard tickets submitted side by side (one launch per shard, hits straight to pinned memory) + host merge by key")) if in_library else
                               (f"{world} ranks, the whole store on rank 0's GPU (small-store rule: below {SMALL_STORE_BYTES >> 20} MB nothing is sharded; "
                                f"no exchange)" if solo else
                                f"row-shard x{world}" + ((" + RCCL all-gather of per-shard top-k" if use_rccl else
                                                          " + host (gloo) all-gather of per-shard top-k") if world > 1 else "")),
                "pipeline_depth": args.depth,
                "timed_region": "kernels timed and chained inside it (--chain-timed-region)" if args.chain_timed_region else
                                "product mode: scans of neighbouring queries overlap; per-launch times from the calibration pass",
                "merge": "host" if (world > 1 and (args.host_merge or not use_rccl)) else "device",
                "exchange": (("in-library rccl" if exchange_mode == 1 else "in-library tickets" + (f" (rccl failed: {RCCL_FAILURE[0]})" if RCCL_FAILURE[0] else "")) if in_library else "none (small-store rule)" if solo else
                             (("rccl all_gather" if use_rccl else ("host (gloo)" + (f" (rccl failed: {RCCL_FAILURE[0]})" if RCCL_FAILURE[0] else ""))) if world > 1 else "none")),
                "last_result_checksum": checksum,
            },
            "roofline": scan_roofline(bytes_per_launch, kern_ms, launches, elapsed, args.steps, cal, traffic, traffic_source),
        }
        if preflight is not None:
            out["config"]["preflight"] = preflight
        if stream_read_gbps and out["roofline"].get("achieved"):
            out["roofline"]["peak_measured"] = stream_read_gbps          # GB/s: the second denominator, measured in this run on this GPU
            out["roofline"]["frac_of_measured"] = out["roofline"]["achieved"] / stream_read_gbps
            out["roofline"]["peak_measured_note"] = ("wax_hip_time_stream_read: 10 passes of a read-only float4 sum over the same slab, HIP events; "
                                                     "`peak` stays the 8 TB/s spec the contract names")
        if world == 1 and not in_library and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, args, dev, queries)
        elif world == 1:
            out["cpu_baseline"] = None
    if not args.no_secondary:
        # the other BASELINE configurations, timed after the headline (its engine is released first)
        gc.enable()
        eng.close()
        searcher = None
        gc.collect()
        gc.disable()
        sec = []
        if out is not None:
            out["secondary"] = sec
        watchdog = arm_secondary_watchdog(out, args, rank)
        if world == 1 and not in_library:
            s, w = args.steps, args.warmup
            table = {
                "s10k": lambda: secondary_single_query(torch, dev, 10_000, 384, k, max(s, 2000), max(w, 100), args.depth,
                                                       "the 10K-row point of the N matrix: launch-latency-bound, 15 MB per query",
                                                       cpu_seconds=0.0 if args.no_cpu_baseline else 2.0),
                "s1m": lambda: secondary_single_query(torch, dev, 1_000_000, 384, k, max(s, 600), max(w, 50), args.depth,
                                                      cpu_seconds=0.0 if args.no_cpu_baseline else 3.0),
                # what Wax.search(topK:) asks the engine for: candidateLimit = max(topK, min(3 topK, 1000)) (UnifiedSearch.swift:1195-1200) —
                # topK 100 -> k = 300, topK >= 334 -> k = 1000: beyond the fused selection (k <= 192), the general path of DESIGN 4.2
                "s10m_k300": lambda: secondary_single_query(torch, dev, args.rows, 384, 300, max(20, min(s, 60)), 6, args.depth,
                                                            "the headline store at top-300 = Wax.search(topK: 100): general selection"),
                "s1m_k1000": lambda: secondary_single_query(torch, dev, 1_000_000, 384, 1000, max(s, 300), max(w, 30), args.depth,
                                                            "1M rows at top-1000 = Wax.search(topK: 334 ...): general selection"),
                "s1250k": lambda: secondary_single_query(torch, dev, 1_250_000, 384, k, max(s, 600), max(w, 50), args.depth,
                                                         "the headline's per-GPU shard at 8 GPUs (10M / 8 rows): what one rank of BASELINE config 4 scans per query"),
                # a store of the size Wax's own harness uses, at the top_k Wax.search(topK: 34 ...) requests: the one-pass pipeline since
                # round 6 (one scan per query before: 8 ms per batch); latency-bound — see blocking_ms
                "b30k_k100": lambda: secondary_batched(torch, dev, 30_000, 384, 256, 100, max(s, 200), max(w, 20),
                                                       "30000 x 384 (a Wax-sized store), 256 queries per step, cosine top-100, one-pass bf16 MFMA pipeline, "
                                                       "queries and results resident in HBM", live_traffic=False),
                "b1m_q256": lambda: secondary_batched(torch, dev, 1_000_000, 384, 256, k, max(s, 200), max(w, 20),
                                                      "1000000 x 384, 256 queries per step, cosine top-10, bf16 MFMA GEMM + fused top-k, 1 GPU "
                                                      "(BASELINE config 3), queries and results resident in HBM, batches in flight: see batches_in_flight"),
                "b1m_q1024": lambda: secondary_batched(torch, dev, 1_000_000, 384, 1024, k, max(s // 2, 100), max(w, 20),
                                                       "1000000 x 384, 1024 queries per step, cosine top-10, bf16 MFMA GEMM + fused top-k, 1 GPU "
                                                       "(config 3 at four times the batch: the MFMA-bound shape), queries and results resident in "
                                                       "HBM, batches in flight: see batches_in_flight"),
                "c5_shard": lambda: secondary_batched(torch, dev, 1_250_000, 768, 1024, k, max(s // 2, 60), max(w, 10),
                                                      "1250000 x 768 (one GPU's share of 10M x 768 over 8 GPUs), 1024 queries per step, cosine "
                                                      "top-10, bf16 MFMA GEMM + fused top-k (BASELINE config 5, per-GPU part), queries and "
                                                      "results resident in HBM, batches in flight: see batches_in_flight", row_base=3_750_000),
                "c5_full": lambda: secondary_batched(torch, dev, args.c5_rows, 768, 1024, k, max(10, min(s // 8, 25)), 3,
                                                     f"{args.c5_rows} x 768, 1024 queries per step, cosine top-10, bf16 MFMA GEMM + fused top-k, ALL "
                                                     "rows on ONE GPU (BASELINE config 5 at full size: the N = 1 point of its 1/2/4/8-GPU scaling "
                                                     "curve), queries and results resident in HBM, batches in flight: see batches_in_flight"),
                "clustered_k10": lambda: secondary_batched(torch, dev, 1_000_000, 384, 256, 10, max(s // 2, 100), max(w, 10),
                                                           "1000000 x 384 CLUSTERED corpus (20 tight clusters: normalize(centre + 0.3 gaussian), half "
                                                           "of the queries inside a cluster), 256 queries per step, cosine top-10: the batched path "
                                                           "where bf16 cannot separate neighbours — certificate fallbacks share exact passes",
                                                           corpus="clustered"),
                "clustered_k100": lambda: secondary_batched(torch, dev, 1_000_000, 384, 256, 100, max(s // 4, 40), max(w // 2, 5),
                                                            "the same clustered corpus and queries at top-100 (the dense-neighbourhood regime of "
                                                            "tools/fuzz_batch.py: the k-th neighbour has hundreds of rows inside its bf16 error band)",
                                                            corpus="clustered"),
                "dups17": lambda: secondary_batched(torch, dev, 1_000_000, 384, 256, 10, max(s // 2, 100), max(w, 10),
                                                    f"1000000 x 384 with {DUP_ROWS} exact duplicates of one row, {DUP_QUERIES} of the 256 queries per step "
                                                    "(17 %) aimed at them: ties no bf16 filter can order — those queries take the exact path, "
                                                    "sharing passes over the f32 store (16 per pass, bit-identical to the single-query kernel); "
                                                    "cosine top-10", corpus="dups"),
                "detembed": lambda: secondary_batched(torch, dev, 1_000_000, 384, 256, 10, max(s // 2, 100), max(w, 10),
                                                      "1000000 x 384 rows of the reference's DeterministicEmbedder (FNV-1a / LCG on \"doc-<i>\", "
                                                      "RAGBenchmarkSupport.swift:114-157), 256 queries per step, cosine top-10", corpus="detembed"),
            }
            for name in want:
                if name not in table:
                    continue
                try:
                    r = table[name]()
                    r["name"] = name
                    sec.append(r)
                except Exception as ex:  # noqa: BLE001 — a secondary failure must not lose the headline line
                    sec.append({"name": name, "error": f"{type(ex).__name__}: {ex}"})
            iid = next((x for x in sec if x.get("name") == "b1m_q256" and "error" not in x), None)
            for x in sec:
                if iid and x.get("corpus") in ("clustered", "detembed", "dups") and "error" not in x and x.get("queries_per_step") == 256:
                    x["ms_per_step_vs_iid_config3"] = x["ms_per_step"] / iid["ms_per_step"]
        else:
            # N > 1: the rest of the north star's N matrix on the same sharded path (torchrun shape), then config 5
            if world > 1:
                for name, rows_, steps_, warm_, label in (
                        ("s1m", 1_000_000, max(args.steps, 600), max(args.warmup, 50), "N matrix: 1M rows; BASELINE config 2's corpus over the node"),
                        ("s10k", 10_000, max(args.steps, 2000), max(args.warmup, 100), "N matrix: 10K rows — exchange-latency-bound")):
                    if args.secondary != "all" and name not in want:
                        continue
                    try:
                        r = sharded_single_query(torch, dist, args, rank, world, local_rank, use_rccl, rows_, 384, k, steps_, warm_, label)
                        r["name"] = name
                        sec.append(r)
                    except Exception as ex:  # noqa: BLE001
                        sec.append({"name": name, "error": f"{type(ex).__name__}: {ex}"})
            if in_library:
                same_dev = bool(os.environ.get("WAX_BENCH_SAME_DEVICE"))
                devs_ = [0 if same_dev else g for g in range(args.gpus)]
                # the headline store under BOTH exchanges of the handle, whatever the headline itself used: one run = the comparison
                for name, rccl_ in (("h_tickets", False), ("h_rccl", True)):
                    if args.secondary != "all" and name not in want:
                        continue
                    try:
                        RCCL_FAILURE[0] = None
                        r = handle_single_query(torch, args, devs_, n, dims, k, args.steps, args.warmup,     # (the headline's own steps: same last query, same checksum)
                                                "the headline store, exchange = " + ("RCCL all-gather (device gather)" if rccl_ else "per-shard tickets + host merge"),
                                                want_rccl=rccl_)
                        r["name"] = name
                        if rccl_ and RCCL_FAILURE[0]:
                            r["exchange"] = f"tickets (rccl failed: {RCCL_FAILURE[0]})"
                        sec.append(r)
                    except Exception as ex:  # noqa: BLE001
                        sec.append({"name": name, "error": f"{type(ex).__name__}: {ex}"})
                for name, rows_, steps_, warm_, label in (
                        ("s1m", 1_000_000, max(args.steps, 600), max(args.warmup, 50), "N matrix: 1M rows; BASELINE config 2's corpus over the node"),
                        ("s10k", 10_000, max(args.steps, 2000), max(args.warmup, 100), "N matrix: 10K rows — a 15 MB store is not spread (small-store rule)")):
                    if args.secondary != "all" and name not in want:
                        continue
                    try:
                        r = handle_single_query(torch, args, devs_, rows_, 384, k, steps_, warm_, label)
                        r["name"] = name
                        sec.append(r)
                    except SystemExit:
                        raise
                    except Exception as ex:  # noqa: BLE001
                        sec.append({"name": name, "error": f"{type(ex).__name__}: {ex}"})
            if args.secondary == "all" or "c5" in want:
                try:
                    r = config5_sharded(torch, dist, args, rank, world, in_library, use_rccl)
                    r["name"] = "c5"
                    sec.append(r)
                except Exception as ex:  # noqa: BLE001
                    sec.append({"name": "c5", "error": f"{type(ex).__name__}: {ex}"})
        watchdog.cancel()
    if rank == 0:
        emit_once(out, args.detail_out)
    if world > 1:
        dist_barrier(torch, dist, use_rccl, local_rank)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
