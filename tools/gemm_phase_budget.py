#!/usr/bin/env python3
"""Per-phase cycle budget of the filtering GEMM (`batch_gemm_rq_kernel`, BASELINE configs 3 / 5) from s_memtime.

The PROF instantiation of the kernel ("batch_prof_ptr" = device address of a [grid * 8][RQ_PROF_WORDS] u32 buffer) has every
wave add up the shader cycles it spends in each phase of its tile loop; this script runs one blocking batch on it, reads the
buffer and prints the budget: mean / max per phase for the early (0-3) and late (4-7) waves, the workgroups' wall spans from
s_memrealtime (start skew, tail imbalance) and the same launch's HIP-event time on the product kernel beside it.

    python tools/gemm_phase_budget.py --rows 1000000 --dims 384 --nq 256
    python tools/gemm_phase_budget.py --rows 1250000 --dims 768 --nq 1024
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

WORDS = 20
NAMES = ["prologue", "select", "cold", "cold_tiles", "wait_arrivals", "dma_issue", "kloop", "dma_wait", "loop", "epilogue", "tiles",
         "survivors", "rt0_lo", "rt0_hi", "rt1_lo", "rt1_hi", "xcc"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=384)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tune", action="append", default=[], help="key=value tuning applied before the runs")
    ap.add_argument("--out", default=None)
    ap.add_argument("--ab-key", default="batch_rega", help="tuning key the A/B runs over (round 6's kernel variants were switched by a key that no longer exists)")
    ap.add_argument("--opts", type=int, nargs="+", default=[1], help="values of --ab-key to run, one line each")
    ap.add_argument("--ab-rounds", type=int, default=0, help="rounds of the interleaved product-kernel A/B over --opts (0 = none)")
    args = ap.parse_args()
    import torch
    import wax_amd as wax
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = wax.HIPVectorEngine(metric=0, dimensions=args.dims)
    eng.reserve(args.rows)
    for r0, x in bench.device_rows(torch, 0, args.rows, args.dims, dev):
        eng.addBatchDevice(np.arange(r0, r0 + x.shape[0], dtype=np.uint64), x)
    for kv in args.tune:
        k, v = kv.split("=")
        eng.setTuning(k, int(v))
    q = bench.unit_queries(args.nq, args.dims)
    dq = torch.from_numpy(q).to(dev)
    dout = torch.empty((args.nq, args.topk, 2), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        eng.searchBatchHitsDevice(dq.data_ptr(), args.nq, args.topk, dout.data_ptr(), args.topk, st)
        torch.cuda.synchronize()

    for _ in range(3):
        run()
    ref = dout.clone()
    # interleaved A/B of the product kernels: the clock drifts by +-10 % within a process, so every variant is timed in every round
    # and the medians over the rounds are compared
    times = {o: [] for o in args.opts}
    if args.ab_rounds:
        eng.setTuning("time_kernels", 2)
        for _ in range(args.ab_rounds):
            for o in dict.fromkeys(args.opts):
                eng.setTuning(args.ab_key, o)
                run()
                eng.setTuning("reset_stats", 1)
                for _ in range(args.reps):
                    run()
                stt = eng.stats()
                times[o].append(stt.batch_gemm_ms_total / max(stt.batch_gemms_timed, 1) * 1e3)
        eng.setTuning("time_kernels", 0)
    for opt in dict.fromkeys(args.opts):
        eng.setTuning(args.ab_key, opt)
        budget(args, eng, torch, dev, run, ref, dout, opt, times[opt])


def budget(args, eng, torch, dev, run, ref, dout, opt, ab_times):
    for _ in range(2):
        run()
    # product kernel, timed by the library's dispatch-bound HIP events
    eng.setTuning("time_kernels", 2)
    eng.setTuning("reset_stats", 1)
    for _ in range(args.reps):
        run()
    stt = eng.stats()
    prod_us = stt.batch_gemm_ms_total / max(stt.batch_gemms_timed, 1) * 1e3
    # PROF kernel
    groups = (args.nq + 255) // 256
    grid = 256 // groups * groups
    prof = torch.zeros((grid * 8, WORDS), dtype=torch.int32, device=dev)
    eng.setTuning("batch_prof_ptr", prof.data_ptr())
    eng.setTuning("reset_stats", 1)
    for _ in range(args.reps):
        run()
    stt = eng.stats()
    prof_us = stt.batch_gemm_ms_total / max(stt.batch_gemms_timed, 1) * 1e3
    same = bool(torch.equal(ref, dout))
    p = prof.cpu().numpy().view(np.uint32).astype(np.int64).reshape(grid, 8, WORDS)   # the LAST launch's counts
    eng.setTuning("batch_prof_ptr", 0)
    eng.setTuning("time_kernels", 0)
    used = p[:, 0, 10] > 0
    p = p[used]
    if p.shape[0] == 0:        # this variant has no phase-timing build: times only
        line = json.dumps({"rows": args.rows, "dims": args.dims, "nq": args.nq, "topk": args.topk, "tune": args.tune, "ab_key": args.ab_key,
                           "batch_opt": opt, "product_kernel_us_hip_events": prod_us,
                           "product_kernel_us_ab_median": (float(np.median(ab_times)) if ab_times else None),
                           "product_kernel_us_ab_rounds": [round(x, 1) for x in ab_times], "prof_answers_equal_product": same, "phases": None})
        print(line, flush=True)
        if args.out:
            with open(args.out, "a") as f:
                f.write(line + "\n")
        return
    rt0 = (p[:, :, 12] | (p[:, :, 13] << 32)).astype(np.float64) * 10.0   # ns (100 MHz)
    rt1 = (p[:, :, 14] | (p[:, :, 15] << 32)).astype(np.float64) * 10.0
    k0, k1 = rt0.min(), rt1.max()
    wg_start = rt0.min(axis=1) - k0
    wg_end = rt1.max(axis=1) - k0
    span_us = (k1 - k0) / 1e3
    loop_cyc = p[:, :, 8].astype(np.float64)
    wall_wave_us = (rt1 - rt0) / 1e3
    total_cyc = (p[:, :, 0] + p[:, :, 8] + p[:, :, 9]).astype(np.float64)
    ghz = float(np.median(total_cyc / np.maximum(wall_wave_us, 1e-9)) / 1e3)
    out = {"rows": args.rows, "dims": args.dims, "nq": args.nq, "topk": args.topk, "tune": args.tune, "ab_key": args.ab_key, "batch_opt": opt,
           "product_kernel_us_hip_events": prod_us, "product_kernel_us_ab_median": (float(np.median(ab_times)) if ab_times else None),
           "product_kernel_us_ab_rounds": [round(x, 1) for x in ab_times], "prof_kernel_us_hip_events": prof_us, "prof_answers_equal_product": same,
           "workgroups": int(p.shape[0]), "kernel_span_us_first_entry_to_last_exit": span_us,
           "shader_clock_ghz_median_wave": ghz,
           "workgroup_start_skew_us": {"p50": float(np.median(wg_start) / 1e3), "max": float(wg_start.max() / 1e3)},
           "workgroup_end_before_kernel_end_us": {"p50": float(np.median(span_us - wg_end / 1e3)), "max": float((span_us - wg_end / 1e3).max()),
                                                   "mean": float((span_us - wg_end / 1e3).mean())},
           "tiles_per_workgroup": {"min": int(p[:, 0, 10].min()), "max": int(p[:, 0, 10].max())},
           "survivors_total": int(p[:, :, 11].sum())}
    phases = {}
    tiles = p[:, :, 10].astype(np.float64)
    for name, idx in (("prologue", 0), ("select", 1), ("cold", 2), ("wait_arrivals", 4), ("dma_issue", 5), ("kloop", 6), ("dma_wait", 7),
                      ("loop", 8), ("epilogue", 9)):
        v = p[:, :, idx].astype(np.float64)
        row = {}
        for label, sl in (("early", slice(0, 4)), ("late", slice(4, 8))):
            x = v[:, sl]
            row[label] = {"mean_cycles": float(x.mean()), "max_cycles": float(x.max()),
                          "mean_cycles_per_tile": float((x / np.maximum(tiles[:, sl], 1)).mean()),
                          "frac_of_loop": float((x / np.maximum(loop_cyc[:, sl], 1)).mean())}
        phases[name] = row
    rest = loop_cyc - p[:, :, 1] - p[:, :, 4] - p[:, :, 5] - p[:, :, 6] - p[:, :, 7]
    phases["loop_unattributed"] = {"mean_cycles": float(rest.mean()), "frac_of_loop": float((rest / np.maximum(loop_cyc, 1)).mean())}
    phases["cold_tiles_per_wave"] = {"mean": float(p[:, :, 3].mean()), "of_tiles": float((p[:, :, 3] / np.maximum(tiles, 1)).mean())}
    cold_n = p[:, :, 3].astype(np.float64)
    phases["cold_cycles_per_cold_tile"] = float(p[:, :, 2].sum() / max(cold_n.sum(), 1.0))
    out["phases"] = phases
    # per-XCC spread of workgroup wall time
    xcc = p[:, 0, 16] & 0xF
    out["per_xcc_workgroup_wall_us"] = {int(x): float(((rt1.max(axis=1) - rt0.min(axis=1))[xcc == x]).mean() / 1e3) for x in np.unique(xcc)}
    line = json.dumps(out)
    print(line, flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
