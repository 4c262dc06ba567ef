#!/usr/bin/env python3
"""Derived figures of the filtering GEMM from the SQ counter passes of tools/sessions/final.sh (profiles/<round>/z_pmc_sq_counters_*.json):
matrix-pipe busy share, parked / stalled wave shares, LDS activity, clock (cycles against the bench's HIP-event time of the same session).

    python tools/pmc_derive.py profiles/r06 > profiles/r06/z_pmc_sq_counters_rq_derived.json
"""
import json
import os
import sys

d = sys.argv[1]
XCDS, SIMDS, CUS = 8, 1024, 256


def bench_ms(name):
    try:
        rec = json.load(open(os.path.join(d, "z_bench_n1_final.json")))
        return next(s["kernel_avg_ms"] for s in rec["secondary"] if s["name"] == name)
    except Exception:
        return None


out = {}
for f, shapes in (("z_pmc_sq_counters_768_rq_final_build.json", [("768 Q=1024 1.25M rows", "<768, 32, 3, 3, false, false, false, false>", "c5_shard")]),
                  ("z_pmc_sq_counters_384_rq_final_build.json", [("384 Q=1024 1M rows", "<384, 64, 2, 3, false, false, false, false>", "b1m_q1024"),
                                                                  ("384 Q=256 1M rows (non-temporal requests)", "<384, 64, 2, 3, false, false, true, false>", "b1m_q256")])):
    try:
        rec = json.load(open(os.path.join(d, f)))
    except OSError:
        continue
    for label, suffix, sec in shapes:
        k = next((k for k in rec if k.endswith("batch_gemm_rq_kernel" + suffix)), None)
        if k is None:
            continue
        c = {n: v["mean"] for n, v in rec[k].items()}
        cyc = c["GRBM_GUI_ACTIVE"] / XCDS
        ms = bench_ms(sec)
        out[label] = {
            "kernel": k.replace("void ", ""),
            "cycles_per_xcd": round(cyc),
            "bench_hip_event_ms_same_session": ms,
            "approx_clock_ghz": round(cyc / (ms * 1e6), 3) if ms else None,
            "matrix_pipe_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / cyc, 4),
            "waves_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
            "issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4),
            "lds_issue_stall_frac (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES)": round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 4),
            "lds_array_active_frac (SQ_LDS_IDX_ACTIVE / cycles / 256 CUs)": round(c["SQ_LDS_IDX_ACTIVE"] / cyc / CUS, 4),
            "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
        }
out["note"] = ("matrix_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); round 5's values for the same three shapes: "
               "0.6432 / 0.5848 / 0.5117 (profiles/r05/z_pmc_sq_counters_rq_derived.json); the counter pass serialises kernels and clocks differently "
               "from the bench run whose HIP-event time stands beside it")
print(json.dumps(out, indent=1))
