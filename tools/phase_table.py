#!/usr/bin/env python3
"""One line per record of tools/gemm_phase_budget.py: kernel time (A/B median) and cycles per tile by phase."""
import json
import sys

for path in sys.argv[1:]:
    for l in open(path):
        d = json.loads(l)
        p = d["phases"]
        if not p:
            print(d["dims"], d["nq"], "k", d["topk"], d.get("ab_key", ""), d.get("batch_opt", 0), "ab_median_us %s prod_us %.1f (no phase-timing build of this variant)" % (
                d.get("product_kernel_us_ab_median"), d["product_kernel_us_hip_events"]))
            continue
        f = lambda n, w: round(p[n][w]["mean_cycles_per_tile"])  # noqa: E731
        ab = d.get("product_kernel_us_ab_median")
        print(d["dims"], d["nq"], "k", d["topk"], "opt", d.get("batch_opt", 0),
              "ab_median_us %s prod_us %.1f prof_us %.1f eq %s ghz %.2f | early sel %d (cold %d) wait %d dma %d k %d dmaw %d | late sel %d (cold %d) wait %d dma %d k %d dmaw %d | loop/tile %d | cold/coldtile %d" % (
                  ("%.1f" % ab) if ab else "-", d["product_kernel_us_hip_events"], d["prof_kernel_us_hip_events"], d["prof_answers_equal_product"],
                  d["shader_clock_ghz_median_wave"],
                  f("select", "early"), f("cold", "early"), f("wait_arrivals", "early"), f("dma_issue", "early"), f("kloop", "early"), f("dma_wait", "early"),
                  f("select", "late"), f("cold", "late"), f("wait_arrivals", "late"), f("dma_issue", "late"), f("kloop", "late"), f("dma_wait", "late"),
                  f("loop", "early"), round(p["cold_cycles_per_cold_tile"])))
