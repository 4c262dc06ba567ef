#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run: per-kernel mean counter value (and FETCH_SIZE converted to
HBM bytes per launch with the gfx950 corrections of MI355X_MICROARCH.md §HBM: the counter is in
KiB... (rocprofv3 reports FETCH_SIZE in kilobytes) and reads exactly 1/2 of the bytes of a wide
coalesced stream on gfx950 => bytes = FETCH_SIZE * 1024 * 2)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            rows.extend(csv.DictReader(f))
    agg = defaultdict(lambda: defaultdict(list))
    for r in rows:
        name = r.get("Kernel_Name") or r.get("Kernel Name") or "?"
        cname = r.get("Counter_Name") or r.get("Counter Name") or "?"
        try:
            val = float(r.get("Counter_Value") or r.get("Counter Value") or "nan")
        except ValueError:
            continue
        agg[name][cname].append(val)
    out = {}
    for name, cs in agg.items():
        short = name.split("(")[0][-120:]
        out[short] = {}
        for cname, vals in cs.items():
            mean = sum(vals) / len(vals)
            ent = {"launches": len(vals), "mean": mean}
            if cname == "FETCH_SIZE":
                ent["hbm_bytes_per_launch_corrected"] = mean * 1024.0 * 2.0
                ent["hbm_bytes_per_launch_uncorrected"] = mean * 1024.0
            out[short][cname] = ent
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
