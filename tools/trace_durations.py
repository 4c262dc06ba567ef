#!/usr/bin/env python3
"""Per-launch durations (us) of the kernels whose name contains argv[2] in a rocprofv3 kernel_trace.csv, in launch order."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r.get("Kernel_Name", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows))
