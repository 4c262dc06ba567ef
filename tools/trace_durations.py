#!/usr/bin/env python3
"""stdin: rocprofv3 kernel_trace.csv rows (no header) of one kernel family -> durations (us) in launch order, one per line,
with the kernel's template arguments. Column positions follow rocprofv3 1.x: Start/End timestamps are the two integer
columns after the kernel name."""
import csv
import sys

rows = []
for r in csv.reader(sys.stdin):
    ints = [(i, int(x)) for i, x in enumerate(r) if x.isdigit() and len(x) >= 15]
    if len(ints) < 2:
        continue
    name = next((x for x in r if "wax::" in x), "")
    rows.append((ints[-2][1], ints[-1][1] - ints[-2][1], name))
rows.sort()
for s, d, name in rows:
    print(f"{d / 1e3:9.1f}  {name[:90]}")
