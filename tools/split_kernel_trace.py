#!/usr/bin/env python3
"""bench.py's default command launches wax::scan_kernel for two workloads (the 10M-row headline and the 1M-row secondary
config 2), which `rocprofv3 --stats` averages into one row. This splits a kernel_trace.csv per kernel into duration
clusters (a gap of more than 3x between neighbouring sorted durations starts a new cluster) and prints calls / average /
min / max per cluster, so that each bench.py `kernel_avg_ms` has its rocprof counterpart."""
import csv
import sys
from collections import defaultdict


def main(path):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if "wax::" not in name:
            continue
        per[name.split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("kernel,cluster,calls,average_ns,min_ns,max_ns")
    for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        ds.sort()
        clusters, cur = [], [ds[0]]
        for d in ds[1:]:
            if d > 3 * cur[-1]:
                clusters.append(cur)
                cur = [d]
            else:
                cur.append(d)
        clusters.append(cur)
        for i, c in enumerate(clusters):
            print(f"\"{name}\",{i},{len(c)},{sum(c) / len(c):.1f},{c[0]},{c[-1]}")


if __name__ == "__main__":
    main(sys.argv[1])
