#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 kernel_trace.csv: start / end relative to the first one shown,
queue, and which dispatches overlap (start before the previous end)."""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
print("start_us,end_us,dur_us,queue,kernel")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void wax::", "").replace("wax::", "").split("(")[0][:48]
    print(f"{s / 1e3:9.1f},{e / 1e3:9.1f},{(e - s) / 1e3:7.1f},{r.get('Queue_Id', '?')},{name}")
