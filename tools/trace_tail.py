#!/usr/bin/env python3
"""Reduce a rocprofv3 kernel_trace.csv to the last N wax:: dispatches: kernel, start (us, relative), duration (us),
grid threads, VGPRs, scratch. Used to look at one batched search's launch sequence."""
import csv
import sys


def main():
    path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = [r for r in csv.DictReader(open(path)) if "wax::" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    if not rows:
        return
    t0 = int(rows[0]["Start_Timestamp"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "start_us", "duration_us", "grid_threads", "vgpr", "scratch"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        w.writerow([r["Kernel_Name"][:60], round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), r.get("Grid_Size", ""),
                    r.get("VGPR_Count", ""), r.get("Scratch_Size", "")])


if __name__ == "__main__":
    main()
