#!/usr/bin/env python3
"""How the GPU box's host scales the oracle's tuned CPU scan: cgroup CPU quota, affinity mask, and sample GB/s at
1 .. all threads. Explains the `cpu_baseline.variants` numbers bench.py reports (the box is a container slice)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402


def read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


def main():
    oracle.build()
    out = {"cpu.max": read("/sys/fs/cgroup/cpu.max"), "cpuset": read("/sys/fs/cgroup/cpuset.cpus.effective"),
           "affinity": len(os.sched_getaffinity(0)), "os.cpu_count": os.cpu_count(), "omp_max": oracle.max_threads(),
           "loadavg": read("/proc/loadavg"), "sweep": []}
    n, d = 1_000_000, 384
    tmax = oracle.max_threads()
    rng = np.random.default_rng(3)
    q = rng.standard_normal(d).astype(np.float32)
    for t in sorted({1, 2, 4, 8, 16, 32, 64, tmax}):
        if t > tmax:
            continue
        x = oracle.numa_sample(n, d, t)
        x[:] = 0.001
        oracle.scan_topk_fast(0, x, q, 10, t)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 1.5:
            oracle.scan_topk_fast(0, x, q, 10, t)
            reps += 1
        el = time.perf_counter() - t0
        out["sweep"].append({"threads": t, "gbps": round(n * d * 4 * reps / el / 1e9, 2), "ms": round(el / reps * 1e3, 2)})
        del x
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
