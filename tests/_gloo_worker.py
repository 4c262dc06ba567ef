"""Worker for tests/test_sharded_gloo.py: one rank of a world_size-N gloo job on CPU.

Per-shard top-k comes from the oracle here (no GPU in this container) — what is under test
is the N>1 host path: shard bounds, global-row keys, the all-gather exchange and the
G*k -> k merge, which are the same code bench.py runs over RCCL."""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def order_key(d: np.ndarray, rows: np.ndarray) -> np.ndarray:
    b = d.astype(np.float32).view(np.int32).astype(np.int64)
    o = b ^ ((b >> 31) & 0x7FFFFFFF)
    return (o << 32) | rows.astype(np.int64)


def main():
    import torch
    import torch.distributed as dist
    import oracle
    from wax_amd import HIPVectorEngine, VectorMetric, sharded

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, k, metric = int(os.environ["WAX_N"]), int(os.environ["WAX_D"]), int(os.environ["WAX_K"]), 0
    pattern = os.environ.get("WAX_PATTERN", "gauss")
    lo, hi = sharded.shard_bounds(n, world, rank, align=64)
    rows = oracle.gaussian_unit_rows(lo, hi - lo, d) if pattern == "gauss" else oracle.tie_pattern(lo, hi - lo, d)
    queries = oracle.gaussian_unit_queries(3, d)
    if pattern != "gauss":
        queries = np.abs(queries)
    out = []
    for q in queries:
        local = np.empty((k, 2), dtype=np.int64)
        local[:, 0] = sharded.KEY_PAD
        local[:, 1] = -1
        if hi > lo:
            dd = oracle.distances(metric, rows, q)
            idx, dsel = oracle.topk_heap(dd, k, total=True)
            local[:len(idx), 0] = order_key(dsel, idx + lo)          # key carries the GLOBAL row
            local[:len(idx), 1] = (idx + lo + 1000).astype(np.int64)  # frameId = row + 1000
        gathered = sharded.all_gather_hits(torch.from_numpy(local), world)
        merged = sharded.merge_hits_host(gathered.numpy(), k)
        ids, scores = HIPVectorEngine.hitsToResults(VectorMetric.cosine, merged)
        out.append({"ids": [int(x) for x in ids], "scores": [float(s) for s in scores]})
    # every rank must hold the same answer
    blob = json.dumps(out)
    allb = [None] * world
    dist.all_gather_object(allb, blob)
    assert all(b == blob for b in allb)
    if rank == 0:
        json.dump(out, open(os.environ["WAX_OUT"], "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
