"""Worker for tests/test_bench_first_contact.py: one rank of a world_size-2 job on CPU that runs bench.init_distributed() where RCCL
cannot work (no GPU in this container, or WAX_BENCH_FAKE_RCCL_FAILURE=1) and reports what it decided."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    use_rccl = bench.init_distributed(torch, dist, rank, world, torch.device("cpu"), True)
    # whatever was decided, the barrier and a reduction of the ranks' elapsed times must still work (what bench.py does next)
    bench.dist_barrier(torch, dist, use_rccl, 0)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    with open(os.environ["WAX_OUT"] + f".{rank}", "w") as f:
        json.dump({"rank": rank, "use_rccl": bool(use_rccl), "why": bench.RCCL_FAILURE[0], "max": float(t.item())}, f)
    bench.dist_barrier(torch, dist, use_rccl, 0)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
