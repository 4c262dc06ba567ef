"""Row-sharded search across GPUs: one process per GPU, one HIPVectorEngine per process,
a per-query exchange of per-shard top-k over RCCL (torch.distributed backend "nccl"), then a
G*k -> k merge. The reference has no multi-device path; this is the one new
parallelism the MI355X build adds.

Exchange record: wax_hip_hit {int64 key = ordered(distance):global_row, uint64 frame_id}
carried as an int64[kpad, 2] tensor. Payload per query is G*kpad*16 bytes (8 GPUs, k=10:
1.3 KB) — latency-bound, so queries are software-pipelined over two HIP streams: the scan
of query i+1 overlaps the all-gather + merge of query i.

The merge has two interchangeable implementations with identical results:
  * device: libwaxhip's merge kernel (wax_hip_merge_hits_device) on the same stream;
  * host:   numpy sort of the gathered int64 keys (also what the gloo CPU tests exercise).
"""
from __future__ import annotations

from collections import deque
from typing import Deque, List, Optional, Tuple

import numpy as np

from . import _abi
from .engine import HIPVectorEngine, clampTopK
from .vector_metric import VectorMetric

KEY_PAD = _abi.KEY_PAD


def shard_bounds(n_rows: int, world: int, rank: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous row block of `rank`: [g*ceil(N/G), min(N, (g+1)*ceil(N/G))),
    with the block size rounded up to `align` rows."""
    per = -(-n_rows // world)
    per = -(-per // align) * align
    lo = min(n_rows, rank * per)
    hi = min(n_rows, (rank + 1) * per)
    return lo, hi


def merge_hits_host(gathered: np.ndarray, k: int) -> np.ndarray:
    """gathered: int64[m, 2] (key, frame_id bits). Returns the k smallest keys ascending, padded."""
    g = np.asarray(gathered).reshape(-1, 2)
    keys = g[:, 0]
    order = np.argsort(keys, kind="stable")[:k]
    out = g[order].copy()
    if out.shape[0] < k:
        pad = np.empty((k - out.shape[0], 2), dtype=np.int64)
        pad[:, 0] = KEY_PAD
        pad[:, 1] = -1
        out = np.concatenate([out, pad], axis=0)
    return out


def all_gather_hits(local_hits, world: int):
    """All-gather a [kpad, 2] int64 tensor over the default process group -> [world*kpad, 2]."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local_hits
    out = torch.empty((world * local_hits.shape[0], 2), dtype=torch.int64, device=local_hits.device)
    if local_hits.is_cuda:
        dist.all_gather_into_tensor(out, local_hits)
    else:  # gloo has no all_gather_into_tensor for every torch build: use the list form
        parts = [torch.empty_like(local_hits) for _ in range(world)]
        dist.all_gather(parts, local_hits)
        out = torch.cat(parts, dim=0)
    return out


class ShardedSearcher:
    """Pipelined sharded search for one rank. submit() enqueues; collect() returns in FIFO order."""

    def __init__(self, engine: HIPVectorEngine, rank: int, world: int, topK: int, depth: int = 4,  # noqa: N803
                 n_streams: int = 2, host_merge: bool = False, exchange: str = "rccl"):
        """exchange="rccl": all-gather of device buffers over RCCL (the product path).
        exchange="host": per-shard hits are downloaded and all-gathered on the host through the default
        (gloo) group, then merged on the host — the control / fallback of SURVEY.md §8e, also what lets
        the N>1 code run where RCCL cannot (two ranks sharing one GPU in tests)."""
        import torch

        self.engine = engine
        self.rank, self.world = rank, world
        self.kpad = clampTopK(topK)
        self.topK = topK
        self.exchange = exchange
        self.host_merge = host_merge or exchange == "host"
        self.dev = torch.device("cuda", engine.device)
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(max(1, n_streams))]
        self.depth = max(1, depth)
        self.local = [torch.empty((self.kpad, 2), dtype=torch.int64, device=self.dev) for _ in range(self.depth)]
        self.gathered = [torch.empty((self.kpad * world, 2), dtype=torch.int64, device=self.dev)
                         for _ in range(self.depth)]
        self.merged = [torch.empty((self.kpad, 2), dtype=torch.int64, device=self.dev) for _ in range(self.depth)]
        n_host = self.kpad * (world if (self.host_merge and exchange == "rccl") else 1)
        self.host = [torch.empty((n_host, 2), dtype=torch.int64).pin_memory() for _ in range(self.depth)]
        self.events = [torch.cuda.Event() for _ in range(self.depth)]
        # The library chains the scan kernels across the streams (they never overlap each other: each
        # owns the whole HBM pipe); the merge / all-gather / download of query i overlap the scan of i+1.
        self.inflight: Deque[int] = deque()
        self.seq = 0

    def submit(self, query) -> None:
        import torch
        import torch.distributed as dist

        if len(self.inflight) >= self.depth:
            raise RuntimeError("pipeline full: collect() first")
        b = self.seq % self.depth
        st = self.streams[self.seq % len(self.streams)]
        self.seq += 1
        with torch.cuda.stream(st):
            self.engine.searchShardDevice(query, self.topK, self.local[b].data_ptr(), st.cuda_stream)
            if self.world > 1 and self.exchange == "rccl":
                dist.all_gather_into_tensor(self.gathered[b], self.local[b])  # RCCL over xGMI
                src = self.gathered[b]
            else:
                src = self.local[b]
            if self.host_merge:
                self.host[b].copy_(src, non_blocking=True)
            else:
                if self.world > 1:
                    HIPVectorEngine.mergeHitsDevice(src.data_ptr(), src.shape[0], self.kpad, self.merged[b].data_ptr(),
                                                    st.cuda_stream)
                    src = self.merged[b]
                self.host[b].copy_(src, non_blocking=True)
            self.events[b].record(st)
        self.inflight.append(b)

    def collect(self) -> Tuple[np.ndarray, np.ndarray]:
        b = self.inflight.popleft()
        self.events[b].synchronize()
        hits = self.host[b].numpy()
        if self.world > 1 and self.exchange == "host":
            hits = all_gather_hits(self.host[b], self.world).numpy()  # gloo, CPU tensors
        if self.host_merge and self.world > 1:
            hits = merge_hits_host(hits, self.kpad)
        return HIPVectorEngine.hitsToResults(self.engine.metric, hits)

    def search(self, query) -> Tuple[np.ndarray, np.ndarray]:
        self.submit(query)
        return self.collect()


# ---------------------------------------------------------------------------
# batched queries over a row-sharded corpus (BASELINE config 5 shape)

def decode_hits(metric: VectorMetric, hits: np.ndarray):
    """Vectorised twin of wax_hip_hits_to_results for int64[..., k, 2] hit arrays: returns
    (frame_ids u64[..., k], scores f32[..., k], valid bool[..., k]); padded / non-finite entries are
    invalid (MetalVectorEngine.swift:597) and sort last because keys ascend."""
    h = np.asarray(hits, dtype=np.int64)
    keys = h[..., 0]
    o = (keys >> 32).astype(np.int32)
    bits = o ^ ((o >> 31) & np.int32(0x7FFFFFFF))
    dist = bits.view(np.float32)
    valid = (keys != KEY_PAD) & np.isfinite(dist) & (h[..., 1] != -1)
    scores = (np.float32(1.0) - dist) if VectorMetric(metric) is VectorMetric.cosine else -dist
    return h[..., 1].view(np.uint64), scores.astype(np.float32), valid


def merge_batch_hits_host(gathered: np.ndarray, k: int) -> np.ndarray:
    """gathered: int64[world, nq, kpad, 2] -> int64[nq, k, 2]: per query the k smallest keys over all shards."""
    g = np.asarray(gathered, dtype=np.int64)
    world, nq, kpad, _ = g.shape
    flat = np.transpose(g, (1, 0, 2, 3)).reshape(nq, world * kpad, 2)
    order = np.argsort(flat[:, :, 0], axis=1, kind="stable")[:, :k]
    return np.take_along_axis(flat, order[:, :, None], axis=1)


def sharded_search_batch(engine: HIPVectorEngine, queries, topK: int, world: int, exchange: str = "rccl"):  # noqa: N803
    """Every rank scans ITS shard for all queries (bf16 MFMA path + exact re-score inside the engine),
    the per-shard top-k hits are all-gathered (nq*kpad*16 bytes per rank) and merged per query by key
    = (distance asc, GLOBAL row asc): the answer is identical at every shard count.

    exchange="rccl": device-resident end to end — the queries go up once, the shard's hits stay in HBM
    (wax_hip_search_batch_hits_device), RCCL all-gathers them, one workgroup per query merges world*kpad -> kpad, and
    only the merged [nq][kpad] hits come down. `queries` may also be a CUDA tensor already in HBM."""
    import torch
    import torch.distributed as dist

    kpad = clampTopK(topK)
    if exchange == "rccl":
        dev = torch.device("cuda", engine.device)
        q = queries if isinstance(queries, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32))
        q = q.to(device=dev, dtype=torch.float32).contiguous()
        nq = int(q.shape[0])
        st = torch.cuda.current_stream(dev)
        local = torch.empty((nq, kpad, 2), dtype=torch.int64, device=dev)
        engine.searchBatchHitsDevice(q.data_ptr(), nq, topK, local.data_ptr(), kpad, st.cuda_stream)
        if world == 1:
            return decode_hits(engine.metric, local.cpu().numpy())
        out = torch.empty((world,) + tuple(local.shape), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out.view(-1), local.view(-1))   # RCCL over xGMI
        merged = torch.empty((nq, kpad, 2), dtype=torch.int64, device=dev)
        HIPVectorEngine.mergeBatchHitsDevice(out.data_ptr(), world, nq, kpad, kpad, merged.data_ptr(), st.cuda_stream)
        return decode_hits(engine.metric, merged.cpu().numpy())
    hits, _ = engine.searchBatchHits(queries, topK)
    nq, kcap, _ = hits.shape
    if kcap < kpad:  # a shard smaller than k: pad its lists
        pad = np.empty((nq, kpad - kcap, 2), dtype=np.int64)
        pad[:, :, 0] = KEY_PAD
        pad[:, :, 1] = -1
        hits = np.concatenate([hits, pad], axis=1)
    if world > 1:
        local = torch.from_numpy(np.ascontiguousarray(hits))
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        gathered = torch.stack(parts, dim=0).numpy()
        hits = merge_batch_hits_host(gathered, kpad)
    return decode_hits(engine.metric, hits)


class ShardedBatchSearcher:
    """Pipelined batched search for one rank of a row-sharded corpus (BASELINE config 5's launch shape under torchrun):
    submit() enqueues the batch on this rank's shard (wax_hip_search_batch_submit_device: returns at once), collect()
    finishes the oldest batch — the engine's own collect (certificates, exact re-runs), then ONE all-gather of the
    [nq][kpad] hits per rank over RCCL and a per-query merge by key on the device. With two batches in flight the
    all-gather and merge of batch i run while the GEMM of batch i + 1 already occupies the GPU. Collectives are issued in
    submission order on every rank. exchange="host": hits are downloaded, all-gathered through the default (gloo) group and
    merged on the host (control path; what lets test ranks share one GPU)."""

    def __init__(self, engine: HIPVectorEngine, rank: int, world: int, topK: int, nq: int, depth: int = 2,  # noqa: N803
                 exchange: str = "rccl"):
        import torch

        self.engine, self.rank, self.world, self.topK, self.nq = engine, rank, world, topK, int(nq)
        self.kpad = clampTopK(topK)
        self.exchange = exchange
        self.depth = max(1, depth)
        self.dev = torch.device("cuda", engine.device)
        shape = (self.nq, self.kpad, 2)
        self.local = [torch.empty(shape, dtype=torch.int64, device=self.dev) for _ in range(self.depth)]
        self.gathered = [torch.empty((world,) + shape, dtype=torch.int64, device=self.dev) for _ in range(self.depth)] if world > 1 else None
        self.merged = [torch.empty(shape, dtype=torch.int64, device=self.dev) for _ in range(self.depth)]
        self.comm_stream = torch.cuda.Stream(device=self.dev)
        self.tickets: Deque[Tuple[int, int]] = deque()
        self.seq = 0

    def pending(self) -> int:
        return len(self.tickets)

    def submit(self, d_queries) -> None:
        """d_queries: CUDA f32 tensor [nq, dims] on this rank's GPU; must stay untouched until the matching collect()."""
        import torch

        if len(self.tickets) >= self.depth:
            raise RuntimeError("pipeline full: collect() first")
        assert d_queries.is_cuda and int(d_queries.shape[0]) == self.nq
        b = self.seq % self.depth
        self.seq += 1
        st = torch.cuda.current_stream(self.dev)
        t = self.engine.searchBatchSubmitDevice(d_queries.data_ptr(), self.nq, self.topK, self.local[b].data_ptr(), self.kpad,
                                                st.cuda_stream)
        self.tickets.append((t, b))

    def collect(self):
        """Merged hits of the oldest batch: an int64 CUDA tensor [nq, kpad, 2] (key, frame id), complete on return
        (exchange="host": a numpy array)."""
        import torch
        import torch.distributed as dist

        t, b = self.tickets.popleft()
        self.engine.searchBatchCollectDevice(t)          # this rank's hits are complete in self.local[b]
        if self.world == 1:
            return self.local[b]
        if self.exchange == "rccl":
            with torch.cuda.stream(self.comm_stream):
                dist.all_gather_into_tensor(self.gathered[b].view(-1), self.local[b].view(-1))   # RCCL over xGMI
                HIPVectorEngine.mergeBatchHitsDevice(self.gathered[b].data_ptr(), self.world, self.nq, self.kpad, self.kpad,
                                                     self.merged[b].data_ptr(), self.comm_stream.cuda_stream)
            self.comm_stream.synchronize()
            return self.merged[b]
        local = self.local[b].cpu()
        parts = [torch.empty_like(local) for _ in range(self.world)]
        dist.all_gather(parts, local)
        return merge_batch_hits_host(torch.stack(parts, dim=0).numpy(), self.kpad)
