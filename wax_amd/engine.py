"""HIPVectorEngine — host-side mirror of the reference's `VectorSearchEngine` protocol
(Sources/WaxVectorSearch/VectorSearchEngine.swift:10-18) and of `MetalVectorEngine`'s concrete
surface (MetalVectorEngine.swift:144-146, 153, 318, 404, 682, 716), bound to libwaxhip's C ABI.

Method and argument names follow the Swift API (`search(vector:topK:)`, `addBatch(frameIds:vectors:)`
...) so the parity tests read like the reference's own tests. Everything that computes runs in the
HIP library; this file only marshals arguments. No CPU fallback exists.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .errors import EncodingError, InvalidToc, raise_for_status
from .vector_metric import VectorMetric

MAX_EMBEDDING_DIMENSIONS = 1_000_000  # Constants.maxEmbeddingDimensions (WaxCore/Constants.swift:51)
MAX_RESULTS = 10_000                  # MetalVectorEngine.maxResults (:18)


def _as_f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u64p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def clampTopK(topK: int) -> int:  # noqa: N802,N803 — MetalVectorEngine.clampTopK (:842-846)
    if topK < 1:
        return 1
    if topK > MAX_RESULTS:
        return MAX_RESULTS
    return int(topK)


class BufferPoolStats:
    """MetalVectorEngine.BufferPoolStats (:43-46)."""

    def __init__(self, transientAllocations: int, reuseCount: int):  # noqa: N803
        self.transientAllocations = transientAllocations
        self.reuseCount = reuseCount


class HIPVectorEngine:
    """MI355X brute-force vector engine (cosine / dot / l2) behind the VectorSearchEngine protocol."""

    # -- availability -------------------------------------------------------
    @staticmethod
    def isAvailable() -> bool:  # noqa: N802 — MetalVectorEngine.isAvailable (:144-146)
        return bool(_abi.lib().wax_hip_available())

    # -- lifecycle ----------------------------------------------------------
    def __init__(self, metric: VectorMetric = VectorMetric.cosine, dimensions: int = 0, device: int = -1,
                 devices: Optional[Sequence[int]] = None):
        """MetalVectorEngine.init(metric:dimensions:) (:153-274). `devices=[...]`: one engine row-sharded over several
        GPUs inside the library (wax_hip_engine_create_sharded): same API, same results."""
        self._lib = _abi.lib()
        self._h = ctypes.c_void_p()
        self.metric = VectorMetric(metric)
        self._dirty = False
        if dimensions < 0:
            raise InvalidToc("dimensions must be > 0")
        if devices is not None:
            devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._lib.wax_hip_engine_create_sharded(int(self.metric), int(min(dimensions, 2**32 - 1)), devs, len(devices),
                                                         ctypes.byref(self._h))
        else:
            rc = self._lib.wax_hip_engine_create(int(self.metric), int(min(dimensions, 2**32 - 1)), int(device),
                                                 ctypes.byref(self._h))
        if rc != _abi.OK and rc == _abi.ERR_INVALID_ARGUMENT:
            raise InvalidToc(_abi.last_error())  # "dimensions must be > 0" is invalidToc in the reference (:155)
        raise_for_status(rc)
        self.dimensions = int(self._lib.wax_hip_dimensions(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.wax_hip_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @classmethod
    def load(cls, wax, metric: VectorMetric, dimensions: int, device: int = -1) -> "HIPVectorEngine":
        """MetalVectorEngine.load(from:metric:dimensions:) (:318-328): committed bytes, then pending WAL embeddings."""
        engine = cls(metric=metric, dimensions=dimensions, device=device)
        data = wax.readCommittedVecIndexBytes()
        if data is not None:
            engine.deserialize(data)
        for emb in wax.pendingEmbeddingMutations():
            engine.add(frameId=emb.frameId, vector=emb.vector)
        return engine

    # -- properties ---------------------------------------------------------
    @property
    def count(self) -> int:
        return int(self._lib.wax_hip_count(self._h))

    @property
    def device(self) -> int:
        return int(self._lib.wax_hip_device_of(self._h))

    @property
    def shardCount(self) -> int:  # noqa: N802
        return int(self._lib.wax_hip_shard_count(self._h))

    def shardInfo(self, shard: int) -> Tuple[int, int, int]:  # noqa: N802
        """(device ordinal, first global row, rows) of one shard."""
        dev = ctypes.c_int(0)
        base = ctypes.c_uint64(0)
        rows = ctypes.c_uint64(0)
        raise_for_status(self._lib.wax_hip_shard_info(self._h, int(shard), ctypes.byref(dev), ctypes.byref(base), ctypes.byref(rows)))
        return int(dev.value), int(base.value), int(rows.value)

    # -- mutation -----------------------------------------------------------
    def add(self, frameId: int, vector) -> None:  # noqa: N803
        """add(frameId:vector:) (:330-357): upsert by frame id."""
        v = _as_f32(vector).reshape(-1)
        rc = self._lib.wax_hip_add(self._h, int(frameId), _fp(v), v.size)
        raise_for_status(rc)
        self._dirty = True

    def addBatch(self, frameIds: Sequence[int], vectors) -> None:  # noqa: N802,N803
        """addBatch(frameIds:vectors:) (:359-402)."""
        n = len(frameIds)
        if n == 0:
            return  # :360
        if isinstance(vectors, np.ndarray) and vectors.ndim == 2:
            if vectors.shape[0] != n:
                raise EncodingError("addBatch: frameIds.count != vectors.count")
            mat = _as_f32(vectors)
            width = mat.shape[1]
        else:
            if len(vectors) != n:
                raise EncodingError("addBatch: frameIds.count != vectors.count")  # :361-363
            for vec in vectors:  # :367-370 — every vector validated before anything is written
                if len(vec) != self.dimensions:
                    raise EncodingError(f"vector dimension mismatch: expected {self.dimensions}, got {len(vec)}")
            mat = _as_f32(vectors).reshape(n, self.dimensions)
            width = self.dimensions
        ids = np.ascontiguousarray(frameIds, dtype=np.uint64)
        rc = self._lib.wax_hip_add_batch(self._h, _u64p(ids), _fp(mat), n, width)
        raise_for_status(rc)
        self._dirty = True

    def addBatchStreaming(self, frameIds: Sequence[int], vectors, chunkSize: int = 256) -> None:  # noqa: N802,N803
        """addBatchStreaming(frameIds:vectors:chunkSize:) (:404-421)."""
        n = len(frameIds)
        if n == 0:
            return
        if len(vectors) != n:
            raise EncodingError("addBatchStreaming: frameIds.count != vectors.count")
        if n <= chunkSize:
            return self.addBatch(frameIds, vectors)
        for start in range(0, n, chunkSize):
            end = min(start + chunkSize, n)
            self.addBatch(frameIds[start:end], vectors[start:end])

    def addBatchDevice(self, frameIds, rows) -> None:  # noqa: N802,N803
        """Append rows that already live in this GPU's HBM (a torch tensor [n, dims] f32, contiguous)."""
        ids = np.ascontiguousarray(frameIds, dtype=np.uint64)
        n = int(ids.size)
        if n == 0:
            return
        if tuple(rows.shape) != (n, self.dimensions):
            raise EncodingError(f"vector dimension mismatch: expected {self.dimensions}, got {int(rows.shape[-1])}")
        if not rows.is_contiguous() or str(rows.dtype) != "torch.float32":
            raise EncodingError("addBatchDevice: rows must be a contiguous float32 tensor")
        rc = self._lib.wax_hip_add_batch_device(self._h, _u64p(ids), ctypes.c_void_p(rows.data_ptr()), n,
                                                self.dimensions)
        raise_for_status(rc)
        self._dirty = True

    def applyPutEmbeddings(self, payloads: bytes) -> int:  # noqa: N802
        """Pending-embedding replay (UnifiedSearchEngineCache.swift:252-283): `payloads` is WAL putEmbedding entry
        payloads back to back (WALEntryCodec.swift:39-54); validated as a whole, applied as one addBatch.
        Returns the number of records applied."""
        data = bytes(payloads)
        applied = ctypes.c_uint64(0)
        rc = self._lib.wax_hip_apply_put_embeddings(self._h, data, len(data), ctypes.byref(applied))
        raise_for_status(rc)
        if applied.value:
            self._dirty = True
        return int(applied.value)

    @staticmethod
    def encodePutEmbedding(frameId: int, vector) -> bytes:  # noqa: N802,N803
        """WALEntryCodec.encode(.putEmbedding) (WALEntryCodec.swift:39-54): 0x04, u64 frameId, u32 dim, f32 LE."""
        v = np.ascontiguousarray(vector, dtype="<f4").reshape(-1)
        if v.size > 1_000_000:
            raise EncodingError("embedding dimension exceeds limit")
        return b"\x04" + int(frameId).to_bytes(8, "little") + int(v.size).to_bytes(4, "little") + v.tobytes()

    def remove(self, frameId: int) -> None:  # noqa: N803
        """remove(frameId:) (:423-444): order-preserving delete; unknown id is a no-op."""
        before = self.count
        rc = self._lib.wax_hip_remove(self._h, int(frameId))
        raise_for_status(rc)
        if self.count != before:
            self._dirty = True

    def reserve(self, rows: int) -> None:
        raise_for_status(self._lib.wax_hip_reserve(self._h, int(rows)))

    # -- search -------------------------------------------------------------
    def _result_capacity(self, topK: int) -> int:  # noqa: N803
        """Entries to allocate for one query's results. The row count is only a hint that keeps the arrays small on
        small engines (+64: room for rows added concurrently); the library is told the capacity and never writes
        past it, so a racing add can at worst truncate to the best `cap` results, never overflow."""
        return max(1, min(clampTopK(topK), self.count + 64))

    def searchArrays(self, vector, topK: int) -> Tuple[np.ndarray, np.ndarray]:  # noqa: N802,N803
        q = _as_f32(vector).reshape(-1)
        cap = self._result_capacity(topK)
        ids = np.empty(cap, dtype=np.uint64)
        scores = np.empty(cap, dtype=np.float32)
        got = ctypes.c_uint32(0)
        rc = self._lib.wax_hip_search(self._h, _fp(q), q.size, int(max(min(topK, 2**31 - 1), -2**31)), _u64p(ids),
                                      _fp(scores), cap, ctypes.byref(got))
        raise_for_status(rc)
        return ids[:got.value].copy(), scores[:got.value].copy()

    def searchFiltered(self, vector, topK: int, frameIds=None, minScore=None) -> Tuple[np.ndarray, np.ndarray]:  # noqa: N802,N803
        """The vector lane's candidate filters applied on the device (UnifiedSearch.swift:1241-1258): `frameIds` is
        FrameFilter.frameIds (allow-list; None = no list, empty = nothing allowed), `minScore` is SearchRequest.minScore.
        Returns the best topK among the ALLOWED frames (a pre-filter), best first, minus those scoring below minScore."""
        q = _as_f32(vector).reshape(-1)
        cap = self._result_capacity(topK)
        ids = np.empty(cap, dtype=np.uint64)
        scores = np.empty(cap, dtype=np.float32)
        got = ctypes.c_uint32(0)
        allow = None if frameIds is None else np.ascontiguousarray(list(frameIds) if not isinstance(frameIds, np.ndarray)
                                                                   else frameIds, dtype=np.uint64)
        rc = self._lib.wax_hip_search_filtered(
            self._h, _fp(q), q.size, int(max(min(topK, 2**31 - 1), -2**31)),
            0 if allow is None else 1, None if allow is None or allow.size == 0 else _u64p(allow),
            0 if allow is None else int(allow.size),
            0 if minScore is None else 1, 0.0 if minScore is None else float(minScore),
            _u64p(ids), _fp(scores), cap, ctypes.byref(got))
        raise_for_status(rc)
        return ids[:got.value].copy(), scores[:got.value].copy()

    def searchFilteredHits(self, vector, topK: int, allow) -> List[Tuple[int, float]]:  # noqa: N802,N803
        """searchFiltered as [(frameId, score)] (the shape the transcribed reference cases assert on)."""
        ids, scores = self.searchFiltered(vector, topK, frameIds=allow)
        return [(int(i), float(s)) for i, s in zip(ids, scores)]

    def search(self, vector, topK: int) -> List[Tuple[int, float]]:  # noqa: N803
        """VectorSearchEngine.search(vector:topK:) -> [(frameId, score)] best first (:446-627)."""
        ids, scores = self.searchArrays(vector, topK)
        return [(int(i), float(s)) for i, s in zip(ids, scores)]

    def submit(self, vector, topK: int) -> int:  # noqa: N803
        q = _as_f32(vector).reshape(-1)
        t = ctypes.c_uint64(0)
        rc = self._lib.wax_hip_search_submit(self._h, _fp(q), q.size, int(topK), ctypes.byref(t))
        raise_for_status(rc)
        return int(t.value)

    def collect(self, ticket: int, topK: int) -> Tuple[np.ndarray, np.ndarray]:  # noqa: N803
        """`topK` sizes the result arrays: pass what was given to submit() (a smaller value truncates to the best)."""
        cap = clampTopK(topK)   # the ticket was answered on the row count at submit time: size by topK alone
        ids = np.empty(cap, dtype=np.uint64)
        scores = np.empty(cap, dtype=np.float32)
        got = ctypes.c_uint32(0)
        rc = self._lib.wax_hip_search_collect(self._h, int(ticket), _u64p(ids), _fp(scores), cap, ctypes.byref(got))
        raise_for_status(rc)
        return ids[:got.value].copy(), scores[:got.value].copy()

    def searchBatch(self, vectors, topK: int):  # noqa: N802,N803
        qs = _as_f32(vectors)
        if qs.ndim != 2:
            raise EncodingError("searchBatch: vectors must be [nq, dims]")
        nq, width = qs.shape
        kcap = max(1, min(clampTopK(topK), max(self.count, 1)))   # row width of the arrays (passed as the stride)
        ids = np.zeros((nq, kcap), dtype=np.uint64)
        scores = np.zeros((nq, kcap), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        rc = self._lib.wax_hip_search_batch(self._h, _fp(qs), nq, width, int(topK), _u64p(ids), _fp(scores), kcap,
                                            counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
        raise_for_status(rc)
        return ids, scores, counts

    def searchBatchHits(self, vectors, topK: int):  # noqa: N802,N803
        """Raw ranked candidates: int64[nq, kcap, 2] of (key, frame_id bits), ascending key, KEY_PAD padded."""
        qs = _as_f32(vectors)
        if qs.ndim != 2:
            raise EncodingError("searchBatchHits: vectors must be [nq, dims]")
        nq, width = qs.shape
        kcap = max(1, min(clampTopK(topK), max(self.count, 1)))
        hits = np.empty((nq, kcap, 2), dtype=np.int64)
        hits[:, :, 0] = _abi.KEY_PAD
        hits[:, :, 1] = -1
        counts = np.zeros(nq, dtype=np.uint32)
        rc = self._lib.wax_hip_search_batch_hits(self._h, _fp(qs), nq, width, int(topK),
                                                 hits.ctypes.data_as(ctypes.POINTER(_abi.Hit)), kcap,
                                                 counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
        raise_for_status(rc)
        return hits, counts

    def searchBatchHitsDevice(self, d_queries_ptr: int, nq: int, topK: int, d_out_hits_ptr: int, out_stride: int,  # noqa: N802,N803
                              stream: int = 0) -> None:
        """Device-resident batch search: queries [nq, dims] f32 and the output [nq, out_stride] wax_hip_hit both live
        in this GPU's HBM (raw device pointers, e.g. torch tensors' data_ptr()); blocking; `stream` is the stream that
        produced the queries. Results are complete on return."""
        rc = self._lib.wax_hip_search_batch_hits_device(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), self.dimensions,
                                                        int(topK), ctypes.c_void_p(d_out_hits_ptr), int(out_stride),
                                                        ctypes.c_void_p(stream))
        raise_for_status(rc)

    def searchBatchSubmitDevice(self, d_queries_ptr: int, nq: int, topK: int, d_out_hits_ptr: int, out_stride: int,  # noqa: N802,N803
                                stream: int = 0) -> int:
        """Pipelined form of searchBatchHitsDevice: enqueues the batch and returns a ticket at once; the buffers must
        stay untouched until searchBatchCollectDevice(ticket)."""
        t = ctypes.c_uint64(0)
        rc = self._lib.wax_hip_search_batch_submit_device(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), self.dimensions,
                                                          int(topK), ctypes.c_void_p(d_out_hits_ptr), int(out_stride),
                                                          ctypes.c_void_p(stream), ctypes.byref(t))
        raise_for_status(rc)
        return int(t.value)

    def searchBatchCollectDevice(self, ticket: int) -> int:  # noqa: N802
        """Waits for a submitted batch (uncertified queries are re-run exactly, in place); returns how many were."""
        fb = ctypes.c_uint32(0)
        raise_for_status(self._lib.wax_hip_search_batch_collect_device(self._h, int(ticket), ctypes.byref(fb)))
        return int(fb.value)

    # -- sharded search (one engine per GPU; exchange over RCCL by the caller) ----
    def setRowBase(self, rowBase: int) -> None:  # noqa: N802,N803
        raise_for_status(self._lib.wax_hip_set_row_base(self._h, int(rowBase)))

    def searchShardDevice(self, vector, topK: int, out_hits_ptr: int, stream: int = 0) -> None:  # noqa: N802,N803
        q = _as_f32(vector).reshape(-1)
        rc = self._lib.wax_hip_search_shard_device(self._h, _fp(q), q.size, int(topK), ctypes.c_void_p(out_hits_ptr),
                                                   ctypes.c_void_p(stream))
        raise_for_status(rc)

    @staticmethod
    def mergeBatchHitsDevice(in_ptr: int, n_shards: int, nq: int, k_in: int, k: int, out_ptr: int, stream: int = 0) -> None:  # noqa: N802
        """[n_shards][nq][k_in] gathered hits (device) -> [nq][k] merged hits (device), one workgroup per query."""
        rc = _abi.lib().wax_hip_merge_batch_hits_device(ctypes.c_void_p(in_ptr), int(n_shards), int(nq), int(k_in), int(k),
                                                        ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream))
        raise_for_status(rc)

    @staticmethod
    def mergeHitsDevice(in_ptr: int, n: int, k: int, out_ptr: int, stream: int = 0) -> None:  # noqa: N802
        rc = _abi.lib().wax_hip_merge_hits_device(ctypes.c_void_p(in_ptr), int(n), int(k), ctypes.c_void_p(out_ptr),
                                                  ctypes.c_void_p(stream))
        raise_for_status(rc)

    @staticmethod
    def hitsToResults(metric: VectorMetric, hits: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:  # noqa: N802
        """hits: structured/2-column array of (int64 key, uint64 frame_id) pairs, shape [n, 2] viewed as int64."""
        raw = np.ascontiguousarray(hits).view(np.int64).reshape(-1, 2)
        n = raw.shape[0]
        ids = np.empty(max(n, 1), dtype=np.uint64)
        scores = np.empty(max(n, 1), dtype=np.float32)
        got = ctypes.c_uint32(0)
        rc = _abi.lib().wax_hip_hits_to_results(int(metric), raw.ctypes.data_as(ctypes.POINTER(_abi.Hit)), n,
                                                _u64p(ids), _fp(scores), ctypes.byref(got))
        raise_for_status(rc)
        return ids[:got.value].copy(), scores[:got.value].copy()

    # -- persistence --------------------------------------------------------
    def serialize(self) -> bytes:
        """serialize() (:682-714): "MV2V" segment, encoding 2."""
        out = ctypes.POINTER(ctypes.c_uint8)()
        length = ctypes.c_size_t(0)
        rc = self._lib.wax_hip_serialize(self._h, ctypes.byref(out), ctypes.byref(length))
        raise_for_status(rc)
        try:
            return ctypes.string_at(out, length.value)
        finally:
            self._lib.wax_hip_free(out)

    def deserialize(self, data: bytes) -> None:
        """deserialize(_:) (:716-815)."""
        data = bytes(data)
        rc = self._lib.wax_hip_deserialize(self._h, data, len(data))
        raise_for_status(rc)
        self._dirty = False  # :813

    def stageForCommit(self, into) -> None:  # noqa: N802 — stageForCommit(into:) (:818-828)
        if not self._dirty:
            return
        blob = self.serialize()
        into.stageVecIndexForNextCommit(bytes=blob, vectorCount=self.count, dimension=self.dimensions,
                                        similarity=self.metric.toVecSimilarity())
        self._dirty = False

    # -- observability ------------------------------------------------------
    def stats(self) -> _abi.Stats:
        st = _abi.Stats()
        raise_for_status(self._lib.wax_hip_stats(self._h, ctypes.byref(st)))
        return st

    def debugBufferPoolStats(self) -> BufferPoolStats:  # noqa: N802 — (:119-121)
        st = self.stats()
        return BufferPoolStats(int(st.transient_allocations), int(st.reuse_count))

    def setTuning(self, key: str, value: int) -> None:  # noqa: N802
        raise_for_status(self._lib.wax_hip_set_tuning(self._h, key.encode(), int(value)))

    def getTuning(self, key: str) -> int:  # noqa: N802
        return int(self._lib.wax_hip_get_tuning(self._h, key.encode()))

    def timeScanKernel(self, vector, topK: int, iters: int) -> float:  # noqa: N802,N803
        q = _as_f32(vector).reshape(-1)
        ms = ctypes.c_double(0.0)
        raise_for_status(self._lib.wax_hip_time_scan_kernel(self._h, _fp(q), q.size, int(topK), int(iters),
                                                            ctypes.byref(ms)))
        return float(ms.value)

    def timeStreamRead(self, iters: int) -> float:  # noqa: N802
        ms = ctypes.c_double(0.0)
        raise_for_status(self._lib.wax_hip_time_stream_read(self._h, int(iters), ctypes.byref(ms)))
        return float(ms.value)
