"""CPU oracle for the Wax vector scan + top-k path — TEST INFRASTRUCTURE ONLY.

Thin ctypes binding over ``oracle/wax_oracle.c`` (the C restatement of the
reference's arithmetic; every C function cites the reference file:line it
follows) plus the seeded synthetic-input generators SURVEY.md §8d prescribes.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / reported baseline. The
product (``wax_amd``) never imports it and has no CPU fallback.

Pinning: rank/membership/tolerance cases from the reference's own tests are
pinned through ``tests/golden/reference_cases.json``; the numeric USearch
boundary is *parity unpinned* (see the header of ``wax_oracle.c``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwaxoracle.so")

METRIC_COSINE, METRIC_DOT, METRIC_L2 = 0, 1, 2
MODE_TRUTH_F64, MODE_METAL_F32 = 0, 1

# SURVEY.md §8d seeds
CORPUS_SEED = 20260220
QUERY_SEED = 7


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "wax_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "CC=gcc"], check=True)
    return _LIB_PATH


# ---------------------------------------------------------------------------
# oracle/_ref: the reference's own Metal compute shaders compiled as C++ from /root/reference and run on the CPU
# (oracle/ref_metal/). Only built where the reference checkout exists; tests that need it skip elsewhere — the committed
# fixtures under tests/golden/metal_shader_vectors.json carry its outputs to every machine.
REFERENCE_ROOT = os.environ.get("WAX_REFERENCE_ROOT", "/root/reference")
_REF_LIB_PATH = os.path.join(_HERE, "_ref", "libwaxref_metal.so")
_ref_lib = None


def build_ref(force: bool = False) -> Optional[str]:
    """Compile oracle/_ref/libwaxref_metal.so (recipe: oracle/ref_metal/Makefile). Returns None where the reference checkout
    is absent (the GPU box): nothing there may depend on it."""
    shaders = os.path.join(REFERENCE_ROOT, "Sources", "WaxVectorSearch", "Shaders", "CosineDistance.metal")
    if not os.path.exists(shaders):
        return _REF_LIB_PATH if os.path.exists(_REF_LIB_PATH) else None
    if force or not os.path.exists(_REF_LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(_HERE, "ref_metal"), "-s", "-B" if force else "-s", f"REF={REFERENCE_ROOT}"], check=True)
    return _REF_LIB_PATH


def ref_lib():
    """ctypes handle of the reference-shader library, or None when it is not available here."""
    global _ref_lib
    if _ref_lib is None:
        path = build_ref()
        if not path:
            return None
        L = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.waxref_metal_cosine_distances.restype = ctypes.c_int
        L.waxref_metal_cosine_distances.argtypes = [ctypes.c_int, f32p, f32p, ctypes.c_uint32, ctypes.c_uint32, f32p]
        L.waxref_metal_topk.restype = ctypes.c_int
        L.waxref_metal_topk.argtypes = [f32p, ctypes.c_uint32, ctypes.c_uint32, f32p, u32p, u32p]
        _ref_lib = L
    return _ref_lib


def ref_metal_distances(vectors, query, simd8: Optional[bool] = None) -> np.ndarray:
    """cosineDistanceKernelSIMD8 / SIMD4 of the reference (CosineDistance.metal:152-328) executed on the CPU, dispatched as
    MetalVectorEngine.search does (:494-507). simd8=None: the engine's rule, D >= 384 (:24, :185)."""
    L = ref_lib()
    if L is None:
        raise RuntimeError("oracle/_ref is not available here (no reference checkout, no prebuilt library)")
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    n, d = v.shape
    out = np.empty(n, dtype=np.float32)
    use8 = (d >= 384) if simd8 is None else bool(simd8)
    f32p = ctypes.POINTER(ctypes.c_float)
    rc = L.waxref_metal_cosine_distances(int(use8), v.ctypes.data_as(f32p), q.ctypes.data_as(f32p), n, d, out.ctypes.data_as(f32p))
    assert rc == 0
    return out


def ref_metal_topk(dist, k: int):
    """The GPU top-k of the reference (TopKReduction.metal kernels under the dispatch loop of MetalVectorEngine.swift:517-585).
    Returns (indices u32[k], distances f32[k], passes), or raises NonTermination where the reference's host loop would never
    finish (k > 128 with enough rows: see oracle/ref_metal/ref_metal.cpp)."""
    L = ref_lib()
    if L is None:
        raise RuntimeError("oracle/_ref is not available here (no reference checkout, no prebuilt library)")
    d = np.ascontiguousarray(dist, dtype=np.float32)
    od = np.empty(k, dtype=np.float32)
    oi = np.empty(k, dtype=np.uint32)
    passes = ctypes.c_uint32()
    rc = L.waxref_metal_topk(d.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), d.shape[0], k, od.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                             oi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(passes))
    if rc == -2:
        raise NonTermination(f"the reference's reduction loop makes no progress for n = {d.shape[0]}, k = {k}")
    if rc != 0:
        raise ValueError(f"waxref_metal_topk: {rc}")
    return oi, od, int(passes.value)


class NonTermination(Exception):
    pass


def formula_rows(seed: int, n: int, dims: int) -> np.ndarray:
    """Deterministic f32 test rows from integer arithmetic only (no RNG stream, no BLAS): identical on every machine, so the
    reference-shader fixture stores OUTPUTS only. Values in (-1, 1) / sqrt(dims)-ish scale, never all zero."""
    i = np.arange(n, dtype=np.int64)[:, None]
    j = np.arange(dims, dtype=np.int64)[None, :]
    v = ((i * 7919 + j * 104729 + (i * j) * 31 + int(seed) * 15485863) % 20011 - 10005).astype(np.float64) / 10005.0
    return (v / np.sqrt(np.float64(max(dims, 1)))).astype(np.float32)


def formula_unit_query(seed: int, dims: int) -> np.ndarray:
    """Deterministic unit-norm f32 query: formula values normalised in f64 with a sequential sum."""
    j = np.arange(dims, dtype=np.int64)
    v = ((j * 48271 + int(seed) * 2147483 + 12345) % 10007 - 5003).astype(np.float64) / 5003.0
    norm = np.sqrt(np.cumsum(v * v)[-1])
    return (v / norm).astype(np.float32)


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        i64p = ctypes.POINTER(ctypes.c_int64)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.wax_oracle_clamp_topk.restype = ctypes.c_int32
        L.wax_oracle_clamp_topk.argtypes = [ctypes.c_int64]
        L.wax_oracle_score_from_distance.restype = ctypes.c_float
        L.wax_oracle_score_from_distance.argtypes = [ctypes.c_int, ctypes.c_float]
        L.wax_oracle_magnitude.restype = ctypes.c_float
        L.wax_oracle_magnitude.argtypes = [f32p, ctypes.c_uint32]
        L.wax_oracle_normalize_l2.restype = None
        L.wax_oracle_normalize_l2.argtypes = [f32p, ctypes.c_uint32, f32p]
        L.wax_oracle_is_normalized_l2.restype = ctypes.c_int
        L.wax_oracle_is_normalized_l2.argtypes = [f32p, ctypes.c_uint32, ctypes.c_float]
        L.wax_oracle_cosine_distances_metal.restype = None
        L.wax_oracle_cosine_distances_metal.argtypes = [f32p, f32p, ctypes.c_uint64, ctypes.c_uint32, f32p]
        L.wax_oracle_distances_f64.restype = None
        L.wax_oracle_distances_f64.argtypes = [ctypes.c_int, f32p, f32p, ctypes.c_uint64, ctypes.c_uint32, f32p]
        for name in ("wax_oracle_topk_heap", "wax_oracle_topk_sort", "wax_oracle_topk_total"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int64
            fn.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, i64p, f32p]
        L.wax_oracle_search.restype = ctypes.c_int64
        L.wax_oracle_search.argtypes = [ctypes.c_int, ctypes.c_int, f32p, u64p, ctypes.c_uint64, ctypes.c_uint32,
                                        f32p, ctypes.c_uint32, ctypes.c_int64, u64p, f32p, f32p, i64p]
        L.wax_oracle_max_threads.restype = ctypes.c_int
        L.wax_oracle_scan_topk_mt.restype = ctypes.c_int64
        L.wax_oracle_scan_topk_mt.argtypes = [ctypes.c_int, f32p, ctypes.c_uint64, ctypes.c_uint32, f32p,
                                              ctypes.c_int64, ctypes.c_int, i64p, f32p]
        L.wax_oracle_scan_topk_fast.restype = ctypes.c_int64
        L.wax_oracle_scan_topk_fast.argtypes = L.wax_oracle_scan_topk_mt.argtypes
        L.wax_oracle_search_batch.restype = ctypes.c_int64
        L.wax_oracle_search_batch.argtypes = [ctypes.c_int, f32p, ctypes.c_uint64, ctypes.c_uint32, f32p, ctypes.c_uint32,
                                              ctypes.c_int64, i64p, f32p, i64p]
        L.wax_oracle_first_touch_rows.restype = None
        L.wax_oracle_first_touch_rows.argtypes = [f32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
        L.wax_oracle_copy_rows.restype = None
        L.wax_oracle_copy_rows.argtypes = [f32p, f32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
        L.wax_oracle_mv2v_size.restype = ctypes.c_uint64
        L.wax_oracle_mv2v_size.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
        L.wax_oracle_mv2v_serialize.restype = ctypes.c_uint64
        L.wax_oracle_mv2v_serialize.argtypes = [ctypes.c_int, f32p, u64p, ctypes.c_uint64, ctypes.c_uint32, u8p]
        L.wax_oracle_mv2v_parse.restype = ctypes.c_int
        L.wax_oracle_mv2v_parse.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, u64p, u64p, u64p]
        L.wax_oracle_deterministic_embed.restype = None
        L.wax_oracle_deterministic_embed.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, f32p]
        L.wax_oracle_tie_pattern.restype = None
        L.wax_oracle_tie_pattern.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, f32p]
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


# --------------------------------------------------------------------------
# reference-semantics helpers

def clamp_topk(top_k: int) -> int:
    return int(lib().wax_oracle_clamp_topk(int(top_k)))


def score_from_distance(metric: int, d: float) -> float:
    return float(lib().wax_oracle_score_from_distance(metric, ctypes.c_float(d)))


def normalize_l2(v) -> np.ndarray:
    v = _f32(v)
    out = np.empty_like(v)
    lib().wax_oracle_normalize_l2(_p(v, ctypes.c_float), v.size, _p(out, ctypes.c_float))
    return out


def is_normalized_l2(v, tolerance: float = 1e-3) -> bool:
    v = _f32(v)
    return bool(lib().wax_oracle_is_normalized_l2(_p(v, ctypes.c_float), v.size, ctypes.c_float(tolerance)))


def distances(metric: int, vectors, query, mode: int = MODE_TRUTH_F64) -> np.ndarray:
    vectors = _f32(vectors)
    query = _f32(query)
    n, d = vectors.shape
    out = np.empty(n, dtype=np.float32)
    if mode == MODE_METAL_F32:
        assert metric == METRIC_COSINE
        lib().wax_oracle_cosine_distances_metal(_p(vectors, ctypes.c_float), _p(query, ctypes.c_float), n, d,
                                                _p(out, ctypes.c_float))
    else:
        lib().wax_oracle_distances_f64(metric, _p(vectors, ctypes.c_float), _p(query, ctypes.c_float), n, d,
                                       _p(out, ctypes.c_float))
    return out


def topk_heap(dist, k: int, use_sort: bool = False, total: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Reference heap (default), full (distance,index) sort (use_sort) or total-order heap (total)."""
    dist = _f32(dist)
    m = max(0, min(int(k), dist.size))
    idx = np.empty(max(m, 1), dtype=np.int64)
    dd = np.empty(max(m, 1), dtype=np.float32)
    fn = lib().wax_oracle_topk_sort if use_sort else (lib().wax_oracle_topk_total if total else lib().wax_oracle_topk_heap)
    got = fn(_p(dist, ctypes.c_float), dist.size, int(k), _p(idx, ctypes.c_int64), _p(dd, ctypes.c_float))
    return idx[:got].copy(), dd[:got].copy()


class DimensionMismatch(Exception):
    pass


def search(metric: int, vectors, frame_ids, query, top_k: int, mode: int = MODE_TRUTH_F64):
    """Oracle for VectorSearchEngine.search(vector:topK:).

    Returns (frame_ids u64[m], scores f32[m], distances f32[m], rows i64[m]),
    best first under (distance asc, row asc).
    """
    vectors = _f32(vectors)
    query = _f32(query)
    if vectors.ndim != 2:
        raise ValueError("vectors must be [n, d]")
    n, d = vectors.shape
    ids = None if frame_ids is None else np.ascontiguousarray(frame_ids, dtype=np.uint64)
    cap = max(1, min(clamp_topk(top_k), max(n, 1)))
    out_ids = np.empty(cap, dtype=np.uint64)
    out_scores = np.empty(cap, dtype=np.float32)
    out_d = np.empty(cap, dtype=np.float32)
    out_rows = np.empty(cap, dtype=np.int64)
    got = lib().wax_oracle_search(metric, mode, _p(vectors, ctypes.c_float),
                                  None if ids is None else _p(ids, ctypes.c_uint64), n, d,
                                  _p(query, ctypes.c_float), query.size, int(top_k),
                                  _p(out_ids, ctypes.c_uint64), _p(out_scores, ctypes.c_float),
                                  _p(out_d, ctypes.c_float), _p(out_rows, ctypes.c_int64))
    if got < 0:
        raise DimensionMismatch(f"vector dimension mismatch: expected {d}, got {query.size}")
    return out_ids[:got].copy(), out_scores[:got].copy(), out_d[:got].copy(), out_rows[:got].copy()


def max_threads() -> int:
    return int(lib().wax_oracle_max_threads())


def scan_topk_mt(metric: int, vectors, query, top_k: int, threads: int):
    """The timed CPU baseline: f32 scan + per-thread heaps + merge."""
    vectors = _f32(vectors)
    query = _f32(query)
    n, d = vectors.shape
    cap = max(1, min(clamp_topk(top_k), max(n, 1)))
    idx = np.empty(cap, dtype=np.int64)
    dd = np.empty(cap, dtype=np.float32)
    got = lib().wax_oracle_scan_topk_mt(metric, _p(vectors, ctypes.c_float), n, d, _p(query, ctypes.c_float),
                                        int(top_k), int(threads), _p(idx, ctypes.c_int64), _p(dd, ctypes.c_float))
    return idx[:got].copy(), dd[:got].copy()


def scan_topk_fast(metric: int, vectors, query, top_k: int, threads: int):
    """Tuned CPU baseline: metric-specialised FMA inner loop, same selection as scan_topk_mt."""
    vectors = _f32(vectors)
    query = _f32(query)
    n, d = vectors.shape
    cap = max(1, min(clamp_topk(top_k), max(n, 1)))
    idx = np.empty(cap, dtype=np.int64)
    dd = np.empty(cap, dtype=np.float32)
    got = lib().wax_oracle_scan_topk_fast(metric, _p(vectors, ctypes.c_float), n, d, _p(query, ctypes.c_float),
                                          int(top_k), int(threads), _p(idx, ctypes.c_int64), _p(dd, ctypes.c_float))
    return idx[:got].copy(), dd[:got].copy()


def search_batch(metric: int, vectors, queries, top_k: int):
    """Oracle for a whole batch in one pass over the rows: per query the (distance asc, row asc) top-k with
    f64-accumulated distances bit-identical to `search` / `distances`. Returns (rows i64[nq, kk],
    distances f32[nq, kk], counts i64[nq]); scores = score_from_distance(metric, distance)."""
    vectors = _f32(vectors)
    queries = _f32(queries)
    n, d = vectors.shape
    nq, dq = queries.shape
    if dq != d:
        raise DimensionMismatch(f"vector dimension mismatch: expected {d}, got {dq}")
    kk = max(1, min(clamp_topk(top_k), max(n, 1)))
    rows = np.full((nq, kk), -1, dtype=np.int64)
    dist = np.full((nq, kk), np.inf, dtype=np.float32)
    counts = np.zeros(nq, dtype=np.int64)
    lib().wax_oracle_search_batch(metric, _p(vectors, ctypes.c_float), n, d, _p(queries, ctypes.c_float), nq,
                                  int(top_k), _p(rows, ctypes.c_int64), _p(dist, ctypes.c_float),
                                  _p(counts, ctypes.c_int64))
    return rows, dist, counts


def scores_from_distances(metric: int, dist) -> np.ndarray:
    """Vectorised VectorMetric.score(fromDistance:) (VectorMetric.swift:32-43)."""
    dist = np.asarray(dist, dtype=np.float32)
    s = (np.float32(1.0) - dist) if metric == METRIC_COSINE else -dist
    return np.where(np.isfinite(dist), s, np.float32(0.0)).astype(np.float32)


def numa_sample(n: int, d: int, threads: int) -> np.ndarray:
    """An [n, d] f32 array whose pages are first-touched by the threads that will scan them."""
    out = np.empty((n, d), dtype=np.float32)
    lib().wax_oracle_first_touch_rows(_p(out, ctypes.c_float), n, d, int(threads))
    return out


def copy_rows(dst: np.ndarray, src, threads: int) -> None:
    src = _f32(src)
    assert dst.shape == src.shape and dst.dtype == np.float32 and dst.flags.c_contiguous
    lib().wax_oracle_copy_rows(_p(dst, ctypes.c_float), _p(src, ctypes.c_float), dst.shape[0], dst.shape[1], int(threads))


def mv2v_serialize(metric: int, vectors, frame_ids) -> bytes:
    vectors = _f32(vectors)
    ids = np.ascontiguousarray(frame_ids, dtype=np.uint64)
    if vectors.ndim == 1:
        raise ValueError("vectors must be [n, d]")
    n, d = vectors.shape
    size = int(lib().wax_oracle_mv2v_size(n, d))
    buf = np.empty(size, dtype=np.uint8)
    wrote = lib().wax_oracle_mv2v_serialize(metric, _p(vectors, ctypes.c_float), _p(ids, ctypes.c_uint64), n, d,
                                            _p(buf, ctypes.c_uint8))
    assert wrote == size
    return buf.tobytes()


def mv2v_parse(data: bytes, expect_metric: int = -1, expect_dims: int = 0):
    """Returns (err_code, vectors or None, frame_ids or None). err_code 0 = ok."""
    arr = np.frombuffer(data, dtype=np.uint8)
    arr = np.ascontiguousarray(arr)
    count = ctypes.c_uint64(0)
    voff = ctypes.c_uint64(0)
    ioff = ctypes.c_uint64(0)
    if arr.size == 0:
        return 1, None, None
    rc = lib().wax_oracle_mv2v_parse(_p(arr, ctypes.c_uint8), arr.size, expect_metric, expect_dims,
                                     ctypes.byref(count), ctypes.byref(voff), ctypes.byref(ioff))
    if rc != 0:
        return rc, None, None
    dims = int(np.frombuffer(data[8:12], dtype="<u4")[0])
    n = count.value
    vec = np.frombuffer(data, dtype="<f4", count=n * dims, offset=voff.value).reshape(n, dims).copy()
    ids = np.frombuffer(data, dtype="<u8", count=n, offset=ioff.value).copy()
    return 0, vec, ids


# --------------------------------------------------------------------------
# seeded synthetic inputs

def deterministic_embed(text: str, dims: int, normalize: bool = True) -> np.ndarray:
    """Reference DeterministicEmbedder (RAGBenchmarkSupport.swift:114-157)."""
    raw = np.frombuffer(text.encode("utf-8"), dtype=np.uint8)
    raw = np.ascontiguousarray(raw) if raw.size else np.zeros(1, dtype=np.uint8)
    out = np.empty(dims, dtype=np.float32)
    lib().wax_oracle_deterministic_embed(_p(raw, ctypes.c_uint8), len(text.encode("utf-8")), dims, int(normalize),
                                         _p(out, ctypes.c_float))
    return out


def tie_pattern(row0: int, n: int, dims: int) -> np.ndarray:
    """MetalVectorEngineBenchmark.swift:33-38 pattern: ((i+d) % 256)/255."""
    out = np.empty((n, dims), dtype=np.float32)
    lib().wax_oracle_tie_pattern(row0, n, dims, _p(out, ctypes.c_float))
    return out


SHARD_ROWS = 65536  # generation granule: corpus is reproducible for any GPU count


def gaussian_unit_rows(row0: int, n: int, dims: int, seed: int = CORPUS_SEED) -> np.ndarray:
    """Unit-norm Gaussian rows [row0, row0+n): granule g = row // SHARD_ROWS is
    drawn from PCG64(seed + g), L2-normalised in f64, cast to f32."""
    out = np.empty((n, dims), dtype=np.float32)
    r = row0
    end = row0 + n
    while r < end:
        g = r // SHARD_ROWS
        g0 = g * SHARD_ROWS
        take_lo = r - g0
        take_hi = min(end - g0, SHARD_ROWS)
        rng = np.random.Generator(np.random.PCG64(seed + g))
        block = rng.standard_normal((take_hi, dims))  # rows of a granule are drawn in order
        block = block[take_lo:take_hi]
        block /= np.linalg.norm(block, axis=1, keepdims=True)
        out[r - row0:r - row0 + block.shape[0]] = block.astype(np.float32)
        r = g0 + take_hi
    return out


def gaussian_unit_queries(nq: int, dims: int, seed: int = QUERY_SEED) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    q = rng.standard_normal((nq, dims))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def rrf_fuse(lists, k: int = 60):
    """HybridSearch.rrfFusion(lists:k:) (HybridSearch.swift:25-52): lists = [(weight, [frameId, ...]), ...] ->
    (ids u64[m], scores f32[m], best_rank u32[m], sources u32[m]) sorted by (score desc, bestRank asc, id asc)."""
    weights = np.asarray([w for w, _ in lists], dtype=np.float32)
    parts = [np.asarray(ids, dtype=np.uint64).reshape(-1) for _, ids in lists]
    offsets = np.zeros(len(lists) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(p) for p in parts])
    flat = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
    total = int(offsets[-1])
    out_ids = np.empty(max(total, 1), dtype=np.uint64)
    out_scores = np.empty(max(total, 1), dtype=np.float32)
    out_rank = np.empty(max(total, 1), dtype=np.uint32)
    out_src = np.empty(max(total, 1), dtype=np.uint32)
    f = lib().wax_oracle_rrf_fuse
    f.restype = ctypes.c_int64
    f.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    flat = np.ascontiguousarray(flat)
    m = f(len(lists), weights.ctypes.data, flat.ctypes.data if total else None, offsets.ctypes.data, int(k),
          out_ids.ctypes.data, out_scores.ctypes.data, out_rank.ctypes.data, out_src.ctypes.data)
    return out_ids[:m].copy(), out_scores[:m].copy(), out_rank[:m].copy(), out_src[:m].copy()


def rrf_fuse_two(text_ids, vector_ids, k: int = 60, alpha: float = 0.5):
    """HybridSearch.rrfFusion(textResults:vectorResults:k:alpha:) (HybridSearch.swift:8-23): alpha clamped to [0, 1]."""
    a = np.float32(min(1.0, max(0.0, alpha)))
    return rrf_fuse([(a, text_ids), (np.float32(1.0) - a, vector_ids)], k)
