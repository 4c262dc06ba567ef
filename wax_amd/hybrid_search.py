"""HybridSearch (Sources/Wax/UnifiedSearch/HybridSearch.swift): reciprocal-rank fusion of ranked frame-id lists, on the
device (wax_hip_rrf_fuse / wax_hip_rrf_fuse_batch_device, wax_amd/csrc/rrf.hip). Same names and argument meaning as the
Swift enum; results are [(frameId, score)] sorted by (score desc, bestRank asc, frameId asc) — bit-identical scores."""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np

from . import _abi
from .errors import raise_for_status

RRF_MAX_LANES = 8
RRF_MAX_ENTRIES = 4096


def rrfFusionArrays(lists: Sequence[Tuple[float, Sequence[int]]], k: int = 60, device: int = -1):  # noqa: N802
    """lists = [(weight, frameIds), ...] -> (ids u64[m], scores f32[m], bestRank u32[m], sources u32[m])."""
    lib = _abi.lib()
    n = len(lists)
    weights = (ctypes.c_float * max(n, 1))(*[float(np.float32(w)) for w, _ in lists])
    arrays = [np.ascontiguousarray(np.asarray(ids, dtype=np.uint64).reshape(-1)) for _, ids in lists]
    ptrs = (ctypes.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrays])
    counts = (ctypes.c_uint32 * max(n, 1))(*[a.size for a in arrays])
    total = int(sum(a.size for a in arrays))
    cap = max(total, 1)
    out_ids = np.empty(cap, dtype=np.uint64)
    out_scores = np.empty(cap, dtype=np.float32)
    out_rank = np.empty(cap, dtype=np.uint32)
    out_src = np.empty(cap, dtype=np.uint32)
    got = ctypes.c_uint32(0)
    rc = lib.wax_hip_rrf_fuse(weights, ptrs, counts, n, int(k), int(device),
                              out_ids.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                              out_scores.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                              out_rank.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                              out_src.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cap, ctypes.byref(got))
    raise_for_status(rc)
    m = got.value
    return out_ids[:m].copy(), out_scores[:m].copy(), out_rank[:m].copy(), out_src[:m].copy()


def rrfFusion(lists=None, k: int = 60, textResults=None, vectorResults=None, alpha: float = 0.5) -> List[Tuple[int, float]]:  # noqa: N802,N803
    """rrfFusion(lists:k:) (HybridSearch.swift:25-52), or rrfFusion(textResults:vectorResults:k:alpha:) (:8-23) when
    `lists` is None: alpha clamped to [0, 1]; the score halves of the (frameId, score) inputs are ignored (rank-based)."""
    if lists is None:
        a = np.float32(min(1.0, max(0.0, alpha)))
        lists = [(a, [int(t[0]) for t in (textResults or [])]), (np.float32(1.0) - a, [int(t[0]) for t in (vectorResults or [])])]
    ids, scores, _, _ = rrfFusionArrays(lists, k)
    return [(int(i), float(s)) for i, s in zip(ids, scores)]


def rrfFusionBatchDevice(lanes, nq: int, k: int, d_out_ids: int, d_out_scores: int, out_stride: int,  # noqa: N802
                         d_out_best_rank: int = 0, d_out_sources: int = 0, d_out_counts: int = 0, stream: int = 0) -> None:
    """wax_hip_rrf_fuse_batch_device. lanes = [(d_ids_ptr, d_counts_ptr or 0, stride, pitch, weight), ...] — raw device
    pointers (e.g. torch tensors' data_ptr()); a wax_hip_hit array is a lane with pitch 2 and d_ids_ptr = hits_ptr + 8."""
    lib = _abi.lib()
    arr = (_abi.RrfLane * max(len(lanes), 1))()
    for i, (ids, counts, stride, pitch, weight) in enumerate(lanes):
        arr[i].d_ids = ids
        arr[i].d_counts = counts or None
        arr[i].stride = int(stride)
        arr[i].pitch = int(pitch)
        arr[i].weight = float(np.float32(weight))
    rc = lib.wax_hip_rrf_fuse_batch_device(arr, len(lanes), int(nq), int(k), ctypes.c_void_p(d_out_ids), ctypes.c_void_p(d_out_scores),
                                           ctypes.c_void_p(d_out_best_rank or None), ctypes.c_void_p(d_out_sources or None),
                                           int(out_stride), ctypes.c_void_p(d_out_counts or None), ctypes.c_void_p(stream or None))
    raise_for_status(rc)
