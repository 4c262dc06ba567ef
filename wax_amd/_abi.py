"""ctypes binding of the C ABI declared in include/wax_hip.h.

This is plumbing: the product is libwaxhip.so. The library is loaded from
wax_amd/lib/ (in-tree build); a missing library is a hard error — there is no
Python or CPU fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List

from . import build as _build

HEADER_PATH = os.path.join(_build.ROOT, "include", "wax_hip.h")

# status codes (wax_hip_status)
OK = 0
ERR_DIM_MISMATCH = -1
ERR_CAPACITY = -2
ERR_NO_DEVICE = -3
ERR_ALLOC = -4
ERR_BAD_SEGMENT = -5
ERR_METRIC_UNSUPPORTED = -6
ERR_INVALID_ARGUMENT = -7
ERR_INTERNAL = -8

MAX_RESULTS = 10000
KEY_PAD = (1 << 63) - 1
ID_PAD = (1 << 64) - 1


class Hit(ctypes.Structure):
    """wax_hip_hit"""
    _fields_ = [("key", ctypes.c_int64), ("frame_id", ctypes.c_uint64)]


class Stats(ctypes.Structure):
    """wax_hip_stats_t"""
    _fields_ = [
        ("searches", ctypes.c_uint64),
        ("rows_scanned", ctypes.c_uint64),
        ("bytes_scanned", ctypes.c_uint64),
        ("transient_allocations", ctypes.c_uint64),
        ("reuse_count", ctypes.c_uint64),
        ("reserved_rows", ctypes.c_uint64),
        ("last_scan_kernel_ms", ctypes.c_double),
        ("scan_kernel_ms_total", ctypes.c_double),
        ("scan_kernels_timed", ctypes.c_uint64),
        ("batch_gemm_ms_total", ctypes.c_double),
        ("batch_gemms_timed", ctypes.c_uint64),
        ("batch_gemm_rows", ctypes.c_uint64),
        ("batch_gemm_queries", ctypes.c_uint64),
    ]


class RrfLane(ctypes.Structure):
    """wax_hip_rrf_lane (include/wax_hip.h)."""
    _fields_ = [("d_ids", ctypes.c_void_p), ("d_counts", ctypes.c_void_p), ("stride", ctypes.c_uint32),
                ("pitch", ctypes.c_uint32), ("weight", ctypes.c_float)]


_engine_p = ctypes.c_void_p
_f32p = ctypes.POINTER(ctypes.c_float)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_hitp = ctypes.POINTER(Hit)

# name -> (restype, argtypes); must cover every function include/wax_hip.h declares
SIGNATURES: Dict[str, tuple] = {
    "wax_hip_rrf_fuse_batch_device": (ctypes.c_int, [ctypes.POINTER(RrfLane), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]),
    "wax_hip_rrf_fuse": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_int32, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float),
                                        ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32,
                                        ctypes.POINTER(ctypes.c_uint32)]),
    "wax_hip_available": (ctypes.c_int, []),
    "wax_hip_device_count": (ctypes.c_int, []),
    "wax_hip_abi_version": (ctypes.c_uint32, []),
    "wax_hip_last_error": (ctypes.c_char_p, []),
    "wax_hip_engine_create": (ctypes.c_int, [ctypes.c_uint8, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(_engine_p)]),
    "wax_hip_engine_create_sharded": (ctypes.c_int, [ctypes.c_uint8, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                     ctypes.POINTER(_engine_p)]),
    "wax_hip_shard_count": (ctypes.c_int, [_engine_p]),
    "wax_hip_shard_info": (ctypes.c_int, [_engine_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _u64p, _u64p]),
    "wax_hip_engine_destroy": (None, [_engine_p]),
    "wax_hip_dimensions": (ctypes.c_uint32, [_engine_p]),
    "wax_hip_count": (ctypes.c_uint64, [_engine_p]),
    "wax_hip_metric_of": (ctypes.c_uint8, [_engine_p]),
    "wax_hip_device_of": (ctypes.c_int, [_engine_p]),
    "wax_hip_add": (ctypes.c_int, [_engine_p, ctypes.c_uint64, _f32p, ctypes.c_uint32]),
    "wax_hip_add_batch": (ctypes.c_int, [_engine_p, _u64p, _f32p, ctypes.c_uint64, ctypes.c_uint32]),
    "wax_hip_apply_put_embeddings": (ctypes.c_int, [_engine_p, ctypes.c_char_p, ctypes.c_uint64, _u64p]),
    "wax_hip_search_filtered": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int, _u64p,
                                               ctypes.c_uint64, ctypes.c_int, ctypes.c_float, _u64p, _f32p,
                                               ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]),
    "wax_hip_merge_batch_hits_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]),
    "wax_hip_add_batch_device": (ctypes.c_int, [_engine_p, _u64p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32]),
    "wax_hip_remove": (ctypes.c_int, [_engine_p, ctypes.c_uint64]),
    "wax_hip_reserve": (ctypes.c_int, [_engine_p, ctypes.c_uint64]),
    "wax_hip_result_capacity": (ctypes.c_uint32, [ctypes.c_int32]),
    "wax_hip_search": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_int32, _u64p, _f32p, ctypes.c_uint32, _u32p]),
    "wax_hip_search_submit": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_int32, _u64p]),
    "wax_hip_search_collect": (ctypes.c_int, [_engine_p, ctypes.c_uint64, _u64p, _f32p, ctypes.c_uint32, _u32p]),
    "wax_hip_search_batch": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, _u64p, _f32p, ctypes.c_uint32, _u32p]),
    "wax_hip_search_batch_hits": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, _hitp, ctypes.c_uint32, _u32p]),
    "wax_hip_search_batch_hits_device": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                                        ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "wax_hip_search_batch_submit_device": (ctypes.c_int, [_engine_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                                          ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, _u64p]),
    "wax_hip_search_batch_collect_device": (ctypes.c_int, [_engine_p, ctypes.c_uint64, _u32p]),
    "wax_hip_set_row_base": (ctypes.c_int, [_engine_p, ctypes.c_uint64]),
    "wax_hip_search_shard_device": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "wax_hip_merge_hits_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]),
    "wax_hip_hits_to_results": (ctypes.c_int, [ctypes.c_uint8, _hitp, ctypes.c_uint32, _u64p, _f32p, _u32p]),
    "wax_hip_serialize": (ctypes.c_int, [_engine_p, ctypes.POINTER(_u8p), ctypes.POINTER(ctypes.c_size_t)]),
    "wax_hip_deserialize": (ctypes.c_int, [_engine_p, ctypes.c_char_p, ctypes.c_size_t]),
    "wax_hip_free": (None, [ctypes.c_void_p]),
    "wax_hip_stats": (ctypes.c_int, [_engine_p, ctypes.POINTER(Stats)]),
    "wax_hip_set_tuning": (ctypes.c_int, [_engine_p, ctypes.c_char_p, ctypes.c_int64]),
    "wax_hip_get_tuning": (ctypes.c_int64, [_engine_p, ctypes.c_char_p]),
    "wax_hip_time_scan_kernel": (ctypes.c_int, [_engine_p, _f32p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]),
    "wax_hip_time_stream_read": (ctypes.c_int, [_engine_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]),
}


def declared_symbols(header_path: str = HEADER_PATH) -> List[str]:
    """Every function name the public header declares (used by the CPU tests)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wax_hip_[a-z_0-9]+)\s*\(", text)))


class LibraryMissing(RuntimeError):
    pass


_lib = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64. Two HIP runtimes
    in one process fight over the device (whichever initialises second sees "no ROCm-capable
    device"), so when torch is installed it must load ITS runtime first; libwaxhip.so (NEEDED
    libamdhip64.so.7) then binds to the copy already in the process. A host without torch
    (the Swift / C++ case) just uses the system ROCm runtime."""
    if os.environ.get("WAX_HIP_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
        if getattr(torch.version, "hip", None):
            torch.cuda.is_available()  # dlopens torch/lib/libamdhip64.so and initialises it
    except Exception:  # noqa: BLE001 — torch absent or broken: system runtime only
        pass


def lib() -> ctypes.CDLL:
    """Load libwaxhip.so and bind every declared symbol. Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    path = os.environ.get("WAX_HIP_LIB") or _build.LIB_PATH       # (WAX_HIP_LIB: an experiment build of the same ABI, A/B sessions only)
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} not found: build it with `python -m wax_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    L = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError => the ABI and the header disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


def last_error() -> str:
    msg = lib().wax_hip_last_error()
    return msg.decode("utf-8", "replace") if msg else ""
