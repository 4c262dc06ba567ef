"""wax_amd — MI355X (gfx950) backend for Wax's brute-force vector scan + top-k path.

The product is `wax_amd/lib/libwaxhip.so` (hand-written HIP kernels behind the C ABI in
`include/wax_hip.h`). This package is the host-side mirror of the reference's
`VectorSearchEngine` interface used by the tests and by `bench.py`.
"""
from .errors import CapacityExceeded, EncodingError, InvalidToc, WaxError
from .vector_metric import VectorEnginePreference, VectorMetric
from .engine import HIPVectorEngine, BufferPoolStats, clampTopK
from . import vector_math as VectorMath
from . import vector_serializer as VectorSerializer
from . import hybrid_search as HybridSearch

__all__ = [
    "HIPVectorEngine", "BufferPoolStats", "clampTopK", "VectorMetric", "VectorEnginePreference", "VectorMath",
    "VectorSerializer", "HybridSearch", "WaxError", "EncodingError", "CapacityExceeded", "InvalidToc",
]
