"""Caller-side query normalisation (reference: Sources/Wax/Utilities/VectorMath.swift:14-33, 120-135).

Host logic only: UnifiedSearch normalises a query before handing it to a GPU engine when
|‖q‖ - 1| > 1e-3 (UnifiedSearch.swift:142-146; VectorSearchSession.swift:70-76).
"""
from __future__ import annotations

import numpy as np


def magnitude(vector) -> float:
    v = np.asarray(vector, dtype=np.float32)
    if v.size == 0:
        return 0.0
    return float(np.sqrt(np.sum(v * v, dtype=np.float32)))


def normalizeL2(vector) -> np.ndarray:  # noqa: N802
    v = np.asarray(vector, dtype=np.float32)
    if v.size == 0:
        return v
    mag = np.float32(magnitude(v))
    if not mag > 0:
        return v
    return (v * (np.float32(1.0) / mag)).astype(np.float32)


def isNormalizedL2(vector, tolerance: float = 1e-3) -> bool:  # noqa: N802
    v = np.asarray(vector, dtype=np.float32)
    if v.size == 0:
        return False
    return abs(magnitude(v) - 1.0) <= tolerance
