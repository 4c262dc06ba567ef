"""VectorMetric (reference: Sources/WaxVectorSearch/VectorMetric.swift:5-55)."""
from __future__ import annotations

import enum
import math


class VectorMetric(enum.IntEnum):
    """Raw values are VecSimilarity's on-disk values (WaxCore/FileFormat/MV2SEnums.swift:34-38)."""
    cosine = 0
    dot = 1
    l2 = 2

    def score(self, fromDistance: float) -> float:  # noqa: N803 — mirrors score(fromDistance:)
        """VectorMetric.swift:32-43: non-finite -> 0; cosine -> 1 - d; dot/l2 -> -d."""
        d = float(fromDistance)
        if not math.isfinite(d):
            return 0.0
        if self is VectorMetric.cosine:
            return 1.0 - d
        return -d

    def toVecSimilarity(self) -> int:  # noqa: N802
        return int(self)


class VectorEnginePreference(enum.Enum):
    """VectorSearchEngine.swift:4-8, plus the case a HIP-enabled build adds."""
    auto = "auto"
    metalPreferred = "metalPreferred"
    cpuOnly = "cpuOnly"
    hipPreferred = "hipPreferred"
