"""WaxError cases thrown on the vector-search path (reference: Sources/WaxCore/WaxError.swift:4-18)."""
from __future__ import annotations

from . import _abi


class WaxError(Exception):
    """Base of the reference's `WaxError` enum."""


class EncodingError(WaxError):
    """WaxError.encodingError(reason:) — e.g. "vector dimension mismatch: expected X, got Y"
    (MetalVectorEngine.swift:830-833, 361-363)."""


class CapacityExceeded(WaxError):
    """WaxError.capacityExceeded(limit:requested:) (MetalVectorEngine.swift:157-162, 858-860)."""


class InvalidToc(WaxError):
    """WaxError.invalidToc(reason:) — everything infrastructural: no device, allocation failure,
    malformed vec segment (MetalVectorEngine.swift:154-169, 718-808)."""


def raise_for_status(code: int) -> None:
    """Map a wax_hip_status to the WaxError the Swift shim would throw."""
    if code == _abi.OK:
        return
    msg = _abi.last_error()
    if code in (_abi.ERR_DIM_MISMATCH, _abi.ERR_INVALID_ARGUMENT):
        raise EncodingError(msg)
    if code == _abi.ERR_CAPACITY:
        raise CapacityExceeded(msg)
    raise InvalidToc(msg)
