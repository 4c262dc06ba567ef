"""VectorSerializer — "MV2V" vec-segment header codec (reference:
Sources/WaxVectorSearch/VectorSerializer.swift:5-252). Host-side byte parsing only: lets callers
tell a flat (encoding 2) segment — which HIPVectorEngine ingests directly — from a USearch
(encoding 1) one, exactly like `detectEncoding` / `decodeVecSegment`.
"""
from __future__ import annotations

import enum
import struct
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .errors import InvalidToc

MAGIC = b"MV2V"          # VecSegmentHeaderV1.magic (:177)
HEADER_SIZE = 36         # VecSegmentHeaderV1.encodedSize (:176)


class VecEncoding(enum.IntEnum):  # :25-28
    uSearch = 1
    metal = 2


@dataclass(frozen=True)
class SegmentInfo:  # :6-18
    similarity: int
    dimension: int
    vectorCount: int
    payloadLength: int


def detectEncoding(data: bytes) -> VecEncoding:  # noqa: N802 — :32-52
    if len(data) < 8:
        raise InvalidToc(f"vec segment too small: {len(data)} bytes")
    if data[:4] != MAGIC:
        raise InvalidToc("vec segment magic mismatch")
    (version,) = struct.unpack_from("<H", data, 4)
    if version != 1:
        raise InvalidToc(f"unsupported vec segment version {version}")
    try:
        return VecEncoding(data[6])
    except ValueError:
        raise InvalidToc(f"unsupported vec segment encoding {data[6]}")


def decodeHeader(data: bytes) -> Tuple[SegmentInfo, int]:  # noqa: N802 — decodeAnyEncoding (:214-250)
    if len(data) < HEADER_SIZE:
        raise InvalidToc(f"vec segment too small: {len(data)} bytes")
    if data[:4] != MAGIC:
        raise InvalidToc("vec segment magic mismatch")
    version, encoding, similarity, dimension, count, payload = struct.unpack_from("<HBBIQQ", data, 4)
    if version != 1:
        raise InvalidToc(f"unsupported vec segment version {version}")
    if encoding not in (1, 2):
        raise InvalidToc(f"unsupported vec segment encoding {encoding}")
    if similarity > 2:
        raise InvalidToc(f"vec similarity must be 0..2 (got {similarity})")
    if data[28:36] != b"\x00" * 8:
        raise InvalidToc("vec segment reserved bytes must be zero")
    return SegmentInfo(similarity, dimension, count, payload), encoding


def decodeVecSegment(data: bytes):  # noqa: N802 — :84-157
    """Returns ("uSearch", info, payload_bytes) or ("metal", info, vectors[n,d] f32, frameIds[n] u64)."""
    info, encoding = decodeHeader(data)
    if encoding == VecEncoding.uSearch:
        expected = HEADER_SIZE + info.payloadLength
        if len(data) != expected:
            raise InvalidToc(f"vec segment length mismatch: expected {expected}, got {len(data)}")
        return ("uSearch", info, data[HEADER_SIZE:])
    vec_len = info.payloadLength
    if vec_len != info.vectorCount * info.dimension * 4:
        raise InvalidToc("vec vector data length mismatch")
    off = HEADER_SIZE
    if len(data) < off + vec_len + 8:
        raise InvalidToc("vec segment missing frameIds length")
    (id_len,) = struct.unpack_from("<Q", data, off + vec_len)
    if id_len != info.vectorCount * 8:
        raise InvalidToc("vec frameId data length mismatch")
    expected = off + vec_len + 8 + id_len
    if len(data) != expected:
        raise InvalidToc(f"vec segment length mismatch: expected {expected}, got {len(data)}")
    vectors = np.frombuffer(data, dtype="<f4", count=info.vectorCount * info.dimension, offset=off)
    vectors = vectors.reshape(info.vectorCount, info.dimension).copy()
    ids = np.frombuffer(data, dtype="<u8", count=info.vectorCount, offset=off + vec_len + 8).copy()
    return ("metal", info, vectors, ids)
