#!/usr/bin/env python3
"""Generates tests/golden/*.json.

Two kinds of content, kept apart:

1. reference_cases.json — the assertions the REFERENCE's own tests make on this path,
   transcribed by hand with file:line (the reference is Swift/Metal and cannot run here, so
   these are the only pins the reference itself provides: rank / membership / tolerance on toy
   corpora, byte-layout constants). When /root/reference is present the script also reads
   Tests/WaxIntegrationTests/Fixtures/minilm_baseline_embeddings.json and records a digest of
   it (the checkout's fixture is 8 x 384 of exactly 1.0 — a degenerate all-ties case, not real
   MiniLM output; SURVEY.md §4's "8 real vectors" does not hold for this checkout).

2. oracle_vectors.json — seeded synthetic cases with outputs computed by oracle/ (f64 truth).
   These are NOT reference outputs; they freeze the oracle so a later edit to it cannot
   silently move the goalposts, and give the GPU tests fixed expected ids/scores.

Run from the repo root:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402

REF = "/root/reference"

T = "Tests/WaxIntegrationTests/"

REFERENCE_CASES = {
    "_note": "Transcribed from the reference's tests; each 'expect' is an assertion the reference makes.",
    "engine_cases": [
        {
            "name": "vectorEngineAddSearchRemoveRoundtrip",
            "source": T + "VectorSearchEngineTests.swift:7-19",
            "metric": "cosine", "dimensions": 4,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0, 0.0, 0.0]},
                {"op": "add", "frameId": 1, "vector": [0.0, 1.0, 0.0, 0.0]},
                {"op": "search", "vector": [1.0, 0.0, 0.0, 0.0], "topK": 10,
                 "expect": {"nonEmpty": True, "contains": [0]}},
                {"op": "remove", "frameId": 0},
                {"op": "search", "vector": [1.0, 0.0, 0.0, 0.0], "topK": 10,
                 "expect": {"notContains": [0]}},
            ],
        },
        {
            "name": "vectorEngineSerializeDeserializeRoundtripPreservesSearch",
            "source": T + "VectorSearchEngineTests.swift:21-34",
            "metric": "cosine", "dimensions": 4,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0, 0.0, 0.0]},
                {"op": "add", "frameId": 1, "vector": [0.0, 1.0, 0.0, 0.0]},
                {"op": "serialize_deserialize_into_new_engine", "expect": {"blobNonEmpty": True}},
                {"op": "search", "vector": [0.0, 1.0, 0.0, 0.0], "topK": 10,
                 "expect": {"nonEmpty": True, "contains": [1]}},
            ],
        },
        {
            "name": "metalVectorEngineAddBatchUpdatesExistingIdsCorrectly",
            "source": T + "VectorSearchEngineTests.swift:36-47",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 10, "vector": [1.0, 0.0]},
                {"op": "add", "frameId": 20, "vector": [0.0, 1.0]},
                {"op": "addBatch", "frameIds": [20], "vectors": [[0.7, 0.7]]},
                {"op": "search", "vector": [0.7, 0.7], "topK": 1, "expect": {"first": 20}},
            ],
        },
        {
            "name": "vectorSearchSessionCosineSearchNormalizesScaledQueries",
            "source": T + "VectorSearchEngineTests.swift:101-131",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0]},
                {"op": "add", "frameId": 1, "vector": [0.0, 1.0]},
                {"op": "search", "vector": [1.0, 0.0], "topK": 2, "save": "unit", "expect": {"first": 0}},
                {"op": "search", "vector": [12.0, 0.0], "topK": 2, "save": "scaled",
                 "normalizeQueryLikeCaller": True, "expect": {"first": 0}},
                {"op": "compare_first_scores", "a": "unit", "b": "scaled", "tolerance": 0.001},
            ],
        },
        {
            "name": "mv2sVecIndexPersistsAndReopens (engine part)",
            "source": T + "VectorSearchEngineTests.swift:133-165",
            "metric": "cosine", "dimensions": 4,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0, 0.0, 0.0]},
                {"op": "add", "frameId": 1, "vector": [0.0, 1.0, 0.0, 0.0]},
                {"op": "serialize_deserialize_into_new_engine", "expect": {"blobNonEmpty": True}},
                {"op": "search", "vector": [0.9, 0.1, 0.0, 0.0], "topK": 10,
                 "expect": {"nonEmpty": True, "contains": [0]}},
            ],
        },
        {
            "name": "vectorOnlySearch (engine part)",
            "source": T + "UnifiedSearchTests.swift:64-83",
            "metric": "cosine", "dimensions": 4,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0, 0.0, 0.0]},
                {"op": "add", "frameId": 1, "vector": [0.0, 1.0, 0.0, 0.0]},
                {"op": "search", "vector": [0.9, 0.1, 0.0, 0.0], "topK": 10,
                 "normalizeQueryLikeCaller": True, "expect": {"first": 0}},
            ],
        },
        {
            "name": "vectorSearchWithoutManifestUsesPendingEmbeddings (engine part)",
            "source": T + "UnifiedSearchTests.swift:318-345",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [0.0, 1.0]},
                {"op": "search", "vector": [0.0, 1.0], "topK": 5, "expect": {"first": 0}},
            ],
        },
        {
            "name": "metalVectorSearchNormalizesNonNormalizedQueryEmbedding (engine part)",
            "source": T + "UnifiedSearchTests.swift:293-316",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0]},
                {"op": "search", "vector": [2.0, 0.0], "topK": 5, "normalizeQueryLikeCaller": True,
                 "expect": {"nonEmpty": True, "first": 0}},
            ],
        },
        {
            # vector lane of Wax.search with FrameFilter(frameIds:) — UnifiedSearch asks the engine for
            # candidateLimit = max(topK, min(3*topK, 1000)) (UnifiedSearch.swift:1195-1200) and post-filters
            # (passesFrameFilter, :1241-1258); a pre-filtering engine must return the same set here
            "name": "filtersAllowResultsBeyondTopK (engine part)",
            "source": T + "UnifiedSearchTests.swift:133-158",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0]},
                {"op": "add", "frameId": 1, "vector": [0.9, 0.1]},
                {"op": "add", "frameId": 2, "vector": [0.1, 0.9]},
                {"op": "add", "frameId": 3, "vector": [0.0, 1.0]},
                {"op": "search_filtered", "vector": [1.0, 0.0], "topK": 2, "allow": [2, 3],
                 "expect": {"idSetEquals": [2, 3]}},
            ],
        },
        {
            # engine part of the session test: an add followed by a remove must not survive the serialized index
            "name": "vectorSearchSessionAddThenRemoveBeforeCommitPersistsRemoval (engine part)",
            "source": T + "VectorSearchEngineTests.swift:78-99",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 0, "vector": [1.0, 0.0]},
                {"op": "remove", "frameId": 0},
                {"op": "serialize_deserialize_into_new_engine", "expect": {}},
                {"op": "search", "vector": [1.0, 0.0], "topK": 10, "expect": {"notContains": [0]}},
            ],
        },
        {
            "name": "metalSearchReusesTransientBuffers",
            "source": T + "MetalVectorEnginePoolTests.swift:6-20",
            "metric": "cosine", "dimensions": 2,
            "ops": [
                {"op": "add", "frameId": 1, "vector": [1.0, 0.0]},
                {"op": "search", "vector": [1.0, 0.0], "topK": 1, "expect": {"first": 1}},
                {"op": "pool_stats", "save": "first"},
                {"op": "search", "vector": [1.0, 0.0], "topK": 1, "expect": {"first": 1}},
                {"op": "pool_stats", "save": "second",
                 "expect": {"transientAllocationsEqual": "first", "reuseCountAtLeast": "first"}},
            ],
        },
    ],
    "vector_math": [
        {"source": T + "VectorSearchEngineTests.swift:73-76", "vector": [1.0, 0.0, 0.0], "isNormalizedL2": True},
        {"source": T + "VectorSearchEngineTests.swift:73-76", "vector": [2.0, 0.0, 0.0], "isNormalizedL2": False},
    ],
    "constants": {
        "mv2v_magic_hex": "4d563256",  # MetalVectorEngine.swift:686; VectorSerializer.swift:177
        "mv2v_header_size": 36,        # VectorSerializer.swift:176; MetalVectorEngine.swift:718
        "mv2v_version": 1,             # MetalVectorEngine.swift:687
        "mv2v_encoding_flat": 2,       # MetalVectorEngine.swift:689; VectorSerializer.swift:27
        "mv2v_encoding_usearch": 1,    # VectorSerializer.swift:26
        "similarity_raw": {"cosine": 0, "dot": 1, "l2": 2},  # WaxCore/FileFormat/MV2SEnums.swift:34-38
        "max_results": 10000,          # MetalVectorEngine.swift:18
        "initial_reserve": 64,         # MetalVectorEngine.swift:19
        "gpu_topk_threshold": 1000,    # MetalVectorEngine.swift:21
        "simd8_dimension_threshold": 384,  # MetalVectorEngine.swift:24
        "max_embedding_dimensions": 1000000,  # WaxCore/Constants.swift:51
        "normalized_tolerance": 0.001,  # VectorMath.swift:131
        "put_embedding_wal": {          # Tests/WaxCoreTests/WALEmbeddingCodecTests.swift:12-33
            "frameId": 1, "dimension": 2, "vector": [1.0, -2.0],
            "encoded_hex": "04" "0100000000000000" "02000000" "0000803f" "000000c0",
        },
    },
    "dimension_mismatch": {
        "source": "Sources/WaxVectorSearch/MetalVectorEngine.swift:830-833",
        "dimensions": 4, "query_len": 3,
        "message": "vector dimension mismatch: expected 4, got 3",
    },
}


def fixture_digest():
    path = os.path.join(REF, T, "Fixtures", "minilm_baseline_embeddings.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    e = np.array(d["embeddings"], dtype=np.float32)
    return {
        "source": T + "Fixtures/minilm_baseline_embeddings.json",
        "dimensions": int(d["dimensions"]),
        "count": int(e.shape[0]),
        "all_values_equal_to": float(e.flat[0]) if np.all(e == e.flat[0]) else None,
        "sha256_f32_le": hashlib.sha256(e.astype("<f4").tobytes()).hexdigest(),
        "note": "every component is exactly 1.0 in this checkout: all rows identical => every pairwise "
                "cosine is 1 and every search is an 8-way exact tie (resolved by ascending row).",
    }


def oracle_vectors():
    cases = []
    specs = [
        # name, metric, n, d, k, generator
        ("gauss_2000x384_k10_cos", oracle.METRIC_COSINE, 2000, 384, 10, "gauss"),
        ("gauss_2000x384_k30_cos", oracle.METRIC_COSINE, 2000, 384, 30, "gauss"),
        ("gauss_1500x768_k10_cos", oracle.METRIC_COSINE, 1500, 768, 10, "gauss"),
        ("gauss_3000x128_k24_cos", oracle.METRIC_COSINE, 3000, 128, 24, "gauss"),
        ("gauss_2000x384_k10_dot", oracle.METRIC_DOT, 2000, 384, 10, "gauss"),
        ("gauss_2000x384_k10_l2", oracle.METRIC_L2, 2000, 384, 10, "gauss"),
        ("lcg_1000x384_k10_cos", oracle.METRIC_COSINE, 1000, 384, 10, "lcg"),
        ("ties_1000x128_k24_cos", oracle.METRIC_COSINE, 1000, 128, 24, "ties"),
        ("gauss_777x100_k10_cos_generic_dims", oracle.METRIC_COSINE, 777, 100, 10, "gauss"),
        ("gauss_500x6_k5_cos_scalar_dims", oracle.METRIC_COSINE, 500, 6, 5, "gauss"),
    ]
    for name, metric, n, d, k, gen in specs:
        if gen == "gauss":
            corpus = oracle.gaussian_unit_rows(0, n, d)
            q = oracle.gaussian_unit_queries(3, d)
        elif gen == "lcg":
            corpus = np.stack([oracle.deterministic_embed(f"doc-{i}", d) for i in range(n)])
            q = np.stack([oracle.deterministic_embed(f"query-{i}", d) for i in range(3)])
        else:
            corpus = oracle.tie_pattern(0, n, d)
            q = oracle.gaussian_unit_queries(3, d)
            q = np.abs(q)
        per_query = []
        for qi in range(q.shape[0]):
            ids, scores, dists, rows = oracle.search(metric, corpus, None, q[qi], k)
            per_query.append({"ids": [int(x) for x in ids], "scores": [float(np.float32(s)) for s in scores]})
        cases.append({
            "name": name, "metric": int(metric), "n": n, "d": d, "k": k, "generator": gen,
            "corpus_sha256": hashlib.sha256(corpus.astype("<f4").tobytes()).hexdigest(),
            "queries_sha256": hashlib.sha256(q.astype("<f4").tobytes()).hexdigest(),
            "results": per_query,
        })
    return {
        "_note": "Outputs of oracle/ (f64-accumulated truth, (distance asc,row asc) order) on seeded synthetic "
                 "inputs. NOT reference outputs. Regenerate with tests/golden/make_golden.py.",
        "seeds": {"corpus": oracle.CORPUS_SEED, "query": oracle.QUERY_SEED, "granule": oracle.SHARD_ROWS},
        "cases": cases,
    }


def main():
    ref = dict(REFERENCE_CASES)
    dig = fixture_digest()
    ref_path = os.path.join(HERE, "reference_cases.json")
    if dig is None and os.path.exists(ref_path):
        dig = json.load(open(ref_path)).get("minilm_fixture")
    ref["minilm_fixture"] = dig
    json.dump(ref, open(ref_path, "w"), indent=1, sort_keys=False)
    json.dump(oracle_vectors(), open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=1)
    print("wrote", ref_path, "and oracle_vectors.json")


if __name__ == "__main__":
    main()
