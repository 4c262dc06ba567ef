#!/usr/bin/env python3
"""Generates tests/golden/metal_shader_vectors.json: inputs and outputs of the reference's OWN Metal compute shaders
(CosineDistance.metal: cosineDistanceKernelSIMD4 / SIMD8; TopKReduction.metal: topKReduceDistances / topKReduceEntries under the
dispatch loop of MetalVectorEngine.swift:517-585), executed on the CPU by oracle/_ref (oracle/ref_metal/: the shader files are
compiled as C++ from where they lie under /root/reference — nothing is copied — under IEEE binary32 semantics).

Run in the container that has the reference checkout:   python oracle/gen_metal_golden.py
The fixture travels (the GPU box has neither /root/reference nor a need for oracle/_ref): tests pin the oracle's restatement —
and the HIP engine — against these outputs of the reference itself. TEST INFRASTRUCTURE ONLY."""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def b64(a):
    return base64.b64encode(np.ascontiguousarray(a).tobytes()).decode("ascii")


def main():
    if oracle.ref_lib() is None:
        raise SystemExit("oracle/_ref is not available: this script runs where /root/reference exists")
    rng = np.random.default_rng(20260924)
    out = {"generator": "oracle/gen_metal_golden.py",
           "source": "CosineDistance.metal:152-328 and TopKReduction.metal:54-167 of christopherkarani/Wax, compiled as C++ against "
                     "oracle/ref_metal/msl/metal_stdlib and run by oracle/ref_metal/ref_metal.cpp (threadgroups of 256 fibers)",
           "encoding": "arrays are base64 of little-endian f32 / u32 bytes; distance-case INPUTS are not stored: "
                       "oracle.formula_rows(seed, rows, dims) with row 5 zeroed and row 6 scaled by 3.5, oracle.formula_unit_query(seed, dims) "
                       "(scaled by 0.37 when unit_query is false)",
           "distance_cases": [], "topk_cases": []}
    # (dims, rows): both kernels, float4 remainders (dims % 4, odd float4 count), a zero row, a non-unit row, a non-unit query.
    # rows = 200: the cooperative query load of the kernels is done by the LIVE threads of a threadgroup only (the bounds check
    # returns before it), so a last threadgroup with fewer rows than D / 4 computes with a partly unloaded query — see
    # test_reference_shaders.py; 200 >= 768 / 4 keeps the fixture on the defined side.
    for seed, (dims, n, unit_query) in enumerate([(4, 200, True), (10, 200, True), (100, 200, True), (384, 200, True), (387, 200, True),
                                                  (768, 200, True), (384, 200, False), (392, 200, True)]):
        x = oracle.formula_rows(seed, n, dims)
        x[5] = 0.0                                  # sqrt(m) <= 1e-6 => similarity 0 (CosineDistance.metal:225 / :323)
        x[6] *= np.float32(3.5)                     # rows need not be unit-norm
        q = oracle.formula_unit_query(seed, dims)
        if not unit_query:
            q = (q * np.float32(0.37)).astype(np.float32)
        d = oracle.ref_metal_distances(x, q)
        out["distance_cases"].append({"seed": seed, "dims": dims, "rows": n, "kernel": "cosineDistanceKernelSIMD8" if dims >= 384 else "cosineDistanceKernelSIMD4",
                                      "unit_query": unit_query, "distances": b64(d)})
    # top-k: distinct distances, heavy ties, and the padding of the last threadgroup
    for name, n, ks in [("distinct", 1500, [1, 10, 30, 64, 65, 100, 128]), ("ties", 1100, [10, 30, 64]), ("one_group_plus", 1001, [10, 24])]:
        if name == "ties":
            d = (rng.integers(0, 40, size=n) / 64.0).astype(np.float32)
        else:
            d = rng.permutation(n).astype(np.float32) / np.float32(n)
        case = {"name": name, "rows": n, "distances": b64(d), "results": []}
        for k in ks:
            idx, dist, passes = oracle.ref_metal_topk(d, k)
            case["results"].append({"k": k, "passes": passes, "indices": b64(idx), "distances": b64(dist)})
        out["topk_cases"].append(case)
    # end to end as MetalVectorEngine.search composes it (:494-611): distance kernel -> GPU top-k -> (index, distance) pairs.
    # Store sizes are multiples of 256 rows (every threadgroup full); k = 30 is the callers' candidateLimit for topK 10.
    out["search_cases"] = []
    for seed, (dims, n, k) in enumerate([(384, 2048, 30), (768, 1024, 30), (128, 1280, 10), (384, 2560, 128)], start=40):
        x = oracle.formula_rows(seed, n, dims)
        q = oracle.formula_unit_query(seed, dims)
        d = oracle.ref_metal_distances(x, q)
        idx, dist, passes = oracle.ref_metal_topk(d, k)
        out["search_cases"].append({"seed": seed, "dims": dims, "rows": n, "k": k, "passes": passes, "indices": b64(idx), "distances": b64(dist)})
    # the host loop's fixed points (MetalVectorEngine.swift:548): no progress for k > 128
    non_term = []
    for n, k in [(1500, 129), (1500, 200), (1500, 256), (5000, 130)]:
        try:
            oracle.ref_metal_topk(np.arange(n, dtype=np.float32), k)
            non_term.append({"rows": n, "k": k, "terminates": True})
        except oracle.NonTermination:
            non_term.append({"rows": n, "k": k, "terminates": False})
    out["host_loop_fixed_points"] = non_term
    path = os.path.join(ROOT, "tests", "golden", "metal_shader_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
