"""The oracle — and, on a GPU, the HIP engine — against outputs of the REFERENCE ITSELF: the Metal compute shaders of
christopherkarani/Wax (CosineDistance.metal, TopKReduction.metal) executed on the CPU by oracle/_ref (oracle/ref_metal/: the shader
files compile unchanged as C++ against a stand-in for <metal_stdlib>; threads of a threadgroup are fibers).

tests/golden/metal_shader_vectors.json (made by oracle/gen_metal_golden.py where the reference checkout exists) carries inputs and
outputs, so these checks run everywhere; the live comparisons run only where oracle/_ref can be built (they skip on the GPU box)."""
import base64
import json
import os

import numpy as np
import pytest

import oracle
from helpers import GOLDEN, assert_parity


def _arr(b, dtype):
    return np.frombuffer(base64.b64decode(b), dtype=dtype).copy()


def _inputs(c):
    """The inputs of a distance case (the fixture stores outputs only): see the fixture's "encoding" note."""
    x = oracle.formula_rows(c["seed"], c["rows"], c["dims"])
    x[5] = 0.0
    x[6] *= np.float32(3.5)
    q = oracle.formula_unit_query(c["seed"], c["dims"])
    if not c["unit_query"]:
        q = (q * np.float32(0.37)).astype(np.float32)
    return x, q


@pytest.fixture(scope="module")
def wax(hip_lib):
    import wax_amd
    if hip_lib.wax_hip_device_count() == 0:
        pytest.skip("no HIP device on this host: the gpu-marked tests run on the MI355X box (pytest -m gpu)")
    assert hip_lib.wax_hip_available() == 1, "a HIP device is visible but it is not gfx950: the HIP path needs an MI355X"
    return wax_amd


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "metal_shader_vectors.json")))


def test_oracle_metal_mode_is_bit_identical_to_the_reference_shaders(golden):
    """wax_oracle_cosine_distances_metal restates cosineDistanceKernelSIMD4 / SIMD8 line by line; the reference's own source,
    run under IEEE semantics, gives the same 32 bits for every row: float4 remainders, D % 4 tails, zero rows, non-unit rows
    and a non-unit query (the kernels do not divide by |q|)."""
    assert len(golden["distance_cases"]) >= 8
    for c in golden["distance_cases"]:
        x, q = _inputs(c)
        want = _arr(c["distances"], np.float32)
        got = oracle.distances(oracle.METRIC_COSINE, x, q, mode=oracle.MODE_METAL_F32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (c["dims"], c["kernel"])
        assert want[5] == 1.0                                             # zero row: similarity 0


def test_f64_truth_is_within_tolerance_of_the_reference_shaders(golden):
    """The parity truth (f64 accumulation, full cosine) against the reference's f32 kernels on unit queries: far inside the
    north star's 1e-5 — which is why one tolerance serves both the Metal and the USearch side of the reference."""
    worst = 0.0
    for c in golden["distance_cases"]:
        if not c["unit_query"]:
            continue
        x, q = _inputs(c)
        truth = oracle.distances(oracle.METRIC_COSINE, x, q, mode=oracle.MODE_TRUTH_F64)
        worst = max(worst, float(np.max(np.abs(truth - _arr(c["distances"], np.float32)))))
    assert worst <= 2e-6, worst


def test_oracle_selection_equals_the_reference_gpu_reduction(golden):
    """topKReduceDistances + topKReduceEntries under the reference's dispatch loop against the oracle's selection: identical
    index lists where distances are distinct; with ties the reference's bitonic / heap passes leave equal distances in an order
    of their own ("ties unordered"), so there the distance lists must match and every index must carry its distance."""
    for case in golden["topk_cases"]:
        d = _arr(case["distances"], np.float32)
        for r in case["results"]:
            k = r["k"]
            idx, dist = _arr(r["indices"], np.uint32), _arr(r["distances"], np.float32)
            o_idx, o_dist = oracle.topk_heap(d, k, total=True)            # (distance asc, index asc): the engine's order
            assert np.array_equal(dist, np.asarray(o_dist, dtype=np.float32)), (case["name"], k)
            assert np.array_equal(d[idx], dist) and len(set(idx.tolist())) == k
            if case["name"] != "ties":
                assert np.array_equal(idx, np.asarray(o_idx).astype(np.uint32)), (case["name"], k)
            assert 2 <= r["passes"] <= 6                                   # 2-6 launches


def test_oracle_search_equals_the_reference_pipeline_end_to_end(golden):
    """Distance kernel -> GPU top-k, as MetalVectorEngine.search composes them, against oracle.search in its Metal-faithful mode:
    the same rows in the same order with the same 32-bit distances (scores = 1 - d, VectorMetric.swift:32-43)."""
    assert len(golden["search_cases"]) >= 4
    for c in golden["search_cases"]:
        x, q = oracle.formula_rows(c["seed"], c["rows"], c["dims"]), oracle.formula_unit_query(c["seed"], c["dims"])
        idx, dist = _arr(c["indices"], np.uint32), _arr(c["distances"], np.float32)
        ids = np.arange(c["rows"], dtype=np.uint64) + 500
        o_ids, o_scores, _, _ = oracle.search(oracle.METRIC_COSINE, x, ids, q, c["k"], mode=oracle.MODE_METAL_F32)
        assert np.array_equal(np.asarray(o_ids, dtype=np.uint64), idx.astype(np.uint64) + 500), (c["dims"], c["rows"])
        assert np.array_equal(np.asarray(o_scores, dtype=np.float32), np.float32(1.0) - dist)


def test_reference_reduction_loop_has_fixed_points_above_k_128(golden):
    """A property of the reference recorded while pinning it: `while currentCount > topKCount` (MetalVectorEngine.swift:548)
    makes no progress once ceil(count / 256) * k == count, which happens for every k > 128. The HIP engine serves those k on the
    device (fused up to 192, radix select beyond); the reference's callers never ask (candidateLimit = 30)."""
    got = {(e["rows"], e["k"]): e["terminates"] for e in golden["host_loop_fixed_points"]}
    assert got == {(1500, 129): False, (1500, 200): False, (1500, 256): False, (5000, 130): False}


def test_live_reference_shaders_against_the_oracle_on_random_shapes():
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref is only built where the reference checkout exists")
    rng = np.random.default_rng(5)
    for dims in (1, 3, 4, 7, 8, 64, 383, 384, 385, 390, 391, 1024):
        n = 256 * int(rng.integers(0, 3)) + int(rng.integers(max(1, (dims + 3) // 4), 257))   # last threadgroup loads the whole query
        x = rng.standard_normal((n, dims)).astype(np.float32)
        q = rng.standard_normal(dims).astype(np.float32)
        for simd8 in (False, True):
            got = oracle.ref_metal_distances(x, q, simd8=simd8)
            # the oracle picks the kernel by the engine's rule (D >= 384); compare like with like
            if simd8 == (dims >= 384):
                want = oracle.distances(oracle.METRIC_COSINE, x, q, mode=oracle.MODE_METAL_F32)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dims, n, simd8)
    d = rng.standard_normal(4000).astype(np.float32)
    for k in (1, 2, 10, 63, 64, 65, 127, 128):
        idx, dist, passes = oracle.ref_metal_topk(d, k)
        o_idx, o_dist = oracle.topk_heap(d, k, total=True)
        assert np.array_equal(idx, np.asarray(o_idx).astype(np.uint32)) and np.array_equal(dist, np.asarray(o_dist, dtype=np.float32))
    with pytest.raises(oracle.NonTermination):
        oracle.ref_metal_topk(d, 200)


def test_golden_file_is_what_the_generator_produces_here(golden):
    """Where the reference checkout exists, regenerate one case and compare with the committed fixture (a stale fixture would
    otherwise go unnoticed)."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref is only built where the reference checkout exists")
    for c in golden["distance_cases"]:
        x, q = _inputs(c)
        assert np.array_equal(oracle.ref_metal_distances(x, q).view(np.uint32), _arr(c["distances"], np.float32).view(np.uint32)), c["dims"]


def test_reference_kernels_load_the_query_with_live_threads_only():
    """Recorded while pinning the reference: both SIMD kernels return on `vectorIndex >= vectorCount` BEFORE the cooperative
    load of the query into threadgroup memory (CosineDistance.metal:165-167 / :246-248 ahead of :176-184 / :259-266), so a last
    threadgroup with fewer rows than D / 4 leaves part of the query unloaded and its rows are scored against whatever the
    threadgroup memory held (oracle/_ref poisons it with NaN). 24 rows x 384 dims: 24 live threads load 24 of the 96 float4; a
    store of 256 + 24 rows gets 256 good rows and 24 undefined ones. The oracle (and the HIP engine) score every row against
    the whole query."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref is only built where the reference checkout exists")
    x, q = oracle.formula_rows(3, 280, 384), oracle.formula_unit_query(3, 384)
    shader = oracle.ref_metal_distances(x, q)
    full = oracle.distances(oracle.METRIC_COSINE, x, q, mode=oracle.MODE_METAL_F32)
    assert np.array_equal(shader[:256].view(np.uint32), full[:256].view(np.uint32))     # a full threadgroup: defined
    assert np.all(np.isnan(shader[256:])) and np.all(np.isfinite(full[256:]))           # 24 live threads, 96 float4 to load


@pytest.mark.gpu
def test_hip_engine_against_the_reference_shader_outputs(golden, wax):
    """The HIP engine on the fixture's corpora: scores within 1e-5 of 1 - d_shader (unit queries: the Metal kernels assume one)
    and the ranking the parity rule allows."""
    for c in golden["distance_cases"]:
        if not c["unit_query"]:
            continue
        n, dims = c["rows"], c["dims"]
        x, q = _inputs(c)
        d_ref = _arr(c["distances"], np.float32)
        eng = wax.HIPVectorEngine(dimensions=dims)
        ids = np.arange(n, dtype=np.uint64) + 1000
        eng.addBatch(ids, x)
        got_ids, got_scores = eng.searchArrays(q, n)
        order = np.lexsort((np.arange(n), d_ref))
        assert_parity(got_ids, got_scores, ids[order], (np.float32(1.0) - d_ref[order]), ctx=f"dims {dims}")
        eng.close()
    # end to end: the reference pipeline's top-k (distance kernel + GPU reduction) against wax_hip_search on the same store
    for c in golden["search_cases"]:
        x, q = oracle.formula_rows(c["seed"], c["rows"], c["dims"]), oracle.formula_unit_query(c["seed"], c["dims"])
        idx, dist = _arr(c["indices"], np.uint32), _arr(c["distances"], np.float32)
        eng = wax.HIPVectorEngine(dimensions=c["dims"])
        ids = np.arange(c["rows"], dtype=np.uint64) + 500
        eng.addBatch(ids, x)
        got_ids, got_scores = eng.searchArrays(q, c["k"])
        assert_parity(got_ids, got_scores, idx.astype(np.uint64) + 500, np.float32(1.0) - dist, ctx=f"search {c['dims']}x{c['rows']}")
        eng.close()
