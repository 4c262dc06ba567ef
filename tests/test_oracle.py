"""CPU tests that pin the oracle: against the reference's own test assertions
(tests/golden/reference_cases.json), against known-answer identities, and against its frozen
outputs (tests/golden/oracle_vectors.json)."""
import hashlib
import struct

import numpy as np
import pytest

import oracle
from helpers import OracleEngine, assert_parity, load_golden, run_reference_case

REF = load_golden("reference_cases.json")
VEC = load_golden("oracle_vectors.json")


@pytest.mark.parametrize("mode", [oracle.MODE_TRUTH_F64, oracle.MODE_METAL_F32])
@pytest.mark.parametrize("case", REF["engine_cases"], ids=lambda c: c["name"])
def test_reference_cases_hold_for_oracle(case, mode):
    run_reference_case(case, lambda m, d: OracleEngine(m, d, mode), oracle.normalize_l2)


def test_vector_math_cases():
    for c in REF["vector_math"]:
        assert oracle.is_normalized_l2(c["vector"]) == c["isNormalizedL2"]


def test_clamp_and_score():
    assert oracle.clamp_topk(0) == 1 and oracle.clamp_topk(-5) == 1
    assert oracle.clamp_topk(10) == 10
    assert oracle.clamp_topk(10 ** 6) == REF["constants"]["max_results"]
    assert oracle.score_from_distance(0, 0.25) == pytest.approx(0.75)
    assert oracle.score_from_distance(1, 0.25) == pytest.approx(-0.25)
    assert oracle.score_from_distance(2, 3.0) == pytest.approx(-3.0)
    for bad in (float("inf"), float("-inf"), float("nan")):
        assert oracle.score_from_distance(0, bad) == 0.0  # VectorMetric.swift:33


def test_dimension_mismatch_message():
    c = REF["dimension_mismatch"]
    eng = OracleEngine(0, c["dimensions"])
    eng.add(0, [1.0] * c["dimensions"])
    with pytest.raises(Exception) as ei:
        eng.search([1.0] * c["query_len"], 1)
    assert c["message"] in str(ei.value)


def test_minilm_fixture_is_an_eight_way_tie():
    fx = REF["minilm_fixture"]
    assert fx is not None and fx["all_values_equal_to"] == 1.0
    rows = np.full((fx["count"], fx["dimensions"]), fx["all_values_equal_to"], dtype=np.float32)
    assert hashlib.sha256(rows.astype("<f4").tobytes()).hexdigest() == fx["sha256_f32_le"]
    ids, scores, dist, idx = oracle.search(0, rows, np.arange(100, 108, dtype=np.uint64), rows[3], 5)
    assert list(ids) == [100, 101, 102, 103, 104]  # all tied => ascending row
    assert np.allclose(scores, 1.0, atol=1e-6)


def test_selection_orders():
    rng = np.random.default_rng(1)
    for n, k in [(1, 1), (5, 10), (100, 7), (1000, 24), (1000, 1000), (5000, 256)]:
        d = rng.standard_normal(n).astype(np.float32)
        d[rng.integers(0, n, n // 3)] = d[0]  # many exact ties
        ti, td = oracle.topk_heap(d, k, total=True)
        si, sd = oracle.topk_heap(d, k, use_sort=True)
        assert list(ti) == list(si) and np.array_equal(td, sd)       # total-order heap == brute-force sort
        hi, hd = oracle.topk_heap(d, k)                                # the reference's heap (:630-680)
        assert np.array_equal(hd, sd)                                  # same distances always
        m = len(sd)
        boundary_tied = m < n and np.sum(d == sd[-1]) > np.sum(sd == sd[-1])
        if not boundary_tied:
            assert list(hi) == list(si)                                # same ids unless the k boundary is tied
    # The reference heap's boundary-tie behaviour depends on heap layout, not on index:
    d = np.array([0.5, 0.1, 0.5, 0.5, 0.1], dtype=np.float32)
    assert list(oracle.topk_heap(d, 3)[0]) == [1, 4, 2]               # root (index 0) is evicted by index 4
    assert list(oracle.topk_heap(d, 3, total=True)[0]) == [1, 4, 0]   # adopted rule: (distance asc, index asc)
    # ...while a later equal value never displaces the root (`value >= heap[0] => skip`, :671)
    d = np.array([0.1, 0.2, 0.3, 0.3, 0.3], dtype=np.float32)
    assert list(oracle.topk_heap(d, 3)[0]) == [0, 1, 2]


def test_metal_f32_and_f64_truth_agree_within_tolerance():
    for d in (128, 383, 384, 390, 768):
        corpus = oracle.gaussian_unit_rows(0, 400, d)
        q = oracle.gaussian_unit_queries(1, d)[0]
        a = oracle.distances(0, corpus, q, oracle.MODE_METAL_F32)
        b = oracle.distances(0, corpus, q, oracle.MODE_TRUTH_F64)
        assert np.max(np.abs(a - b)) < 2e-6


def test_metal_kernel_known_answers():
    # hand-computable: SIMD4 path (D < 384)
    v = np.array([[3.0, 4.0, 0.0, 0.0, 0.0]], dtype=np.float32)  # |v| = 5, scalar tail of 1
    q = np.array([1.0, 0.0, 0.0, 0.0, 0.0], dtype=np.float32)
    d = oracle.distances(0, v, q, oracle.MODE_METAL_F32)
    assert d[0] == np.float32(1.0) - np.float32(3.0) / np.float32(5.0)
    # zero row => similarity 0 => distance 1 (CosineDistance.metal:225)
    z = np.zeros((1, 8), dtype=np.float32)
    assert oracle.distances(0, z, np.ones(8, dtype=np.float32), oracle.MODE_METAL_F32)[0] == 1.0
    assert oracle.distances(0, z, np.ones(8, dtype=np.float32))[0] == 1.0
    # dot / l2 conventions (USearch ip / l2sq)
    a = np.array([[1.0, 2.0, 3.0, 4.0]], dtype=np.float32)
    b = np.array([0.5, 0.5, 0.5, 0.5], dtype=np.float32)
    assert oracle.distances(1, a, b)[0] == pytest.approx(1.0 - 5.0)
    assert oracle.distances(2, a, b)[0] == pytest.approx(0.25 + 2.25 + 6.25 + 12.25)


def test_query_scale_invariance_of_truth_cosine():
    corpus = oracle.gaussian_unit_rows(0, 300, 384)
    q = oracle.gaussian_unit_queries(1, 384)[0]
    a = oracle.search(0, corpus, None, q, 10)
    b = oracle.search(0, corpus, None, q * np.float32(12.0), 10)
    assert list(a[0]) == list(b[0])
    assert np.max(np.abs(a[1] - b[1])) < 1e-6


def test_mv2v_layout_constants_and_roundtrip():
    c = REF["constants"]
    vec = oracle.gaussian_unit_rows(0, 5, 8)
    ids = np.array([7, 1, 2 ** 40, 3, 9], dtype=np.uint64)
    blob = oracle.mv2v_serialize(0, vec, ids)
    assert blob[:4].hex() == c["mv2v_magic_hex"]
    version, enc, sim, dim, count, vbytes = struct.unpack_from("<HBBIQQ", blob, 4)
    assert (version, enc, sim, dim, count, vbytes) == (c["mv2v_version"], c["mv2v_encoding_flat"], 0, 8, 5, 5 * 8 * 4)
    assert blob[28:36] == b"\x00" * 8
    assert len(blob) == c["mv2v_header_size"] + 5 * 8 * 4 + 8 + 5 * 8
    assert blob[36:36 + 160] == vec.astype("<f4").tobytes()
    assert struct.unpack_from("<Q", blob, 36 + 160)[0] == 40
    rc, v2, i2 = oracle.mv2v_parse(blob, 0, 8)
    assert rc == 0 and np.array_equal(v2, vec) and np.array_equal(i2, ids)
    # each validation of MetalVectorEngine.deserialize (:718-808) trips in order
    assert oracle.mv2v_parse(blob[:20])[0] == 1
    assert oracle.mv2v_parse(b"XXXX" + blob[4:])[0] == 2
    assert oracle.mv2v_parse(blob[:4] + b"\x02\x00" + blob[6:])[0] == 3
    assert oracle.mv2v_parse(blob[:6] + b"\x01" + blob[7:])[0] == 4
    assert oracle.mv2v_parse(blob, 1, 8)[0] == 5
    assert oracle.mv2v_parse(blob, 0, 16)[0] == 6
    assert oracle.mv2v_parse(blob[:30] + b"\x01" + blob[31:])[0] == 7
    assert oracle.mv2v_parse(blob[:-8])[0] in (9, 11)
    assert oracle.mv2v_parse(blob + b"\x00")[0] == 12


def test_put_embedding_wal_layout_constant():
    w = REF["constants"]["put_embedding_wal"]
    enc = bytes([0x04]) + struct.pack("<QI", w["frameId"], w["dimension"]) + np.asarray(w["vector"], "<f4").tobytes()
    assert enc.hex() == w["encoded_hex"]


def test_deterministic_embedder_restatement():
    # FNV-1a 64 of "a" is a published constant; LCG constants are Knuth's MMIX
    v = oracle.deterministic_embed("a", 4, normalize=False)
    h = 0xAF63DC4C8601EC8C
    state, exp = h, []
    for _ in range(4):
        state = (state * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        s = state - (1 << 64) if state >= (1 << 63) else state
        exp.append(np.float32(np.float32(s) / np.float32(2 ** 63 - 1)))
    assert np.array_equal(v, np.array(exp, dtype=np.float32))
    u = oracle.deterministic_embed("doc-1", 384)
    assert oracle.is_normalized_l2(u, 1e-5)
    assert not np.array_equal(u, oracle.deterministic_embed("doc-2", 384))


def test_tie_pattern_has_period_256_duplicates():
    t = oracle.tie_pattern(0, 600, 128)
    assert np.array_equal(t[0], t[256]) and np.array_equal(t[1], t[257])
    assert t[0, 0] == 0.0 and t[0, 1] == np.float32(1 / 255.0)
    q = np.abs(oracle.gaussian_unit_queries(1, 128)[0])
    ids, scores, dist, rows = oracle.search(0, t, None, q, 4)
    assert rows[1] == rows[0] + 256 and dist[0] == dist[1]  # exact tie resolved by ascending row


def test_generators_are_shard_invariant():
    full = oracle.gaussian_unit_rows(0, 200000, 8)
    part = oracle.gaussian_unit_rows(65000, 70000, 8)
    assert np.array_equal(full[65000:135000], part)


@pytest.mark.parametrize("case", VEC["cases"], ids=lambda c: c["name"])
def test_frozen_oracle_vectors(case):
    n, d, k = case["n"], case["d"], case["k"]
    if case["generator"] == "gauss":
        corpus, q = oracle.gaussian_unit_rows(0, n, d), oracle.gaussian_unit_queries(3, d)
    elif case["generator"] == "lcg":
        corpus = np.stack([oracle.deterministic_embed(f"doc-{i}", d) for i in range(n)])
        q = np.stack([oracle.deterministic_embed(f"query-{i}", d) for i in range(3)])
    else:
        corpus, q = oracle.tie_pattern(0, n, d), np.abs(oracle.gaussian_unit_queries(3, d))
    assert hashlib.sha256(corpus.astype("<f4").tobytes()).hexdigest() == case["corpus_sha256"]
    assert hashlib.sha256(q.astype("<f4").tobytes()).hexdigest() == case["queries_sha256"]
    for qi, exp in enumerate(case["results"]):
        ids, scores, _, _ = oracle.search(case["metric"], corpus, None, q[qi], k)
        assert [int(x) for x in ids] == exp["ids"]
        assert np.array_equal(scores, np.asarray(exp["scores"], dtype=np.float32))


def test_multithreaded_baseline_matches_truth():
    corpus = oracle.gaussian_unit_rows(0, 20000, 384)
    q = oracle.gaussian_unit_queries(1, 384)[0]
    ids, scores, dist, rows = oracle.search(0, corpus, None, q, 10)
    for threads in (1, 3, oracle.max_threads()):
        mi, md = oracle.scan_topk_mt(0, corpus, q, 10, threads)
        assert_parity(mi, 1.0 - md, rows, scores, ctx=f"mt{threads}")


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_tuned_baseline_matches_truth(metric):
    """The FMA / metric-specialised baseline kernel (bench.py cpu_baseline variants) returns the truth's ranking."""
    corpus = oracle.gaussian_unit_rows(0, 20000, 384) * np.float32(1.0 if metric == 0 else 1.7)
    q = oracle.gaussian_unit_queries(1, 384)[0]
    ids, scores, dist, rows = oracle.search(metric, corpus, None, q, 10)
    for threads in (1, 3, oracle.max_threads()):
        mi, md = oracle.scan_topk_fast(metric, corpus, q, 10, threads)
        assert_parity(mi, oracle.scores_from_distances(metric, md), rows, scores, ctx=f"fast m{metric} t{threads}")
    odd = oracle.gaussian_unit_rows(0, 500, 37)       # dims not a multiple of the 16-lane unroll
    qo = oracle.gaussian_unit_queries(1, 37)[0]
    _, s2, _, r2 = oracle.search(metric, odd, None, qo, 5)
    mi, md = oracle.scan_topk_fast(metric, odd, qo, 5, 2)
    assert_parity(mi, oracle.scores_from_distances(metric, md), r2, s2, ctx=f"fast odd m{metric}")


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_batched_oracle_is_bit_identical_to_single_query_oracle(metric):
    """oracle.search_batch (what the GPU tests use at the BASELINE sizes) == oracle.search, query by query."""
    corpus = oracle.gaussian_unit_rows(0, 9000, 96)
    corpus[17] = 0.0                                   # a zero row (cosine: similarity 0)
    corpus[100] = corpus[3]                            # an exact duplicate: tie by ascending row
    qs = oracle.gaussian_unit_queries(9, 96)
    rows, dist, counts = oracle.search_batch(metric, corpus, qs, 20)
    for i, q in enumerate(qs):
        ids, scores, dd, rr = oracle.search(metric, corpus, None, q, 20)
        assert counts[i] == len(rr)
        assert np.array_equal(rows[i, :counts[i]], rr) and np.array_equal(dist[i, :counts[i]], dd)
        assert np.array_equal(oracle.scores_from_distances(metric, dd), scores)
    small = corpus[:5]
    rows, dist, counts = oracle.search_batch(metric, small, qs[:2], 20)   # k > n
    assert rows.shape == (2, 5) and np.all(counts == 5)


def test_numa_sample_helpers():
    x = oracle.numa_sample(1000, 24, 3)
    assert x.shape == (1000, 24) and not x.any()
    src = oracle.gaussian_unit_rows(0, 1000, 24)
    oracle.copy_rows(x, src, 3)
    assert np.array_equal(x, src)


# ---- reciprocal-rank fusion (HybridSearch.swift:25-52) ----

@pytest.mark.parametrize("case", REF["rrf_cases"], ids=lambda c: c["name"])
def test_rrf_two_list_reference_cases(case):
    ids, scores, best_rank, sources = oracle.rrf_fuse_two(case["text"], case["vector"], case["k"], case["alpha"])
    exp = case["expect"]
    if "count" in exp:
        assert len(ids) == exp["count"]
    if "idSet" in exp:
        assert set(ids.tolist()) == set(exp["idSet"])
    if "first" in exp:
        assert int(ids[0]) == exp["first"]
    assert np.all(np.diff(scores) <= 0)


@pytest.mark.parametrize("case", REF["rrf_multi_cases"], ids=lambda c: c["name"])
def test_rrf_multi_list_reference_cases(case):
    lists = [(l["weight"], l["frameIds"]) for l in case["lists"]]
    a = oracle.rrf_fuse(lists, case["k"])
    b = oracle.rrf_fuse(lists, case["k"])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))                       # idempotent (:5-26)
    rev = oracle.rrf_fuse(lists[::-1], case["k"])
    assert set(a[0].tolist()) == set(rev[0].tolist())                            # same id set under permutation (:28-39)


def test_rrf_known_answers():
    """Hand-computed: k = 60, lists (1.0: [1, 2]), (0.5: [2, 3]) -> 2: 1/62 + 0.5/61, 1: 1/61, 3: 0.5/62; a list with
    weight <= 0 is skipped; ties break by bestRank then id; k < 0 behaves like 0; duplicates inside a list add twice."""
    ids, scores, best, src = oracle.rrf_fuse([(1.0, [1, 2]), (0.5, [2, 3])], 60)
    f = np.float32
    assert ids.tolist() == [2, 1, 3]
    assert scores[0] == f(f(1.0) / f(62)) + f(f(0.5) / f(61)) and scores[1] == f(1.0) / f(61) and scores[2] == f(0.5) / f(62)
    assert best.tolist() == [1, 1, 2] and src.tolist() == [3, 1, 2]
    ids, _, _, _ = oracle.rrf_fuse([(0.0, [9]), (-1.0, [8]), (1.0, [7])], 60)
    assert ids.tolist() == [7]
    # equal scores: the smaller bestRank wins, then the smaller id
    ids, scores, best, _ = oracle.rrf_fuse([(1.0, [5, 4]), (1.0, [4, 5])], 10)
    assert scores[0] == scores[1] and ids.tolist() == [4, 5] and best.tolist() == [1, 1]
    a = oracle.rrf_fuse([(1.0, [3, 1, 2])], -5)
    b = oracle.rrf_fuse([(1.0, [3, 1, 2])], 0)
    assert np.array_equal(a[1], b[1]) and a[1][0] == np.float32(1.0)
    ids, scores, _, _ = oracle.rrf_fuse([(1.0, [6, 6])], 1)
    assert ids.tolist() == [6] and scores[0] == f(f(1.0) / f(2)) + f(f(1.0) / f(3))
    assert len(oracle.rrf_fuse([], 60)[0]) == 0 and len(oracle.rrf_fuse([(1.0, [])], 60)[0]) == 0
