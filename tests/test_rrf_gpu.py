"""Reciprocal-rank fusion on the device (wax_amd/csrc/rrf.hip) against the oracle's restatement of
HybridSearch.rrfFusion (HybridSearch.swift:25-52) and the reference's own assertions (HybridSearchTests.swift,
DeterminismPropertyTests.swift): ids, f32 scores, bestRank and source masks bit for bit."""
import ctypes

import numpy as np
import pytest

import oracle
from helpers import load_golden

pytestmark = pytest.mark.gpu
REF = load_golden("reference_cases.json")


@pytest.fixture(scope="module")
def wax():
    import wax_amd
    if not wax_amd.HIPVectorEngine.isAvailable():
        pytest.skip("no gfx950 device")
    return wax_amd


def _same(dev, ora):
    assert all(np.array_equal(a, b) for a, b in zip(dev, ora)), (dev, ora)


@pytest.mark.parametrize("case", REF["rrf_cases"], ids=lambda c: c["name"])
def test_rrf_reference_cases_two_lists(wax, case):
    text = [(i, 0.5) for i in case["text"]]
    vec = [(i, 0.5) for i in case["vector"]]
    merged = wax.HybridSearch.rrfFusion(textResults=text, vectorResults=vec, k=case["k"], alpha=case["alpha"])
    exp = case["expect"]
    if "count" in exp:
        assert len(merged) == exp["count"]
    if "idSet" in exp:
        assert {m[0] for m in merged} == set(exp["idSet"])
    if "first" in exp:
        assert merged[0][0] == exp["first"]
    ora = oracle.rrf_fuse_two(case["text"], case["vector"], case["k"], case["alpha"])
    assert [m[0] for m in merged] == ora[0].tolist() and [np.float32(m[1]) for m in merged] == ora[1].tolist()


@pytest.mark.parametrize("case", REF["rrf_multi_cases"], ids=lambda c: c["name"])
def test_rrf_reference_cases_multi(wax, case):
    lists = [(l["weight"], l["frameIds"]) for l in case["lists"]]
    a = wax.HybridSearch.rrfFusionArrays(lists, case["k"])
    b = wax.HybridSearch.rrfFusionArrays(lists, case["k"])
    _same(a, b)
    _same(a, oracle.rrf_fuse(lists, case["k"]))
    rev = wax.HybridSearch.rrfFusionArrays(lists[::-1], case["k"])
    assert set(a[0].tolist()) == set(rev[0].tolist())


def test_rrf_random_lists_bit_exact(wax):
    """Random lanes: overlapping ids, duplicates INSIDE a lane (same and different 256-chunks), weights <= 0, k <= 0,
    lists up to the 4 096-entry limit, huge ids."""
    rng = np.random.default_rng(12)
    for trial in range(40):
        n_lists = int(rng.integers(1, 9))
        budget = 4096
        lists = []
        for l in range(n_lists):
            n = int(rng.integers(0, min(budget, 1500) + 1))
            budget -= n
            pool = rng.integers(0, max(4, int(rng.integers(4, 3000))), size=n).astype(np.uint64)
            if trial % 3 == 0:
                pool = pool * np.uint64(0x9E3779B97F4A7C15) + np.uint64(trial)        # spread over the whole u64 range
                pool[pool == np.uint64(0xFFFFFFFFFFFFFFFF)] = np.uint64(7)
            if trial % 2 == 0 and n > 0:                                              # unique inside the lane, like real search results
                pool = np.unique(pool)[rng.permutation(len(np.unique(pool)))]
            w = float(rng.choice([1.0, 0.5, 0.25, 0.7, 0.0, -0.3, 1e-3]))
            lists.append((w, pool))
        k = int(rng.choice([60, 0, -3, 1, 1000]))
        _same(wax.HybridSearch.rrfFusionArrays(lists, k), oracle.rrf_fuse(lists, k))
    with pytest.raises(wax.CapacityExceeded):
        wax.HybridSearch.rrfFusionArrays([(1.0, np.arange(4097, dtype=np.uint64))], 60)
    assert len(wax.HybridSearch.rrfFusionArrays([], 60)[0]) == 0
    assert wax.HybridSearch.rrfFusion(lists=[(1.0, [])], k=60) == []


def test_rrf_batch_device_with_search_hits(wax):
    """The batched service shape: the vector lane is the hit array of wax_hip_search_batch_hits_device, still in HBM
    (pitch 2, padding skipped), the text lane an uploaded id matrix with per-query counts; one launch fuses every query;
    truncation to out_stride keeps the best."""
    import torch
    dev = torch.device("cuda", 0)
    n, dims, nq, k = 50_000, 128, 300, 30
    corpus = oracle.gaussian_unit_rows(6, n, dims)
    ids = np.arange(n, dtype=np.uint64) * 5 + 2
    eng = wax.HIPVectorEngine(dimensions=dims)
    eng.addBatch(ids, corpus)
    queries = oracle.gaussian_unit_queries(nq, dims, seed=9)
    dq = torch.from_numpy(queries).to(dev)
    stride = 40                                                     # wider than k: rows are padded with (KEY_PAD, UInt64.max)
    hits = torch.empty((nq, stride, 2), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, hits.data_ptr(), stride, st)
    rng = np.random.default_rng(3)
    text = rng.choice(ids, size=(nq, 50)).astype(np.uint64)
    tcnt = rng.integers(0, 51, size=nq).astype(np.uint32)
    d_text = torch.from_numpy(text.view(np.int64)).to(dev)
    d_tcnt = torch.from_numpy(tcnt.view(np.int32)).to(dev)
    out_stride = 64
    o_ids = torch.empty((nq, out_stride), dtype=torch.int64, device=dev)
    o_scores = torch.empty((nq, out_stride), dtype=torch.float32, device=dev)
    o_rank = torch.empty((nq, out_stride), dtype=torch.int32, device=dev)
    o_src = torch.empty((nq, out_stride), dtype=torch.int32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    lanes = [(d_text.data_ptr(), d_tcnt.data_ptr(), 50, 1, 0.4), (hits.data_ptr() + 8, 0, stride, 2, 0.6)]
    wax.HybridSearch.rrfFusionBatchDevice(lanes, nq, 60, o_ids.data_ptr(), o_scores.data_ptr(), out_stride, o_rank.data_ptr(),
                                          o_src.data_ptr(), o_cnt.data_ptr(), st)
    torch.cuda.synchronize()
    h = hits.cpu().numpy()
    g_ids, g_scores = o_ids.cpu().numpy().view(np.uint64), o_scores.cpu().numpy()
    g_rank, g_src, g_cnt = o_rank.cpu().numpy().view(np.uint32), o_src.cpu().numpy().view(np.uint32), o_cnt.cpu().numpy()
    for q in range(nq):
        vec_ids = h[q, :k, 1].view(np.uint64)
        assert np.all(h[q, k:, 1] == -1)
        ora = oracle.rrf_fuse([(np.float32(0.4), text[q, :tcnt[q]]), (np.float32(0.6), vec_ids)], 60)
        m = min(len(ora[0]), out_stride)
        assert g_cnt[q] == m
        assert np.array_equal(g_ids[q, :m], ora[0][:m]) and np.array_equal(g_scores[q, :m], ora[1][:m]), q
        assert np.array_equal(g_rank[q, :m], ora[2][:m]) and np.array_equal(g_src[q, :m], ora[3][:m]), q
        assert np.all(g_ids[q, m:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(g_scores[q, m:] == 0)
    # a narrow output keeps the head of the same ranking
    o2 = torch.empty((nq, 5), dtype=torch.int64, device=dev)
    s2 = torch.empty((nq, 5), dtype=torch.float32, device=dev)
    wax.HybridSearch.rrfFusionBatchDevice(lanes, nq, 60, o2.data_ptr(), s2.data_ptr(), 5, stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(o2.cpu().numpy().view(np.uint64)[:, :5], g_ids[:, :5])
    eng.close()


def test_rrf_argument_errors(wax):
    """The limits of include/wax_hip.h are enforced with the reference's error kinds, not by overrunning LDS."""
    import torch
    dev = torch.device("cuda", 0)
    ids = torch.zeros((4, 8), dtype=torch.int64, device=dev)
    out_i = torch.zeros((4, 8), dtype=torch.int64, device=dev)
    out_s = torch.zeros((4, 8), dtype=torch.float32, device=dev)
    lane = (ids.data_ptr(), 0, 8, 1, 1.0)
    with pytest.raises(wax.WaxError):                      # nine lanes
        wax.HybridSearch.rrfFusionBatchDevice([lane] * 9, 4, 60, out_i.data_ptr(), out_s.data_ptr(), 8)
    with pytest.raises(wax.CapacityExceeded):              # 8 x 600 entries per query
        wax.HybridSearch.rrfFusionBatchDevice([(ids.data_ptr(), 0, 600, 1, 1.0)] * 8, 4, 60, out_i.data_ptr(), out_s.data_ptr(), 8)
    with pytest.raises(wax.WaxError):                      # no output
        wax.HybridSearch.rrfFusionBatchDevice([lane], 4, 60, 0, out_s.data_ptr(), 8)
    with pytest.raises(wax.WaxError):                      # a lane without ids
        wax.HybridSearch.rrfFusionBatchDevice([(0, 0, 8, 1, 1.0)], 4, 60, out_i.data_ptr(), out_s.data_ptr(), 8)
    wax.HybridSearch.rrfFusionBatchDevice([lane], 0, 60, out_i.data_ptr(), out_s.data_ptr(), 8)   # no queries: nothing to do
    # every lane skipped (weights <= 0): all rows come back empty
    cnt = torch.full((4,), 7, dtype=torch.int32, device=dev)
    wax.HybridSearch.rrfFusionBatchDevice([(ids.data_ptr(), 0, 8, 1, 0.0)], 4, 60, out_i.data_ptr(), out_s.data_ptr(), 8,
                                          d_out_counts=cnt.data_ptr())
    torch.cuda.synchronize()
    assert cnt.cpu().tolist() == [0, 0, 0, 0] and bool((out_i.cpu() == -1).all())
