"""The in-library multi-GPU engine (wax_hip_engine_create_sharded) against ONE single-device engine on the same
operations: identical results at every shard count — ids, scores, tie order, serialize bytes. On the 1-GPU test box the
shards share GPU 0 (duplicate ordinals are allowed for exactly this); everything except the xGMI hop itself is the code
an 8-GPU host runs: block layout, global-row keys, per-shard streams, peer-copy gather, device merge, rebalancing."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import oracle
from helpers import assert_parity

pytestmark = pytest.mark.gpu



@pytest.fixture(scope="module", autouse=True)
def _spread_small_stores():
    """These tests are about the multi-shard machinery on stores of a few thousand rows: the small-store rule is switched off for
    the handles they create (the variable is read at handle creation; test_small_store_rule_* set the rule back per handle
    through "shard_min_mb") — and restored afterwards, so that nothing else in the session (bench.py subprocesses) inherits it."""
    old = os.environ.get("WAX_HIP_SHARD_MIN_MB")
    os.environ["WAX_HIP_SHARD_MIN_MB"] = "0"
    yield
    if old is None:
        os.environ.pop("WAX_HIP_SHARD_MIN_MB", None)
    else:
        os.environ["WAX_HIP_SHARD_MIN_MB"] = old


@pytest.fixture(scope="module")
def wax(hip_lib):
    import wax_amd
    if hip_lib.wax_hip_device_count() == 0:
        pytest.skip("no HIP device on this host: the gpu-marked tests run on the MI355X box (pytest -m gpu)")
    assert hip_lib.wax_hip_available() == 1
    return wax_amd


def pair(wax, metric, dims, shards):
    one = wax.HIPVectorEngine(metric=wax.VectorMetric(metric), dimensions=dims)
    many = wax.HIPVectorEngine(metric=wax.VectorMetric(metric), dimensions=dims, devices=[0] * shards)
    assert many.shardCount == shards and one.shardCount == 1
    return one, many


def same_search(one, many, q, k):
    a, b = one.searchArrays(q, k), many.searchArrays(q, k)
    assert np.array_equal(a[0], b[0]), (k, a[0][:8], b[0][:8])
    assert np.array_equal(a[1], b[1]), k


@pytest.mark.parametrize("shards", [1, 2, 3])
@pytest.mark.parametrize("metric,dims", [(0, 384), (1, 100), (2, 64)])
def test_sharded_engine_equals_single_engine(wax, shards, metric, dims):
    rng = np.random.default_rng(100 * shards + dims)
    one, many = pair(wax, metric, dims, shards)
    n = 30_000
    corpus = oracle.gaussian_unit_rows(5, n, dims) * (np.float32(1.0) if metric == 0 else rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32))
    ids = rng.permutation(n).astype(np.uint64) * 7 + 3
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatch(ids[:20_000], corpus[:20_000])
    if shards > 1:
        info = [many.shardInfo(g) for g in range(shards)]
        assert [i[1] for i in info] == list(np.cumsum([0] + [i[2] for i in info[:-1]]))      # bases are prefix sums
        assert info[0][2] == -(-n // shards + 63) // 64 * 64 or info[0][2] == ((n + shards - 1) // shards + 63) // 64 * 64
    queries = oracle.gaussian_unit_queries(12, dims, seed=shards)
    for q in queries[:4]:
        for k in (1, 10, 192, 193, 1000):
            same_search(one, many, q, k)
    # mutations: single adds (staged appends), upserts of existing ids (any shard), a batch with repeated and existing ids,
    # removals from every shard, then the rest of the corpus
    for eng in (one, many):
        for i in range(20_000, 20_050):
            eng.add(int(ids[i]), corpus[i])
        eng.add(int(ids[17]), corpus[25_000])                       # upsert, first shard
        eng.add(int(ids[19_990]), corpus[25_001])                   # upsert, last occupied shard
        mix_ids = np.concatenate([ids[20_050:20_060], ids[5:7], ids[20_050:20_052]])
        mix_rows = np.concatenate([corpus[20_050:20_060], corpus[26_000:26_002], corpus[26_002:26_004]])
        eng.addBatch(mix_ids, mix_rows)
        for i in (0, 9_999, 10_000, 19_999, 20_055):
            eng.remove(int(ids[i]))
        eng.remove(123456789012)                                     # absent id: no-op
        eng.addBatch(ids[20_060:], corpus[20_060:])
    assert one.count == many.count
    for q in queries[4:8]:
        for k in (10, 300):
            same_search(one, many, q, k)
    assert one.serialize() == many.serialize()                       # byte-identical MV2V segment: same rows, same order
    # batched search (MFMA path inside every shard; merged per query on the first device)
    b1, b2 = one.searchBatch(queries, 10), many.searchBatch(queries, 10)
    assert np.array_equal(b1[0], b2[0]) and np.array_equal(b1[1], b2[1]) and np.array_equal(b1[2], b2[2])
    h1, h2 = one.searchBatchHits(queries, 25), many.searchBatchHits(queries, 25)
    assert np.array_equal(h1[0], h2[0])                              # keys carry the same GLOBAL rows
    # filtered search: allow-list spanning the shards, minScore
    allow = rng.choice(ids, 500, replace=False)
    for q in queries[8:10]:
        f1, f2 = one.searchFiltered(q, 20, frameIds=allow), many.searchFiltered(q, 20, frameIds=allow)
        assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])
        f1, f2 = one.searchFiltered(q, 20, minScore=0.1), many.searchFiltered(q, 20, minScore=0.1)
        assert np.array_equal(f1[0], f2[0])
    # a long allow-list: every shard resolves it through its own id -> row table in HBM
    long_allow = rng.choice(ids, 9000, replace=False)
    before = many.getTuning("filter_device_searches")
    f1, f2 = one.searchFiltered(queries[8], 40, frameIds=long_allow), many.searchFiltered(queries[8], 40, frameIds=long_allow)
    assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])
    assert many.getTuning("filter_device_searches") - before == shards
    # round trip through a fresh sharded engine
    blob = many.serialize()
    again = wax.HIPVectorEngine(metric=wax.VectorMetric(metric), dimensions=dims, devices=[0] * shards)
    again.deserialize(blob)
    assert again.count == one.count and again.serialize() == blob
    for q in queries[10:]:
        a, b = one.searchArrays(q, 10), again.searchArrays(q, 10)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # and against the oracle
    m = one.count
    blob_rows = np.frombuffer(blob, dtype="<f4", count=m * dims, offset=36).reshape(m, dims)
    blob_ids = np.frombuffer(blob, dtype="<u8", count=m, offset=36 + m * dims * 4 + 8)
    got = many.searchArrays(queries[0], 10)
    e_ids, e_scores, _, _ = oracle.search(metric, blob_rows, blob_ids, queries[0], 10)
    x = oracle.search(metric, blob_rows, blob_ids, queries[0], 18)[1]
    assert_parity(got[0], got[1], e_ids, e_scores, x, f"sharded x{shards} m{metric}")
    for eng in (one, many, again):
        eng.close()


def test_sharded_tie_order_and_growth_without_reserve(wax):
    """Exact duplicates spread over the shards resolve by ascending GLOBAL row exactly as in one engine, also while the
    layout grows by doubling (no reserve): the period-256 tie corpus of MetalVectorEngineBenchmark.swift:33-38."""
    dims = 128
    one, many = pair(wax, 0, dims, 3)
    ties = oracle.tie_pattern(0, 9_000, dims)
    q = np.abs(oracle.gaussian_unit_queries(3, dims))
    done = 0
    for step in (100, 700, 64, 3000, 1, 5135):
        chunk = ties[done:done + step]
        ids = np.arange(done, done + step, dtype=np.uint64)
        for eng in (one, many):
            if step == 1:
                eng.add(int(ids[0]), chunk[0])
            else:
                eng.addBatch(ids, chunk)
        done += step
        for qq in q:
            same_search(one, many, qq, 30)
    assert many.getTuning("rebalances") >= 1 and many.getTuning("shards") == 3
    assert one.serialize() == many.serialize()
    for r in (0, 4500, 8999, 256):
        one.remove(r), many.remove(r)
    for qq in q:
        same_search(one, many, qq, 40)
    assert one.serialize() == many.serialize()
    one.close(), many.close()


def test_sharded_pipelined_tickets_and_concurrency(wax):
    dims, n = 384, 60_000
    corpus = oracle.gaussian_unit_rows(8, n, dims)
    one, many = pair(wax, 0, dims, 2)
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatch(np.arange(n, dtype=np.uint64), corpus)
    queries = oracle.gaussian_unit_queries(40, dims, seed=4)
    expect = [one.searchArrays(q, 10) for q in queries]
    tickets = []
    out = []
    for q in queries:                                                 # more tickets than the handle's slots: collect as we go
        if len(tickets) >= 6:
            out.append(many.collect(tickets.pop(0), 10))
        tickets.append(many.submit(q, 10))
    while tickets:
        out.append(many.collect(tickets.pop(0), 10))
    for (ids, scores), (e_ids, e_scores) in zip(out, expect):
        assert np.array_equal(ids, e_ids) and np.array_equal(scores, e_scores)
    t = many.submit(queries[0], 10)                                   # a writer is refused while this thread holds a ticket
    with pytest.raises(wax.EncodingError):
        many.add(10 ** 9, corpus[0])
    many.collect(t, 10)
    errors = []

    def reader(tid):
        try:
            for i in range(30):
                ids, scores = many.searchArrays(queries[(tid + i) % 40], 10)
                assert len(ids) == 10 and np.all(np.diff(scores) <= 0)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    def writer():
        try:
            for i in range(40):
                many.add(10 ** 9 + i, corpus[i])
                if i % 8 == 0:
                    many.remove(10 ** 9 + i)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    ts = [threading.Thread(target=reader, args=(t_,)) for t_ in range(3)] + [threading.Thread(target=writer)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    assert not errors, errors
    assert many.count == n + 35
    one.close(), many.close()


def test_sharded_device_rows_and_single_device_entry_points(wax):
    import torch
    dims, n = 128, 20_000
    dev = torch.device("cuda", 0)
    rows = torch.nn.functional.normalize(torch.randn((n, dims), device=dev), dim=1).contiguous()
    one, many = pair(wax, 0, dims, 3)
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatchDevice(np.arange(n, dtype=np.uint64), rows)
    assert sum(many.shardInfo(g)[2] for g in range(3)) == n and many.count == n
    q = rows[777].cpu().numpy()
    same_search(one, many, q, 10)
    assert many.searchArrays(q, 1)[0][0] == 777
    with pytest.raises(wax.EncodingError):
        many.addBatchDevice(np.arange(5, dtype=np.uint64), rows[:5].contiguous())       # ids already present
    for call in (lambda: many.setRowBase(5), lambda: many.timeStreamRead(1), lambda: many.timeScanKernel(q, 10, 1),
                 lambda: many.searchShardDevice(q, 10, rows.data_ptr())):
        with pytest.raises(wax.EncodingError) as ei:
            call()
        assert "sharded" in str(ei.value)
    with pytest.raises(wax.InvalidToc) as ei:
        wax.HIPVectorEngine(dimensions=dims, devices=[0, 99])
    assert "HIP device 99 not available" in str(ei.value)
    with pytest.raises(wax.InvalidToc) as ei:
        wax.HIPVectorEngine(dimensions=dims, devices=[])
    assert "device list" in str(ei.value)
    one.close(), many.close()


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_sharded_batched_device_resident_entry_points(wax, shards):
    """wax_hip_search_batch_hits_device / _submit_device / _collect_device on a sharded handle (round 3: pooled per-shard
    workspaces, no thread and no allocation per call): queries and hits in the first device's HBM; the answer equals ONE
    engine's bit for bit — through the one-pass MFMA pipeline, the slab pipeline (small shards), the loop path, wider and
    narrower strides, k above the device-merge limit, more than 1024 queries, pipelined tickets, and an empty handle."""
    import torch
    dev = torch.device("cuda", 0)
    KEY_PAD = (1 << 63) - 1
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(eng, dq, nq, k, stride):
        out = torch.full((nq, stride, 2), 5, dtype=torch.int64, device=dev)
        guard = torch.full((8,), 77, dtype=torch.int64, device=dev)
        eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, out.data_ptr(), stride, stream)
        assert torch.all(guard == 77)
        return out.cpu().numpy()

    for dims, n, nq in [(384, 160_000, 300), (384, 20_000, 200), (768, 150_000, 130), (100, 6_000, 20), (384, 90_000, 1500)]:
        one, many = pair(wax, 0, dims, shards)
        rows = torch.nn.functional.normalize(torch.randn((n, dims), device=dev, generator=torch.Generator(device=dev).manual_seed(n + dims)), dim=1).contiguous()
        ids = np.arange(n, dtype=np.uint64) * 3 + 11
        for eng in (one, many):
            eng.reserve(n)
            eng.addBatchDevice(ids, rows)
        queries = oracle.gaussian_unit_queries(nq, dims, seed=shards + n % 89)
        queries[1] = rows[4321].cpu().numpy()
        dq = torch.from_numpy(queries).to(dev)
        for k, stride in [(10, 10), (10, 16), (10, 4), (200, 200)]:
            a, b = run(one, dq, nq, k, stride), run(many, dq, nq, k, stride)
            assert np.array_equal(a, b), (shards, dims, n, nq, k, stride)
            w = min(k, stride)
            assert np.all(b[:, w:, 0] == KEY_PAD) and b[1, 0, 1] == 4321 * 3 + 11
        h_one, c_one = one.searchBatchHits(queries, 10)               # host-pointer form: same path underneath
        h_many, c_many = many.searchBatchHits(queries, 10)
        assert np.array_equal(h_one, h_many) and np.array_equal(c_one, c_many)
        # pipelined tickets (two in flight, as bench.py's config-5 secondary drives the handle)
        outs = [torch.empty((nq, 10, 2), dtype=torch.int64, device=dev) for _ in range(3)]
        tickets = []
        for i in range(5):
            if len(tickets) == 2:
                many.searchBatchCollectDevice(tickets.pop(0))
            tickets.append(many.searchBatchSubmitDevice(dq.data_ptr(), nq, 10, outs[i % 3].data_ptr(), 10, stream))
        t = many.submit(queries[0], 10)                               # single-query tickets interleave with batch tickets
        for bt in tickets:
            many.searchBatchCollectDevice(bt)
        many.collect(t, 10)
        ref = run(one, dq, nq, 10, 10)
        for o in outs:
            assert np.array_equal(o.cpu().numpy(), ref)
        one.close(), many.close()
    empty = wax.HIPVectorEngine(dimensions=64, devices=[0] * shards)
    dq = torch.from_numpy(oracle.gaussian_unit_queries(5, 64)).to(dev)
    assert np.all(run(empty, dq, 5, 10, 10)[:, :, 0] == KEY_PAD)
    empty.close()


@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_batch_resends_parts_rewritten_at_collect(wax, shards):
    """A shard's [nq][k] part goes to the first device right behind that shard's finish kernel; queries the certificate cannot
    prove are settled LATER, at collect (full retry of all survivors, or the exact path), rewriting rows of the part. The handle
    must send such a part again — with retries alone (no exact-path fallback on that shard) as much as with fallbacks. Corpus:
    200 near-duplicates (2e-4 apart: bf16 cannot order them, so the k' candidates of the first finish miss true neighbours and the
    rows written before the retry are WRONG) in the first shard, a 3 000-fold run of exact duplicates in the last. Both are
    settled by the full retry; with "batch_retry" = 0 both go to the exact path."""
    import torch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    n, dims, nq, k = 240_000, 384, 256, 10
    rows = torch.nn.functional.normalize(torch.randn((n, dims), device=dev, generator=torch.Generator(device=dev).manual_seed(5)), dim=1).contiguous()
    g = torch.Generator(device=dev).manual_seed(6)
    rows[1000:1200] = torch.nn.functional.normalize(rows[999] + 2e-4 * torch.randn((200, dims), device=dev, generator=g), dim=1)   # first shard: full retry
    rows[200_000:203_000] = rows[199_999]  # last shard: a 3 000-fold exact tie
    ids = np.arange(n, dtype=np.uint64) + 7
    one, many = pair(wax, 0, dims, shards)
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatchDevice(ids, rows)
    queries = oracle.gaussian_unit_queries(nq, dims, seed=77)
    queries[:6] = rows[999].cpu().numpy()
    queries[6:10] = rows[199_999].cpu().numpy()
    dq = torch.from_numpy(queries).to(dev)
    a = torch.empty((nq, k, 2), dtype=torch.int64, device=dev)
    b = torch.empty((nq, k, 2), dtype=torch.int64, device=dev)
    one.searchBatchHitsDevice(dq.data_ptr(), nq, k, a.data_ptr(), k, stream)
    # "batch_retry" = 2: the full retry driven from the HOST at collect time (round 3's rung; since round 4 the default, 1, runs it
    # on the device behind the finish kernel once the hint is armed, which settles such queries before the part is ever sent) —
    # the re-send logic this test is about needs rows rewritten at collect
    many.setTuning("batch_retry", 2)
    r0, f0 = many.getTuning("batch_retries"), many.getTuning("batch_fallbacks")
    many.searchBatchHitsDevice(dq.data_ptr(), nq, k, b.data_ptr(), k, stream)
    assert many.getTuning("batch_retries") > r0 and many.getTuning("batch_fallbacks") == f0    # settled by full retries alone
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    many.setTuning("batch_retry", 0)                                                            # ... and by the exact path alone
    b.zero_()
    many.searchBatchHitsDevice(dq.data_ptr(), nq, k, b.data_ptr(), k, stream)
    assert many.getTuning("batch_fallbacks") > f0 and np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    many.setTuning("batch_retry", 2)
    for i in (0, 7, 100):
        s_ids, _ = one.searchArrays(queries[i], k)
        assert np.array_equal(b.cpu().numpy()[i, :, 1].view(np.uint64), s_ids)
    # ticketed form, two in flight
    outs = [torch.empty((nq, k, 2), dtype=torch.int64, device=dev) for _ in range(2)]
    ts = [many.searchBatchSubmitDevice(dq.data_ptr(), nq, k, outs[i].data_ptr(), k, stream) for i in range(2)]
    for t in ts:
        many.searchBatchCollectDevice(t)
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), a.cpu().numpy())
    # two shards settled queries on the host side in the first batch, so the later collects ran the per-shard ladders side by
    # side on the handle's persistent workers
    assert many.getTuning("parallel_collects") >= 2
    # the default ladder with the hint armed: the uncertified queries are settled by the device-side retry kernel behind each
    # shard's finish kernel (all survivors re-scored there) — before the parts are sent, nothing rewritten at collect; same hits
    many.setTuning("batch_retry", 1)
    many.setTuning("retry_hint", 16)
    i0 = many.getTuning("batch_inline_retries")
    b.zero_()
    many.searchBatchHitsDevice(dq.data_ptr(), nq, k, b.data_ptr(), k, stream)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert many.getTuning("batch_inline_retries") - i0 >= 6
    one.close(), many.close()


def test_sharded_submit_beyond_the_slot_pool_and_concurrent_filtered_search(wax):
    """(1) A thread that pipelines more sharded submits than the handle's soft slot cap (8) without collecting gets fresh
    slots instead of waiting for itself while holding the read lock (round-2 advisor finding). (2) Filtered search visits
    the shards concurrently (persistent per-shard workers) and equals one engine's result, long allow-lists included."""
    dims, n = 128, 40_000
    corpus = oracle.gaussian_unit_rows(9, n, dims)
    one, many = pair(wax, 0, dims, 4)
    ids = np.arange(n, dtype=np.uint64) * 2 + 1
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatch(ids, corpus)
    queries = oracle.gaussian_unit_queries(20, dims, seed=6)
    tickets = [many.submit(q, 10) for q in queries]                   # 20 outstanding tickets, none collected yet
    got = [many.collect(t, 10) for t in tickets]
    for q, (g_ids, g_scores) in zip(queries, got):
        e_ids, e_scores = one.searchArrays(q, 10)
        assert np.array_equal(g_ids, e_ids) and np.array_equal(g_scores, e_scores)
    rng = np.random.default_rng(3)
    for n_allow in (5, 3000, 30_000):
        allow = rng.choice(ids, size=n_allow, replace=False)
        for q in queries[:3]:
            a = one.searchFiltered(q, 10, frameIds=allow)
            b = many.searchFiltered(q, 10, frameIds=allow)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), n_allow
    one.close(), many.close()


def test_rccl_exchange_on_a_one_rank_communicator():
    """exchange = 1: one ncclAllGather per query through librccl on a single-process communicator (ncclCommInitAll).
    A 1-GPU box can only form a 1-rank communicator (RCCL refuses duplicate devices — checked too); run in a subprocess
    so that a wedged collective cannot take the test session with it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import oracle, wax_amd as wax
if not wax.HIPVectorEngine.isAvailable():
    print("NO_GPU"); sys.exit(0)
dims, n = 384, 50000
corpus = oracle.gaussian_unit_rows(2, n, dims)
one = wax.HIPVectorEngine(dimensions=dims)
many = wax.HIPVectorEngine(dimensions=dims, devices=[0])
for e in (one, many):
    e.addBatch(np.arange(n, dtype=np.uint64), corpus)
many.setTuning("exchange", 1)
assert many.getTuning("exchange") == 1
for q in oracle.gaussian_unit_queries(20, dims):
    a, b = one.searchArrays(q, 10), many.searchArrays(q, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
assert many.getTuning("rccl_collectives") == 20
dup = wax.HIPVectorEngine(dimensions=dims, devices=[0, 0])
try:
    dup.setTuning("exchange", 1)
    print("DUP_ACCEPTED")
except wax.InvalidToc as ex:
    assert "ncclCommInitAll" in str(ex), ex
    print("DUP_REFUSED")
dup.addBatch(np.arange(100, dtype=np.uint64), corpus[:100])
assert dup.searchArrays(corpus[5], 1)[0][0] == 5          # the peer-copy exchange keeps working
print("RCCL_OK")
''' % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if "NO_GPU" in out.stdout:
        pytest.skip("no gfx950 device")
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_host_staged_gather_equals_device_gather(wax):
    """"gather" = 2 — what a handle falls back to when a device pair lacks peer access: every shard answers through its own
    ticket, k hits each come down, the host merges by key. Same ids, scores and tie order as the device gather, and as one
    engine; the peer-access bookkeeping reports 0 of 0 pairs for shards that share a device."""
    dims, n = 384, 40_000
    corpus = oracle.gaussian_unit_rows(11, n, dims)
    corpus[100:140] = corpus[7]                                            # ties across what become different shards
    corpus[30_000:30_010] = corpus[7]
    ids = np.arange(n, dtype=np.uint64) * 3 + 1
    one, many = pair(wax, 0, dims, 3)
    for e in (one, many):
        e.reserve(n)
        e.addBatch(ids, corpus)
    assert many.getTuning("peer_pairs") == 0 and many.getTuning("peer_enabled") == 0 and many.getTuning("gather") == 0
    queries = list(oracle.gaussian_unit_queries(6, dims, seed=3)) + [corpus[7]]
    dev = [many.searchArrays(q, k) for q in queries for k in (1, 10, 60)]
    many.setTuning("gather", 2)
    assert many.getTuning("gather") == 2
    host = [many.searchArrays(q, k) for q in queries for k in (1, 10, 60)]
    ref = [one.searchArrays(q, k) for q in queries for k in (1, 10, 60)]
    for a, b, c in zip(dev, host, ref):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(b[0], c[0]) and np.array_equal(b[1], c[1])
    # pipelined tickets on the host-gather path
    tickets = [many.submit(q, 10) for q in queries[:4]]
    for t, q in zip(tickets, queries[:4]):
        got = many.collect(t, 10)
        exp = one.searchArrays(q, 10)
        assert np.array_equal(np.asarray(got[0], dtype=np.uint64), exp[0])
    many.setTuning("gather", 0)
    one.close(), many.close()


def test_sharded_batch_submit_runs_blocking_shard_submits_side_by_side(wax):
    """A batch the shards cannot take on the asynchronous MFMA pipeline (here dims % 64 != 0: the loop path, which finishes
    device work inside submit) is handed to the per-shard workers instead of being driven shard after shard by the calling
    thread; same hits as one engine. A batch that does take the MFMA pipeline stays on the calling thread once the mirrors
    exist."""
    n = 20_000
    for dims, expect_parallel in ((100, True), (384, False)):
        corpus = oracle.gaussian_unit_rows(4, n, dims)
        one, many = pair(wax, 0, dims, 3)
        for e in (one, many):
            e.addBatch(np.arange(n, dtype=np.uint64), corpus)
        q = oracle.gaussian_unit_queries(32, dims, seed=5)
        many.searchBatch(q, 10)                                               # builds mirrors where they apply
        before = many.getTuning("parallel_submits")
        a, b = one.searchBatch(q, 10), many.searchBatch(q, 10)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))               # ids, scores, counts
        assert (many.getTuning("parallel_submits") > before) == expect_parallel, (dims, before, many.getTuning("parallel_submits"))
        one.close(), many.close()


# ---------------------------------------------------------------------------
# More than one physical GPU. The 1-GPU test box skips these; they are here for the first node that has two (the driver's
# scaling box): distinct ordinals exercise what duplicate ordinals cannot — per-device kernel attributes (the dynamic-LDS
# opt-in of the GEMM kernels is per device), peer access, peer copies over xGMI and a G-rank RCCL communicator.

def _need_gpus(hip_lib, n):
    if hip_lib.wax_hip_device_count() < n:
        pytest.skip(f"needs {n} GPUs ({hip_lib.wax_hip_device_count()} visible)")


def test_engine_on_a_non_zero_ordinal(wax, hip_lib):
    """A single-device engine on device 1: fused scan, general selection and both GEMM pipelines (whose > 64 KB dynamic
    LDS needs hipFuncSetAttribute on THAT device) against the same engine on device 0."""
    _need_gpus(hip_lib, 2)
    for dims in (384, 768):
        n = 70_000
        corpus = oracle.gaussian_unit_rows(21, n, dims)
        e0 = wax.HIPVectorEngine(dimensions=dims, device=0)
        e1 = wax.HIPVectorEngine(dimensions=dims, device=1)
        for e in (e0, e1):
            e.addBatch(np.arange(n, dtype=np.uint64), corpus)
        qs = oracle.gaussian_unit_queries(64, dims, seed=9)
        for q in qs[:4]:
            for k in (10, 300):
                a, b = e0.searchArrays(q, k), e1.searchArrays(q, k)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert all(np.array_equal(x, y) for x, y in zip(e0.searchBatch(qs, 10), e1.searchBatch(qs, 10)))
        e0.close(), e1.close()


def test_sharded_engine_over_distinct_devices(wax, hip_lib):
    """Every visible GPU behind one handle: peer access is reported per pair, results equal one engine for single queries,
    large k (host merge), batches, and — through RCCL — a communicator with as many ranks as shards."""
    _need_gpus(hip_lib, 2)
    g = min(hip_lib.wax_hip_device_count(), 8)
    dims, n = 384, 200_000
    corpus = oracle.gaussian_unit_rows(31, n, dims)
    ids = np.arange(n, dtype=np.uint64)
    one = wax.HIPVectorEngine(dimensions=dims)
    many = wax.HIPVectorEngine(dimensions=dims, devices=list(range(g)))
    for e in (one, many):
        e.reserve(n)
        e.addBatch(ids, corpus)
    assert many.getTuning("peer_pairs") == g - 1
    print(f"\n[multi-gpu] {g} devices, peer access on {many.getTuning('peer_enabled')} of {g - 1} pairs, gather mode {many.getTuning('gather')}")
    qs = oracle.gaussian_unit_queries(40, dims, seed=2)
    for mode in ("default", "host", "rccl"):
        if mode == "host":
            many.setTuning("gather", 2)
        if mode == "rccl":
            if many.getTuning("peer_enabled") == g - 1:
                many.setTuning("gather", 0)
            many.setTuning("exchange", 1)
            assert many.getTuning("rccl_ranks") == g
        for q in qs[:6]:
            for k in (10, 500):
                same_search(one, many, q, k)
    many.setTuning("exchange", 0)
    assert all(np.array_equal(x, y) for x, y in zip(one.searchBatch(qs, 10), many.searchBatch(qs, 10)))
    assert one.serialize() == many.serialize()
    one.close(), many.close()


def test_small_store_rule_keeps_a_small_store_on_one_device(wax):
    """Round 5: a store below "shard_min_mb" (default 64 MB of rows per shard block) is NOT spread over the devices — eight
    launches and a gather for 2 MB of scan each cost 12 x the single engine's latency at 10K rows (round-4 rehearsal). It lives on
    the first shard, a search is that engine's search behind one ticket lookup (within 1.5 x its latency), and when the store
    outgrows the block the next shard starts to fill: same answers throughout."""
    import time
    dims, shards = 384, 8
    one = wax.HIPVectorEngine(dimensions=dims)
    many = wax.HIPVectorEngine(dimensions=dims, devices=[0] * shards)
    many.setTuning("shard_min_mb", 64)
    assert many.getTuning("shard_min_mb") == 64
    n0 = 10_000
    corpus = oracle.gaussian_unit_rows(3, 60_000, dims)
    ids = np.arange(60_000, dtype=np.uint64) * 5 + 1
    for eng in (one, many):
        eng.reserve(n0)
        eng.addBatch(ids[:n0], corpus[:n0])
    block = many.getTuning("block_rows")
    assert block * dims * 4 >= 64 << 20 and (block - 64) * dims * 4 < 64 << 20
    assert [many.shardInfo(g)[2] for g in range(shards)] == [n0] + [0] * (shards - 1)
    queries = oracle.gaussian_unit_queries(400, dims, seed=77)
    s0 = many.getTuning("single_shard_searches")
    for q in queries[:8]:
        for k in (1, 10, 64, 65, 200):
            same_search(one, many, q, k)
    assert many.getTuning("single_shard_searches") - s0 == 40 and many.getTuning("ticket_searches") == 0

    def lat(eng):
        for q in queries[:50]:
            eng.searchArrays(q, 10)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for q in queries:
                eng.searchArrays(q, 10)
            best = min(best, (time.perf_counter() - t0) / len(queries))
        return best
    a, b = lat(one), lat(many)
    assert b <= 1.5 * a + 5e-6, (a, b)
    # batched search and filtered search on a handle whose other shards are empty
    b1, b2 = one.searchBatch(queries[:64], 10), many.searchBatch(queries[:64], 10)
    assert all(np.array_equal(x, y) for x, y in zip(b1, b2))
    f1, f2 = one.searchFiltered(queries[0], 20, frameIds=ids[:500]), many.searchFiltered(queries[0], 20, frameIds=ids[:500])
    assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])
    # growth past one block: the second shard starts to fill, searches fan out, answers stay the single engine's
    for eng in (one, many):
        eng.addBatch(ids[n0:], corpus[n0:])
    counts = [many.shardInfo(g)[2] for g in range(shards)]
    assert counts[0] == block and counts[1] == 60_000 - block and sum(counts[2:]) == 0, counts
    t0 = many.getTuning("ticket_searches")
    for q in queries[8:14]:
        for k in (10, 100):
            same_search(one, many, q, k)
    assert many.getTuning("ticket_searches") - t0 == 12
    assert one.serialize() == many.serialize()
    one.close(), many.close()


@pytest.mark.parametrize("shards", [2, 5])
def test_ticket_path_equals_device_gather(wax, shards):
    """Round 5 default for single queries on a sharded handle: per-shard tickets submitted side by side by the persistent
    workers + a host merge by key ("ticket_path" = 1) against the round-2 device gather (0): identical ids and scores for every k
    class (k-way merge in the scan kernel, wave lists, general selection beyond 192), pipelined tickets, concurrent callers."""
    dims, n = 384, 40_000
    one, many = pair(wax, 0, dims, shards)
    corpus = oracle.gaussian_unit_rows(9, n, dims)
    corpus[1000:1040] = corpus[999]                                  # a 41-fold exact tie inside the first shard
    corpus[n // shards - 3:n // shards + 3] = corpus[7]              # and a 7-fold one across the first shard boundary
    ids = np.arange(n, dtype=np.uint64) + 11
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatch(ids, corpus)
    queries = np.concatenate([oracle.gaussian_unit_queries(10, dims, seed=5), corpus[[7, 999]]])
    got = {}
    for mode in (1, 0):
        many.setTuning("ticket_path", mode)
        res = []
        for q in queries:
            for k in (1, 10, 64, 65, 192, 193, 700):
                a, b = one.searchArrays(q, k), many.searchArrays(q, k)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (mode, k)
                res.append(b)
        # six tickets in flight from one thread
        tickets = [many.submit(q, 10) for q in queries[:6]]
        for t, q in zip(tickets, queries[:6]):
            r = many.collect(t, 10)
            e = one.searchArrays(q, 10)
            assert np.array_equal(r[0], e[0]) and np.array_equal(r[1], e[1])
        got[mode] = res
    assert many.getTuning("ticket_searches") > 0
    many.setTuning("ticket_path", 1)
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(40):
                q = queries[rng.integers(len(queries))]
                k = int(rng.choice([1, 10, 100, 300]))
                a, b = one.searchArrays(q, k), many.searchArrays(q, k)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)
    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:1]
    one.close(), many.close()


def test_pace_gate_does_not_wait_for_workgroups_that_are_not_running(wax):
    """The 768-d filtering GEMM's pace gate keeps the G query groups that want the same tile within a few tiles of each other
    (L2 sharing). It must only ever wait for workgroups that are ON a CU: with several persistent GEMMs sharing one GPU — four shards
    on device 0 here, two batches in flight in production — the G workgroups of a tile column start far apart, and the round-4
    form (average over all G, 1 ms spin bound) made the early ones sleep through every check: 157 ms instead of 16 per batch in the
    8-shard rehearsal of config 5 (profiles/r05/z_fanout_ABC.jsonl against g_pace_gate_running_workgroups_only.txt). Same answers
    either way; this test is about the time."""
    import time
    import torch
    dev = torch.device("cuda", 0)
    dims, n, nq, shards = 768, 400_000, 1024, 4
    g = torch.Generator(device=dev).manual_seed(3)
    rows = torch.nn.functional.normalize(torch.randn((n, dims), device=dev, generator=g), dim=1).contiguous()
    one = wax.HIPVectorEngine(dimensions=dims)
    many = wax.HIPVectorEngine(dimensions=dims, devices=[0] * shards)
    ids = np.arange(n, dtype=np.uint64)
    for eng in (one, many):
        eng.reserve(n)
        eng.addBatchDevice(ids, rows)
    assert [many.shardInfo(i)[2] for i in range(shards)] == [100_032] * 3 + [n - 3 * 100_032]
    dq = torch.nn.functional.normalize(torch.randn((nq, dims), device=dev, generator=g), dim=1).contiguous()
    st = torch.cuda.current_stream(dev).cuda_stream
    outs = {}

    def batch_ms(eng, tag):
        out = torch.empty((nq, 10, 2), dtype=torch.int64, device=dev)
        for _ in range(2):
            eng.searchBatchHitsDevice(dq.data_ptr(), nq, 10, out.data_ptr(), 10, st)      # mirrors, warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.searchBatchHitsDevice(dq.data_ptr(), nq, 10, out.data_ptr(), 10, st)
        torch.cuda.synchronize()
        outs[tag] = out.cpu().numpy()
        return (time.perf_counter() - t0) / 5 * 1e3
    a, b = batch_ms(one, "one"), batch_ms(many, "many")
    print(f"\n[pace gate] one engine {a:.3f} ms per batch, four shards on one GPU {b:.3f} ms")
    assert np.array_equal(outs["one"], outs["many"])
    assert b < 3.0 * a + 1.0, (a, b)                            # (1.1 - 1.4 x measured; the round-4 gate: > 8 x)
    one.close(), many.close()
