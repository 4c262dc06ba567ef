import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import wax_amd as wax, oracle
dev = torch.device("cuda", 0)
for n, dims, nq, k, base in [(53334, 384, 300, 200, 0), (53334, 384, 300, 200, 53334), (2500, 384, 200, 200, 2500)]:
    rows = torch.nn.functional.normalize(torch.randn((n, dims), device=dev, generator=torch.Generator(device=dev).manual_seed(n + dims)), dim=1).contiguous()
    eng = wax.HIPVectorEngine(dimensions=dims)
    eng.reserve(n); eng.addBatchDevice(np.arange(n, dtype=np.uint64) * 3 + 11, rows)
    if base: eng.setRowBase(base)
    queries = oracle.gaussian_unit_queries(nq, dims, seed=5)
    dq = torch.from_numpy(queries).to(dev)
    out = torch.full((nq, k, 2), 5, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for rep in range(3):
        eng.searchBatchHitsDevice(dq.data_ptr(), nq, k, out.data_ptr(), k, st)
        torch.cuda.synchronize()
        hits = out.cpu().numpy()
        bad = 0
        for qi in range(nq):
            b_ids, b_scores = wax.HIPVectorEngine.hitsToResults(wax.VectorMetric.cosine, hits[qi])
            s_ids, s_scores = eng.searchArrays(queries[qi], k)
            if not (np.array_equal(b_ids, s_ids) and np.array_equal(b_scores, s_scores)):
                if bad < 3:
                    print("  diff q", qi, "batch n", len(b_ids), "single n", len(s_ids), "zeros in batch keys", int((hits[qi][:, 0] == 0).sum()), "first ids", b_ids[:4], s_ids[:4])
                bad += 1
        print(n, dims, nq, k, "base", base, "rep", rep, "bad", bad)
    # single queries against the oracle
    corpus = rows.cpu().numpy()
    ids = np.arange(n, dtype=np.uint64) * 3 + 11
    bad = 0
    for qi in range(0, 40):
        s_ids, s_scores = eng.searchArrays(queries[qi], k)
        e_ids, e_scores, _, _ = oracle.search(0, corpus, ids, queries[qi], k)
        if not np.array_equal(s_ids, e_ids): bad += 1
    print("single vs oracle bad", bad)
    eng.close()
