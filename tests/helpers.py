"""Shared test infrastructure: an oracle-backed engine with the reference's store semantics,
the reference-case runner, and the parity comparison rule of SURVEY.md §8c."""
from __future__ import annotations

import json
import os

import numpy as np

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
METRIC_NAMES = {"cosine": 0, "dot": 1, "l2": 2}


def load_golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


class OracleDimensionMismatch(Exception):
    pass


class OracleEngine:
    """CPU stand-in with MetalVectorEngine's store semantics (a9): upsert by frameId
    (MetalVectorEngine.swift:330-402), order-preserving remove (:423-444), MV2V serialize
    (:682-714) — search through oracle.search. TEST INFRASTRUCTURE ONLY."""

    def __init__(self, metric=0, dimensions=0, mode=oracle.MODE_TRUTH_F64):
        self.metric, self.dimensions, self.mode = int(metric), int(dimensions), mode
        self.rows, self.ids = [], []
        self.allocs, self.reuse = 1, 0

    @property
    def count(self):
        return len(self.ids)

    def add(self, frameId, vector):
        self.addBatch([frameId], [vector])

    def addBatch(self, frameIds, vectors):
        if len(frameIds) != len(vectors):
            raise ValueError("addBatch: frameIds.count != vectors.count")
        for v in vectors:
            if len(v) != self.dimensions:
                raise OracleDimensionMismatch(f"vector dimension mismatch: expected {self.dimensions}, got {len(v)}")
        for fid, v in zip(frameIds, vectors):
            v = np.asarray(v, dtype=np.float32)
            if fid in self.ids:
                self.rows[self.ids.index(fid)] = v  # firstIndex(of:) (:385)
            else:
                self.ids.append(int(fid))
                self.rows.append(v)

    def remove(self, frameId):
        if frameId in self.ids:
            i = self.ids.index(frameId)
            del self.ids[i]
            del self.rows[i]

    def matrix(self):
        if not self.rows:
            return np.zeros((0, self.dimensions), dtype=np.float32)
        return np.stack(self.rows).astype(np.float32)

    def searchFilteredHits(self, vector, topK, allow):
        """The reference's vector lane under FrameFilter(frameIds:): candidateLimit results from the engine
        (UnifiedSearch.swift:1195-1200), post-filtered by the allow-list (:1250), first topK kept."""
        if topK <= 0:
            return []
        limit = max(topK, min(topK * 3, 1000))
        allowed = set(int(a) for a in allow)
        return [h for h in self.search(vector, limit) if h[0] in allowed][:topK]

    def search(self, vector, topK):
        if self.count == 0:
            return []
        self.reuse += 1
        try:
            ids, scores, _, _ = oracle.search(self.metric, self.matrix(), np.array(self.ids, dtype=np.uint64),
                                              np.asarray(vector, dtype=np.float32), topK, self.mode)
        except oracle.DimensionMismatch as e:
            raise OracleDimensionMismatch(str(e))
        return [(int(i), float(s)) for i, s in zip(ids, scores)]

    def serialize(self):
        return oracle.mv2v_serialize(self.metric, self.matrix().reshape(self.count, self.dimensions),
                                     np.array(self.ids, dtype=np.uint64))

    def deserialize(self, data):
        rc, vec, ids = oracle.mv2v_parse(data, self.metric, self.dimensions)
        if rc not in (0,):
            raise ValueError(f"bad segment (check {rc})")
        self.rows = [r for r in vec]
        self.ids = [int(i) for i in ids]

    def debugBufferPoolStats(self):
        class S:
            pass
        s = S()
        s.transientAllocations, s.reuseCount = self.allocs, self.reuse
        return s


def run_reference_case(case, make_engine, normalize):
    """Replays one transcribed reference test against an engine factory and checks the
    reference's own assertions. `normalize(vec)` is the caller-side VectorMath.normalizeL2."""
    metric = METRIC_NAMES[case["metric"]]
    eng = make_engine(metric, case["dimensions"])
    saved = {}
    for op in case["ops"]:
        kind = op["op"]
        exp = op.get("expect", {})
        if kind == "add":
            eng.add(op["frameId"], op["vector"])
        elif kind == "addBatch":
            eng.addBatch(op["frameIds"], op["vectors"])
        elif kind == "remove":
            eng.remove(op["frameId"])
        elif kind == "serialize_deserialize_into_new_engine":
            blob = eng.serialize()
            if exp.get("blobNonEmpty"):
                assert len(blob) > 0
            eng2 = make_engine(metric, case["dimensions"])
            eng2.deserialize(blob)
            eng = eng2
        elif kind == "search":
            q = op["vector"]
            if op.get("normalizeQueryLikeCaller"):
                q = list(normalize(q))
            hits = eng.search(q, op["topK"])
            ids = [h[0] for h in hits]
            if exp.get("nonEmpty"):
                assert hits, case["name"]
            for c in exp.get("contains", []):
                assert c in ids, (case["name"], ids)
            for c in exp.get("notContains", []):
                assert c not in ids, (case["name"], ids)
            if "first" in exp:
                assert hits and ids[0] == exp["first"], (case["name"], hits)
            scores = [h[1] for h in hits]
            assert scores == sorted(scores, reverse=True), (case["name"], scores)
            if "save" in op:
                saved[op["save"]] = hits
        elif kind == "search_filtered":
            hits = eng.searchFilteredHits(op["vector"], op["topK"], op["allow"])
            ids = [h[0] for h in hits]
            if "idSetEquals" in exp:
                assert set(ids) == set(exp["idSetEquals"]), (case["name"], ids)
            scores = [h[1] for h in hits]
            assert scores == sorted(scores, reverse=True), (case["name"], scores)
        elif kind == "compare_first_scores":
            a, b = saved[op["a"]][0][1], saved[op["b"]][0][1]
            assert abs(a - b) < op["tolerance"], (a, b)
        elif kind == "pool_stats":
            st = eng.debugBufferPoolStats()
            if "transientAllocationsEqual" in exp:
                assert st.transientAllocations == saved[exp["transientAllocationsEqual"]].transientAllocations
            if "reuseCountAtLeast" in exp:
                assert st.reuseCount >= saved[exp["reuseCountAtLeast"]].reuseCount
            if "save" in op:
                saved[op["save"]] = st
        else:
            raise AssertionError(f"unknown op {kind}")
    return eng


SCORE_TOL = 1e-5   # north_star: "match the reference CPU path's returned indices/scores within 1e-5"
TIE_TOL = 2e-5     # ids may permute only inside groups whose oracle scores differ < 2e-5


def assert_parity(got_ids, got_scores, exp_ids, exp_scores, all_exp_scores=None, ctx=""):
    """The parity rule of SURVEY.md §8c, as the GPU tests apply it.

    1. Scores: |got - expected| <= SCORE_TOL (1e-5, the north star's tolerance) position by position — ALWAYS, whatever
       the ids are. (The engine's f32 distances and the oracle's f64 truth differ by a few 1e-7 on unit-norm data.)
    2. Ids: identical lists pass. Otherwise the oracle ranking is cut into NEAR-TIE GROUPS: maximal runs of neighbouring
       oracle scores closer than TIE_TOL (2e-5 = twice the score tolerance: two rows that close can legitimately swap
       between an f32 and an f64 evaluation). Inside a group the ids may appear in any order; as SETS every group must
       match exactly.
    3. The boundary group. `all_exp_scores` carries the oracle's scores for k + margin results, so a run that starts
       inside the top k and continues PAST rank k is seen as such. For that group the engine may hold a different
       member of the run than the oracle does (which of several rows within 2e-5 of each other makes the cut is not
       decidable at the stated tolerance), so set equality cannot be demanded: the rule is that the engine returns as many
       DISTINCT ids in those positions as the oracle, and rule 1 still pins each of them to the oracle's score at that
       position within 1e-5 — an id from outside the tied run would fail rule 1. Without `all_exp_scores` the run is
       only followed to rank k and the boundary group is held to set equality like any other (the strict form; the
       full-size parity tests pass the margin, the exact-tie tests — duplicates, constant rows — do not need one because
       the engine's (distance asc, row asc) order is the oracle's parity-mode order there)."""
    got_ids = [int(x) for x in got_ids]
    exp_ids = [int(x) for x in exp_ids]
    assert len(got_ids) == len(exp_ids), f"{ctx}: count {len(got_ids)} != {len(exp_ids)}"
    gs = np.asarray(got_scores, dtype=np.float64)
    es = np.asarray(exp_scores, dtype=np.float64)
    assert np.all(np.abs(gs - es) <= SCORE_TOL), f"{ctx}: max score err {np.max(np.abs(gs - es)) if len(gs) else 0}"
    if got_ids == exp_ids:
        return
    # build near-tie groups over the oracle ranking
    ext = np.asarray(all_exp_scores if all_exp_scores is not None else exp_scores, dtype=np.float64)
    n = len(exp_ids)
    i = 0
    while i < n:
        j = i
        while j + 1 < len(ext) and abs(ext[j] - ext[j + 1]) < TIE_TOL:
            j += 1
        hi = min(j, n - 1)
        grp_exp = set(exp_ids[i:hi + 1])
        grp_got = set(got_ids[i:hi + 1])
        if j >= n:  # group straddles the k boundary: got may contain boundary-tied ids outside exp
            assert len(grp_got) == len(grp_exp), f"{ctx}: duplicate ids at boundary"
        else:
            assert grp_got == grp_exp, f"{ctx}: ids differ outside a tie group at rank {i}: {got_ids} vs {exp_ids}"
        i = hi + 1


def bf16_adversarial_unit_vector(dims):
    """A unit vector (f32) whose bf16 rounding moves its self-similarity by ~2^-7: every non-zero element is (1 + 0.99 * 2^-8) * 2^-e,
    just below a rounding midpoint at the bottom of its binade (so rounding takes 2^-8 of it away), with the exponents chosen
    greedily so that the squares sum to 1 — the normalisation the engine applies then changes nothing. ~20 non-zeros, rest 0."""
    import numpy as np
    a = 2.0 ** -8 * 0.99
    rem, exps = (1 + a) ** -2, []
    for e in range(1, 14):
        c = int(rem // 4.0 ** -e)
        c = min(c, 3 if e < 6 else 10 ** 9)
        exps += [e] * c
        rem -= c * 4.0 ** -e
    x = np.zeros(dims, dtype=np.float64)
    x[:len(exps)] = [(1 + a) * 2.0 ** -e for e in exps]
    return (x / np.sqrt(np.sum(x * x))).astype(np.float32)
