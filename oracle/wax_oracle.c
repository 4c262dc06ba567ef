/*
 * wax_oracle.c — CPU restatement of the reference's vector scan + top-k path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under wax_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker / reported baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the christopherkarani/Wax checkout). No reference source is copied: the
 * reference is Swift + Metal Shading Language, this is a from-scratch C
 * restatement of the arithmetic and of the selection order.
 *
 * Pinning status:
 *   - PINNED, bit for bit, against OUTPUTS OF THE REFERENCE ITSELF on its GPU path: the reference's own Metal
 *     compute shaders (CosineDistance.metal, TopKReduction.metal) are compiled unchanged as C++ from where they lie
 *     under /root/reference (oracle/ref_metal/ -> oracle/_ref/libwaxref_metal.so; a stand-in for <metal_stdlib>,
 *     threadgroup threads as fibers, IEEE binary32 semantics) and executed on the CPU. wax_oracle_cosine_distances_metal
 *     returns the same 32 bits per row as cosineDistanceKernelSIMD4 / SIMD8; the selection equals topKReduceDistances /
 *     topKReduceEntries under the engine's dispatch loop. tests/golden/metal_shader_vectors.json carries those outputs
 *     (generator: oracle/gen_metal_golden.py) to machines without the reference checkout; tests/test_reference_shaders.py.
 *   - PINNED by the reference's own tests (tests/golden/reference_cases.json,
 *     transcribed with file:line): rank / membership on the 2-d and 4-d toy
 *     corpora, the scaled-query tolerance (1e-3), upsert-by-id, remove,
 *     the MV2V header constants and the topK clamp.
 *   - PARITY UNPINNED at the numeric USearch boundary only: the reference's CPU
 *     engine's arithmetic lives in USearch 2.23.0 (Package.resolved rev
 *     7306bb446be5f0f0c529ec8acdc57361cef8a8a7), which is not vendored in the
 *     checkout and cannot be built here (no Swift, no network), and no
 *     reference test pins a numeric score. USearch's published metric
 *     definitions (cos = 1 - ab/(|a||b|), ip = 1 - ab, l2sq = sum (a-b)^2)
 *     are restated in wax_oracle_distances_f64 from its documentation; the f64 truth sits within 2e-6 of the
 *     reference's Metal kernels on unit queries (same test file), far inside the 1e-5 tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WAX_ORACLE_API __attribute__((visibility("default")))

enum { METRIC_COSINE = 0, METRIC_DOT = 1, METRIC_L2 = 2 };

/* ------------------------------------------------------------------------ */
/* a8: clampTopK — MetalVectorEngine.swift:842-846 (twin USearchVectorEngine.swift:331-335) */
WAX_ORACLE_API int32_t wax_oracle_clamp_topk(int64_t top_k) {
    if (top_k < 1) return 1;
    if (top_k > 10000) return 10000;
    return (int32_t)top_k;
}

/* a6: VectorMetric.score(fromDistance:) — VectorMetric.swift:32-43 */
WAX_ORACLE_API float wax_oracle_score_from_distance(int metric, float d) {
    if (!isfinite(d)) return 0.0f;
    if (metric == METRIC_COSINE) return 1.0f - d;
    return -d;
}

/* a11: VectorMath.magnitude / normalizeL2 / isNormalizedL2 — VectorMath.swift:14-33, 120-135.
 * vDSP_svesq's internal summation order is not specified; a plain f32
 * left-to-right sum is used (callers only test |len-1| <= 1e-3). */
WAX_ORACLE_API float wax_oracle_magnitude(const float* v, uint32_t n) {
    if (n == 0) return 0.0f;
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrtf(s);
}

WAX_ORACLE_API void wax_oracle_normalize_l2(const float* v, uint32_t n, float* out) {
    float mag = wax_oracle_magnitude(v, n);
    if (n == 0 || !(mag > 0.0f)) { /* VectorMath.swift:15, 23 — returned unchanged */
        memmove(out, v, (size_t)n * sizeof(float));
        return;
    }
    float inv = 1.0f / mag; /* VectorMath.swift:26 */
    for (uint32_t i = 0; i < n; ++i) out[i] = v[i] * inv;
}

WAX_ORACLE_API int wax_oracle_is_normalized_l2(const float* v, uint32_t n, float tolerance) {
    if (n == 0) return 0; /* VectorMath.swift:132 */
    float len = wax_oracle_magnitude(v, n);
    return fabsf(len - 1.0f) <= tolerance;
}

/* ------------------------------------------------------------------------ */
/* a3, "metal-faithful": cosineDistanceKernelSIMD8 (CosineDistance.metal:233-328,
 * used when D >= 384, MetalVectorEngine.swift:24,185) and cosineDistanceKernelSIMD4
 * (:152-229, D < 384). One row per GPU thread; per-component f32 FMA into
 * float4 accumulators (two independent pairs in SIMD8: even / odd float4),
 * merge a+b, horizontal x+y+z+w, scalar fma tail for D%4, then
 * dist = 1 - dot/sqrt(m) with sqrt(m) <= 1e-6 => similarity 0. ||q|| is NOT
 * divided out (the kernels assume a unit query, :142, :224, :323). */
static float metal_row_distance(const float* v, const float* q, uint32_t d, int simd8) {
    uint32_t dims4 = d >> 2, rem = d & 3;
    float dota[4] = {0, 0, 0, 0}, dotb[4] = {0, 0, 0, 0};
    float maga[4] = {0, 0, 0, 0}, magb[4] = {0, 0, 0, 0};
    if (simd8) {
        uint32_t dims8 = d >> 3;
        for (uint32_t i = 0; i < dims8; ++i) {
            const float* v0 = v + 8 * i;
            const float* q0 = q + 8 * i;
            for (int c = 0; c < 4; ++c) {
                dota[c] = fmaf(q0[c], v0[c], dota[c]);
                dotb[c] = fmaf(q0[4 + c], v0[4 + c], dotb[c]);
                maga[c] = fmaf(v0[c], v0[c], maga[c]);
                magb[c] = fmaf(v0[4 + c], v0[4 + c], magb[c]);
            }
        }
        if (dims4 & 1) { /* remaining float4, CosineDistance.metal:292-298 */
            const float* v0 = v + 8 * dims8;
            const float* q0 = q + 8 * dims8;
            for (int c = 0; c < 4; ++c) {
                dota[c] = fmaf(q0[c], v0[c], dota[c]);
                maga[c] = fmaf(v0[c], v0[c], maga[c]);
            }
        }
        for (int c = 0; c < 4; ++c) { /* merge dual accumulators, :301-302 */
            dota[c] = dota[c] + dotb[c];
            maga[c] = maga[c] + magb[c];
        }
    } else {
        for (uint32_t i = 0; i < dims4; ++i) { /* :196-203 */
            const float* v0 = v + 4 * i;
            const float* q0 = q + 4 * i;
            for (int c = 0; c < 4; ++c) {
                dota[c] = fmaf(q0[c], v0[c], dota[c]);
                maga[c] = fmaf(v0[c], v0[c], maga[c]);
            }
        }
    }
    float dot = ((dota[0] + dota[1]) + dota[2]) + dota[3]; /* :206 / :304 */
    float mag = ((maga[0] + maga[1]) + maga[2]) + maga[3];
    for (uint32_t r = 0; r < rem; ++r) { /* scalar tail :210-221 / :308-319 */
        float vv = v[4 * dims4 + r], qq = q[4 * dims4 + r];
        dot = fmaf(qq, vv, dot);
        mag = fmaf(vv, vv, mag);
    }
    float magnitude = sqrtf(mag);
    float sim = (magnitude > 1e-6f) ? dot / magnitude : 0.0f; /* :225 / :323 */
    return 1.0f - sim;
}

WAX_ORACLE_API void wax_oracle_cosine_distances_metal(const float* vectors, const float* query,
                                                      uint64_t n, uint32_t d, float* out) {
    int simd8 = d >= 384; /* MetalVectorEngine.swift:24, 185 */
    for (uint64_t i = 0; i < n; ++i) out[i] = metal_row_distance(vectors + i * (uint64_t)d, query, d, simd8);
}

/* ------------------------------------------------------------------------ */
/* Parity truth (SURVEY.md §8c "oracle definition adopted"): exact flat scan,
 * f32 storage, f64 accumulation, rounded once to f32.
 *   cosine: full-cosine form of CosineDistance.metal:25-67 == USearch `cos`
 *           (USearchVectorEngine.swift:60-65 with VectorMetric.swift:21-30):
 *           d = 1 - q.v / (|q||v|); zero-norm row (sqrt(m) <= 1e-6, Metal's
 *           rule :225) or zero-norm query => similarity 0 => d = 1.
 *   dot:    USearch `ip`   d = 1 - q.v
 *   l2:     USearch `l2sq` d = sum (q - v)^2 */
WAX_ORACLE_API void wax_oracle_distances_f64(int metric, const float* vectors, const float* query,
                                             uint64_t n, uint32_t d, float* out) {
    double qq = 0.0;
    for (uint32_t j = 0; j < d; ++j) qq += (double)query[j] * (double)query[j];
    double qn = sqrt(qq);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float* v = vectors + (uint64_t)i * d;
        double dot = 0.0, m = 0.0, l2 = 0.0;
        for (uint32_t j = 0; j < d; ++j) {
            double a = (double)query[j], b = (double)v[j];
            dot += a * b;
            m += b * b;
            l2 += (a - b) * (a - b);
        }
        double dist;
        if (metric == METRIC_COSINE) {
            double vn = sqrt(m);
            double sim = (vn > 1e-6 && qn > 1e-6) ? dot / (vn * qn) : 0.0;
            dist = 1.0 - sim;
        } else if (metric == METRIC_DOT) {
            dist = 1.0 - dot;
        } else {
            dist = l2;
        }
        out[i] = (float)dist;
    }
}

/* ------------------------------------------------------------------------ */
/* a5: MetalVectorEngine.topK(distances:count:k:) — MetalVectorEngine.swift:630-680.
 * Size-k max-heap of (distance, index): seed with the first min(k,count)
 * entries, heapify from count/2 down to 0, then for each later entry
 * `value >= heap[0] => skip` (so among equal boundary values the EARLIER index
 * stays), else replace the root and sift down. The reference then sorts
 * ascending by distance only (:678; Swift's sort gives no order for equal
 * distances). The oracle sorts by (distance asc, index asc): the
 * deterministic completion SURVEY.md §8c adopts. */
typedef struct { float d; int64_t i; } heap_ent;

static void sift_down(heap_ent* h, int64_t start, int64_t end) { /* :635-647 */
    int64_t root = start;
    for (;;) {
        int64_t child = root * 2 + 1;
        if (child > end) break;
        int64_t sw = root;
        if (h[sw].d < h[child].d) sw = child;
        if (child + 1 <= end && h[sw].d < h[child + 1].d) sw = child + 1;
        if (sw == root) return;
        heap_ent t = h[root]; h[root] = h[sw]; h[sw] = t;
        root = sw;
    }
}

static int cmp_ent(const void* a, const void* b) {
    const heap_ent* x = (const heap_ent*)a;
    const heap_ent* y = (const heap_ent*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

WAX_ORACLE_API int64_t wax_oracle_topk_heap(const float* distances, int64_t count, int64_t k,
                                            int64_t* out_idx, float* out_dist) {
    if (k <= 0 || count <= 0) return 0; /* :631 */
    int64_t initial = k < count ? k : count; /* :659 */
    heap_ent* h = (heap_ent*)malloc((size_t)initial * sizeof(heap_ent));
    for (int64_t i = 0; i < initial; ++i) { h[i].d = distances[i]; h[i].i = i; }
    for (int64_t i = initial / 2; i >= 0; --i) sift_down(h, i, initial - 1); /* :664-666 */
    for (int64_t i = initial; i < count; ++i) { /* :668-675 */
        float v = distances[i];
        if (v >= h[0].d) continue;
        h[0].d = v; h[0].i = i;
        sift_down(h, 0, initial - 1);
    }
    qsort(h, (size_t)initial, sizeof(heap_ent), cmp_ent); /* :678 + tie rule */
    for (int64_t i = 0; i < initial; ++i) { out_idx[i] = h[i].i; out_dist[i] = h[i].d; }
    free(h);
    return initial;
}

/* Selection under the ADOPTED total order (distance asc, index asc) — what the
 * parity tests compare the GPU against. Same heap as above but comparing
 * (distance, index) lexicographically, so the result is exactly the k smallest
 * pairs; it equals wax_oracle_topk_heap whenever no two candidates tie on
 * distance at the k boundary (the reference's own tie behaviour there depends
 * on heap layout: e.g. [0.5,0.1,0.5,0.5,0.1], k=3 keeps index 2 and evicts
 * index 0 — see tests/test_oracle.py). */
static int ent_less(const heap_ent* a, const heap_ent* b) {
    return a->d < b->d || (a->d == b->d && a->i < b->i);
}

static void sift_down_total(heap_ent* h, int64_t start, int64_t end) {
    int64_t root = start;
    for (;;) {
        int64_t child = root * 2 + 1;
        if (child > end) break;
        int64_t sw = root;
        if (ent_less(&h[sw], &h[child])) sw = child;
        if (child + 1 <= end && ent_less(&h[sw], &h[child + 1])) sw = child + 1;
        if (sw == root) return;
        heap_ent t = h[root]; h[root] = h[sw]; h[sw] = t;
        root = sw;
    }
}

WAX_ORACLE_API int64_t wax_oracle_topk_total(const float* distances, int64_t count, int64_t k,
                                             int64_t* out_idx, float* out_dist) {
    if (k <= 0 || count <= 0) return 0;
    int64_t initial = k < count ? k : count;
    heap_ent* h = (heap_ent*)malloc((size_t)initial * sizeof(heap_ent));
    for (int64_t i = 0; i < initial; ++i) { h[i].d = distances[i]; h[i].i = i; }
    for (int64_t i = initial / 2; i >= 0; --i) sift_down_total(h, i, initial - 1);
    for (int64_t i = initial; i < count; ++i) {
        heap_ent e; e.d = distances[i]; e.i = i;
        if (!ent_less(&e, &h[0])) continue;
        h[0] = e;
        sift_down_total(h, 0, initial - 1);
    }
    qsort(h, (size_t)initial, sizeof(heap_ent), cmp_ent);
    for (int64_t i = 0; i < initial; ++i) { out_idx[i] = h[i].i; out_dist[i] = h[i].d; }
    free(h);
    return initial;
}

/* Brute-force cross-check of the above: full sort by (distance, index). */
WAX_ORACLE_API int64_t wax_oracle_topk_sort(const float* distances, int64_t count, int64_t k,
                                            int64_t* out_idx, float* out_dist) {
    if (k <= 0 || count <= 0) return 0;
    heap_ent* h = (heap_ent*)malloc((size_t)count * sizeof(heap_ent));
    for (int64_t i = 0; i < count; ++i) { h[i].d = distances[i]; h[i].i = i; }
    qsort(h, (size_t)count, sizeof(heap_ent), cmp_ent);
    int64_t m = k < count ? k : count;
    for (int64_t i = 0; i < m; ++i) { out_idx[i] = h[i].i; out_dist[i] = h[i].d; }
    free(h);
    return m;
}

/* ------------------------------------------------------------------------ */
/* a1/a2 composed: MetalVectorEngine.search (MetalVectorEngine.swift:446-627),
 * CPU-selection branch (:614-625): distances -> topK -> frameIds[idx], score.
 * mode 0 = parity truth (f64 accumulate, true cosine), total-order selection
 * mode 1 = metal-faithful f32 (unit-query assumption, cosine only) + the
 *          reference's own heap (its boundary-tie behaviour included)
 * returns number of results, or -1 on dimension mismatch (:830-833). */
WAX_ORACLE_API int64_t wax_oracle_search(int metric, int mode, const float* vectors, const uint64_t* frame_ids,
                                         uint64_t n, uint32_t d, const float* query, uint32_t query_dims,
                                         int64_t top_k, uint64_t* out_ids, float* out_scores,
                                         float* out_distances, int64_t* out_rows) {
    if (n == 0) return 0; /* :448 */
    if (query_dims != d) return -1; /* :449, 830-833 */
    int32_t limit = wax_oracle_clamp_topk(top_k); /* :450 */
    float* dist = (float*)malloc((size_t)n * sizeof(float));
    if (mode == 1 && metric == METRIC_COSINE) wax_oracle_cosine_distances_metal(vectors, query, n, d, dist);
    else wax_oracle_distances_f64(metric, vectors, query, n, d, dist);
    int64_t kk = (int64_t)limit < (int64_t)n ? limit : (int64_t)n;
    int64_t* idx = (int64_t*)malloc((size_t)kk * sizeof(int64_t));
    float* dd = (float*)malloc((size_t)kk * sizeof(float));
    int64_t m = (mode == 1) ? wax_oracle_topk_heap(dist, (int64_t)n, limit, idx, dd) /* :615 */
                            : wax_oracle_topk_total(dist, (int64_t)n, limit, idx, dd);
    int64_t outn = 0;
    for (int64_t i = 0; i < m; ++i) { /* :620-623; non-finite dropped as on the GPU branch :597 */
        if (!isfinite(dd[i])) continue;
        out_ids[outn] = frame_ids ? frame_ids[idx[i]] : (uint64_t)idx[i];
        out_scores[outn] = wax_oracle_score_from_distance(metric, dd[i]);
        if (out_distances) out_distances[outn] = dd[i];
        if (out_rows) out_rows[outn] = idx[i];
        ++outn;
    }
    free(dist); free(idx); free(dd);
    return outn;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline (bench.py cpu_baseline leg): the same scan + heap selection
 * as above (a3 + a5 arithmetic in f32), parallelised the way SURVEY.md §8d
 * prescribes: static row partition, one heap per thread, final merge under
 * the (distance, index) order. f32 multi-accumulator dot products so gcc can
 * vectorise; results agree with the f64 truth to ~1e-6 (checked in tests). */
__attribute__((target_clones("arch=haswell", "default")))
static void row_dot_f32(const float* v, const float* q, uint32_t d, float* dot_out, float* m_out, float* l2_out) {
    float dot[8] = {0}, mm[8] = {0}, ll[8] = {0};
    uint32_t d8 = d & ~7u;
    for (uint32_t j = 0; j < d8; j += 8)
        for (int c = 0; c < 8; ++c) {
            float a = q[j + c], b = v[j + c], e = a - b;
            dot[c] += a * b; mm[c] += b * b; ll[c] += e * e;
        }
    float sd = 0, sm = 0, sl = 0;
    for (int c = 0; c < 8; ++c) { sd += dot[c]; sm += mm[c]; sl += ll[c]; }
    for (uint32_t j = d8; j < d; ++j) { float a = q[j], b = v[j], e = a - b; sd += a * b; sm += b * b; sl += e * e; }
    *dot_out = sd; *m_out = sm; *l2_out = sl;
}

static float row_distance_f32(int metric, const float* v, const float* q, uint32_t d, float qn) {
    float dot, m, l2;
    row_dot_f32(v, q, d, &dot, &m, &l2);
    if (metric == METRIC_COSINE) {
        float vn = sqrtf(m);
        float sim = (vn > 1e-6f && qn > 1e-6f) ? dot / (vn * qn) : 0.0f;
        return 1.0f - sim;
    }
    if (metric == METRIC_DOT) return 1.0f - dot;
    return l2;
}

WAX_ORACLE_API int wax_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

WAX_ORACLE_API int64_t wax_oracle_scan_topk_mt(int metric, const float* vectors, uint64_t n, uint32_t d,
                                               const float* query, int64_t top_k, int threads,
                                               int64_t* out_idx, float* out_dist) {
    if (n == 0) return 0;
    int32_t limit = wax_oracle_clamp_topk(top_k);
    int64_t kk = (int64_t)limit < (int64_t)n ? limit : (int64_t)n;
    if (threads < 1) threads = 1;
    float qn = wax_oracle_magnitude(query, d);
    heap_ent* all = (heap_ent*)malloc((size_t)threads * (size_t)kk * sizeof(heap_ent));
    int64_t* counts = (int64_t*)calloc((size_t)threads, sizeof(int64_t));
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
        heap_ent* h = all + (size_t)t * (size_t)kk;
        int64_t cnt = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            float dist = row_distance_f32(metric, vectors + i * (uint64_t)d, query, d, qn);
            if (cnt < kk) {
                h[cnt].d = dist; h[cnt].i = (int64_t)i; ++cnt;
                if (cnt == kk) for (int64_t s = kk / 2; s >= 0; --s) sift_down(h, s, kk - 1);
            } else {
                if (dist >= h[0].d) continue; /* MetalVectorEngine.swift:671 */
                h[0].d = dist; h[0].i = (int64_t)i;
                sift_down(h, 0, kk - 1);
            }
        }
        counts[t] = cnt;
    }
    int64_t total = 0;
    for (int t = 0; t < threads; ++t) {
        if (counts[t] && total != (int64_t)t * kk) memmove(all + total, all + (size_t)t * (size_t)kk, (size_t)counts[t] * sizeof(heap_ent));
        total += counts[t];
    }
    qsort(all, (size_t)total, sizeof(heap_ent), cmp_ent);
    int64_t m = kk < total ? kk : total;
    for (int64_t i = 0; i < m; ++i) { out_idx[i] = all[i].i; out_dist[i] = all[i].d; }
    free(all); free(counts);
    return m;
}

/* ------------------------------------------------------------------------ */
/* Batched parity truth: the arithmetic of wax_oracle_distances_f64 (same f64 accumulation order per
 * (row, query) pair, so every distance is bit-identical to the single-query truth) + the total-order
 * selection of wax_oracle_topk_total, for nq queries in ONE pass over the rows (a row stays in L1 while
 * all queries visit it). This is what lets the GPU parity tests compare EVERY answer of a 256- / 1024-query
 * batch at the BASELINE sizes (1M x 384, 1.25M x 768) with the oracle in seconds. Rows are partitioned
 * statically over the threads (ascending inside a thread), one heap per (thread, query), merged under
 * (distance asc, row asc). out_rows / out_dist are [nq][kk] with kk = min(clamp(top_k), n) (returned);
 * out_counts[q] <= kk after non-finite distances are dropped (MetalVectorEngine.swift:597). */
WAX_ORACLE_API int64_t wax_oracle_search_batch(int metric, const float* vectors, uint64_t n, uint32_t d,
                                               const float* queries, uint32_t nq, int64_t top_k,
                                               int64_t* out_rows, float* out_dist, int64_t* out_counts) {
    if (n == 0 || nq == 0) return 0;
    int32_t limit = wax_oracle_clamp_topk(top_k);
    int64_t kk = (int64_t)limit < (int64_t)n ? limit : (int64_t)n;
    int threads = wax_oracle_max_threads();
    if ((uint64_t)threads > n) threads = (int)n;
    double* qn = (double*)malloc((size_t)nq * sizeof(double));
    for (uint32_t q = 0; q < nq; ++q) {
        const float* qv = queries + (uint64_t)q * d;
        double qq = 0.0;
        for (uint32_t j = 0; j < d; ++j) qq += (double)qv[j] * (double)qv[j];
        qn[q] = sqrt(qq);
    }
    heap_ent* all = (heap_ent*)malloc((size_t)threads * (size_t)nq * (size_t)kk * sizeof(heap_ent));
    int64_t* cnts = (int64_t*)calloc((size_t)threads * (size_t)nq, sizeof(int64_t));
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
        for (uint64_t i = lo; i < hi; ++i) {
            const float* v = vectors + i * (uint64_t)d;
            double m = 0.0;
            if (metric == METRIC_COSINE)
                for (uint32_t j = 0; j < d; ++j) { double b = (double)v[j]; m += b * b; }
            double vn = sqrt(m);
            for (uint32_t q = 0; q < nq; ++q) {
                const float* qv = queries + (uint64_t)q * d;
                double acc = 0.0;
                if (metric == METRIC_L2) {
                    for (uint32_t j = 0; j < d; ++j) { double a = (double)qv[j], b = (double)v[j]; acc += (a - b) * (a - b); }
                } else {
                    for (uint32_t j = 0; j < d; ++j) { double a = (double)qv[j], b = (double)v[j]; acc += a * b; }
                }
                double dist;
                if (metric == METRIC_COSINE) {
                    double sim = (vn > 1e-6 && qn[q] > 1e-6) ? acc / (vn * qn[q]) : 0.0;
                    dist = 1.0 - sim;
                } else if (metric == METRIC_DOT) {
                    dist = 1.0 - acc;
                } else {
                    dist = acc;
                }
                heap_ent e; e.d = (float)dist; e.i = (int64_t)i;
                heap_ent* h = all + ((size_t)t * nq + q) * (size_t)kk;
                int64_t* c = cnts + (size_t)t * nq + q;
                if (*c < kk) {
                    h[*c] = e; ++*c;
                    if (*c == kk) for (int64_t s = kk / 2; s >= 0; --s) sift_down_total(h, s, kk - 1);
                } else if (ent_less(&e, &h[0])) {
                    h[0] = e;
                    sift_down_total(h, 0, kk - 1);
                }
            }
        }
    }
    heap_ent* merged = (heap_ent*)malloc((size_t)threads * (size_t)kk * sizeof(heap_ent));
    for (uint32_t q = 0; q < nq; ++q) {
        int64_t total = 0;
        for (int t = 0; t < threads; ++t) {
            int64_t c = cnts[(size_t)t * nq + q];
            memcpy(merged + total, all + ((size_t)t * nq + q) * (size_t)kk, (size_t)c * sizeof(heap_ent));
            total += c;
        }
        qsort(merged, (size_t)total, sizeof(heap_ent), cmp_ent);
        int64_t m = kk < total ? kk : total, outn = 0;
        for (int64_t i = 0; i < m; ++i) {
            if (!isfinite(merged[i].d)) continue;
            out_rows[(size_t)q * kk + outn] = merged[i].i;
            out_dist[(size_t)q * kk + outn] = merged[i].d;
            ++outn;
        }
        out_counts[q] = outn;
    }
    free(merged); free(all); free(cnts); free(qn);
    return kk;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline, tuned variant (bench.py cpu_baseline "variants"): the same scan + heap selection as
 * wax_oracle_scan_topk_mt, but with a METRIC-SPECIALISED inner loop (cosine accumulates dot and |v|^2
 * only, dot / l2 one sum) written with explicit fused multiply-adds on 16 independent lanes so that gcc
 * vectorises it to AVX2 / AVX-512 FMA whatever -ffp-contract says. It is a reported baseline, not a
 * parity reference: distances agree with the f64 truth to ~1e-6 (tests/test_oracle.py). */
#define WAX_LANES 16
/* Clones by ISA FEATURE, not by CPU model (an "arch=skylake-avx512" clone is never chosen on an AMD EPYC host and the
 * default clone would then run): avx512f (implies FMA), fma (AVX + FMA3: 256-bit vfmadd), baseline. fp-contract=fast
 * lets gcc fuse a*b+c where the clone has FMA and keeps mul+add where it has not (no libm fmaf call anywhere). */
__attribute__((target_clones("avx512f", "fma", "default"), optimize("fp-contract=fast")))
static float row_distance_fast(int metric, const float* v, const float* q, uint32_t d, float qn) {
    float s0[WAX_LANES] = {0}, s1[WAX_LANES] = {0};
    uint32_t dl = d - d % WAX_LANES, j = 0;
    if (metric == METRIC_COSINE) {
        for (; j < dl; j += WAX_LANES)
            for (int c = 0; c < WAX_LANES; ++c) {
                s0[c] += q[j + c] * v[j + c];
                s1[c] += v[j + c] * v[j + c];
            }
    } else if (metric == METRIC_DOT) {
        for (; j < dl; j += WAX_LANES)
            for (int c = 0; c < WAX_LANES; ++c) s0[c] += q[j + c] * v[j + c];
    } else {
        for (; j < dl; j += WAX_LANES)
            for (int c = 0; c < WAX_LANES; ++c) { float e = q[j + c] - v[j + c]; s0[c] += e * e; }
    }
    float a = 0.f, m = 0.f;
    for (int c = 0; c < WAX_LANES; ++c) { a += s0[c]; m += s1[c]; }
    for (; j < d; ++j) {
        if (metric == METRIC_L2) { float e = q[j] - v[j]; a += e * e; }
        else { a += q[j] * v[j]; m += v[j] * v[j]; }
    }
    if (metric == METRIC_COSINE) {
        float vn = sqrtf(m);
        float sim = (vn > 1e-6f && qn > 1e-6f) ? a / (vn * qn) : 0.0f;
        return 1.0f - sim;
    }
    if (metric == METRIC_DOT) return 1.0f - a;
    return a;
}

WAX_ORACLE_API int64_t wax_oracle_scan_topk_fast(int metric, const float* vectors, uint64_t n, uint32_t d,
                                                 const float* query, int64_t top_k, int threads,
                                                 int64_t* out_idx, float* out_dist) {
    if (n == 0) return 0;
    int32_t limit = wax_oracle_clamp_topk(top_k);
    int64_t kk = (int64_t)limit < (int64_t)n ? limit : (int64_t)n;
    if (threads < 1) threads = 1;
    float qn = wax_oracle_magnitude(query, d);
    heap_ent* all = (heap_ent*)malloc((size_t)threads * (size_t)kk * sizeof(heap_ent));
    int64_t* counts = (int64_t*)calloc((size_t)threads, sizeof(int64_t));
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
        heap_ent* h = all + (size_t)t * (size_t)kk;
        int64_t cnt = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            float dist = row_distance_fast(metric, vectors + i * (uint64_t)d, query, d, qn);
            if (cnt < kk) {
                h[cnt].d = dist; h[cnt].i = (int64_t)i; ++cnt;
                if (cnt == kk) for (int64_t s = kk / 2; s >= 0; --s) sift_down(h, s, kk - 1);
            } else {
                if (dist >= h[0].d) continue; /* MetalVectorEngine.swift:671 */
                h[0].d = dist; h[0].i = (int64_t)i;
                sift_down(h, 0, kk - 1);
            }
        }
        counts[t] = cnt;
    }
    int64_t total = 0;
    for (int t = 0; t < threads; ++t) {
        if (counts[t] && total != (int64_t)t * kk) memmove(all + total, all + (size_t)t * (size_t)kk, (size_t)counts[t] * sizeof(heap_ent));
        total += counts[t];
    }
    qsort(all, (size_t)total, sizeof(heap_ent), cmp_ent);
    int64_t m = kk < total ? kk : total;
    for (int64_t i = 0; i < m; ++i) { out_idx[i] = all[i].i; out_dist[i] = all[i].d; }
    free(all); free(counts);
    return m;
}

/* First-touch `bytes` at `p` with the SAME static partition over `threads` threads as the scans above
 * (row r of an [n][d] f32 array is touched by the thread that will scan it), so that on a multi-socket
 * host every thread later streams from its own NUMA node. Zero-fills. */
WAX_ORACLE_API void wax_oracle_first_touch_rows(float* p, uint64_t n, uint32_t d, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
        if (hi > lo) memset(p + lo * (uint64_t)d, 0, (size_t)(hi - lo) * d * sizeof(float));
    }
}

/* Parallel copy with the same partition (fills a first-touched sample without moving its pages). */
WAX_ORACLE_API void wax_oracle_copy_rows(float* dst, const float* src, uint64_t n, uint32_t d, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
        if (hi > lo) memcpy(dst + lo * (uint64_t)d, src + lo * (uint64_t)d, (size_t)(hi - lo) * d * sizeof(float));
    }
}

/* ------------------------------------------------------------------------ */
/* Row f7: "MV2V" vec segment, encoding 2 — MetalVectorEngine.serialize
 * (MetalVectorEngine.swift:682-714) / VectorSerializer.decodeVecSegment
 * (VectorSerializer.swift:84-157, header :175-251). Little-endian host assumed. */
WAX_ORACLE_API uint64_t wax_oracle_mv2v_size(uint64_t n, uint32_t d) {
    return 36ull + n * (uint64_t)d * 4ull + 8ull + n * 8ull;
}

WAX_ORACLE_API uint64_t wax_oracle_mv2v_serialize(int metric, const float* vectors, const uint64_t* frame_ids,
                                                  uint64_t n, uint32_t d, uint8_t* out) {
    uint8_t* p = out;
    const uint8_t magic[4] = {0x4D, 0x56, 0x32, 0x56}; /* :686 */
    memcpy(p, magic, 4); p += 4;
    uint16_t ver = 1; memcpy(p, &ver, 2); p += 2; /* :687-688 */
    *p++ = 2; /* encoding: metal / flat, :689 */
    *p++ = (uint8_t)metric; /* similarity raw value, :690 */
    memcpy(p, &d, 4); p += 4; /* :691-692 */
    memcpy(p, &n, 8); p += 8; /* :693-694 */
    uint64_t vec_bytes = n * (uint64_t)d * 4ull;
    memcpy(p, &vec_bytes, 8); p += 8; /* :697-699 */
    memset(p, 0, 8); p += 8; /* reserved, :700 */
    memcpy(p, vectors, vec_bytes); p += vec_bytes; /* :703-705 */
    uint64_t id_bytes = n * 8ull;
    memcpy(p, &id_bytes, 8); p += 8; /* :707-709 */
    memcpy(p, frame_ids, id_bytes); p += id_bytes; /* :710 */
    return (uint64_t)(p - out);
}

/* Returns 0 on success, else the 1-based index of the failed check in the
 * order MetalVectorEngine.deserialize performs them (:718-808), and
 * VectorSerializer's whole-length equality (:143-146) as check 12. */
WAX_ORACLE_API int wax_oracle_mv2v_parse(const uint8_t* data, uint64_t len, int expect_metric, uint32_t expect_dims,
                                         uint64_t* out_count, uint64_t* out_vec_offset, uint64_t* out_ids_offset) {
    if (len < 36) return 1; /* :718 */
    const uint8_t magic[4] = {0x4D, 0x56, 0x32, 0x56};
    if (memcmp(data, magic, 4) != 0) return 2; /* :727 */
    uint16_t ver; memcpy(&ver, data + 4, 2);
    if (ver != 1) return 3; /* :736 */
    if (data[6] != 2) return 4; /* :743 */
    if (data[7] > 2 || (expect_metric >= 0 && data[7] != (uint8_t)expect_metric)) return 5; /* :750-753 */
    uint32_t dims; memcpy(&dims, data + 8, 4);
    if (expect_dims && dims != expect_dims) return 6; /* :760 */
    uint64_t count, vec_len; memcpy(&count, data + 12, 8); memcpy(&vec_len, data + 20, 8);
    for (int i = 0; i < 8; ++i) if (data[28 + i] != 0) return 7; /* :778 */
    if (vec_len != count * (uint64_t)dims * 4ull) return 8; /* :782 */
    if (len < 36 + vec_len + 8) return 9; /* :785 */
    uint64_t id_len; memcpy(&id_len, data + 36 + vec_len, 8);
    if (id_len != count * 8ull) return 10; /* :806 */
    if (len < 36 + vec_len + 8 + id_len) return 11;
    if (len != 36 + vec_len + 8 + id_len) return 12; /* VectorSerializer.swift:143-146 */
    *out_count = count; *out_vec_offset = 36; *out_ids_offset = 36 + vec_len + 8;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Seeded synthetic embeddings used by the reference's own benchmarks.
 * DeterministicEmbedder — Tests/WaxIntegrationTests/RAGBenchmarkSupport.swift:114-157:
 * FNV-1a 64 over the UTF-8 text, 64-bit LCG, Float(Int64)/Float(Int64.max),
 * optional VectorMath.normalizeL2. */
WAX_ORACLE_API void wax_oracle_deterministic_embed(const uint8_t* text, uint64_t len, uint32_t dims,
                                                   int normalize, float* out) {
    uint64_t h = 14695981039346656037ull; /* :144 */
    for (uint64_t i = 0; i < len; ++i) { h ^= (uint64_t)text[i]; h *= 1099511628211ull; }
    uint64_t state = h;
    for (uint32_t j = 0; j < dims; ++j) {
        state = state * 6364136223846793005ull + 1442695040888963407ull; /* :134 */
        int64_t s = (int64_t)state;
        out[j] = (float)s / (float)INT64_MAX; /* :136 */
    }
    if (normalize) wax_oracle_normalize_l2(out, dims, out);
}

/* MetalVectorEngineBenchmark.swift:33-38 / :84-89: v[i][d] = ((i+d) % 256)/255.
 * Period-256 duplicate rows => massive exact ties (tie-rule stress). */
WAX_ORACLE_API void wax_oracle_tie_pattern(uint64_t row0, uint64_t n, uint32_t dims, float* out) {
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < dims; ++j)
            out[i * (uint64_t)dims + j] = (float)((row0 + i + j) % 256) / 255.0f;
}

/* ---------------------------------------------------------------------------
 * Reciprocal-rank fusion: HybridSearch.rrfFusion(lists:k:) (HybridSearch.swift:25-52) ==
 * UnifiedSearch.rrfFusionResults (UnifiedSearch.swift:590-699) without the diagnostics.
 * Lists in order; a list with weight <= 0 is skipped (:35, :611); per entry (0-based position p):
 * score[id] += weight / Float(max(0, k) + p + 1) in f32, in that order (:37-38, :614-616);
 * bestRank[id] = min(bestRank[id], p + 1) (:39, :617); sources = the lists that named the id (:618-620).
 * Result: every id seen, sorted by score descending, then bestRank ascending, then id ascending (:44-50, :661-665).
 * ids: the lists concatenated; offsets[n_lists + 1]. Returns the number of distinct ids (all are written, the
 * caller sizes the outputs by offsets[n_lists]). Quadratic-free: open-addressing table + qsort. */
typedef struct { uint64_t id; float score; uint32_t best_rank; uint32_t sources; } rrf_ent;
static int cmp_rrf(const void* a, const void* b) {
    const rrf_ent* x = (const rrf_ent*)a; const rrf_ent* y = (const rrf_ent*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->best_rank != y->best_rank) return x->best_rank < y->best_rank ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
WAX_ORACLE_API int64_t wax_oracle_rrf_fuse(uint32_t n_lists, const float* weights, const uint64_t* ids, const uint64_t* offsets,
                                           int64_t k, uint64_t* out_ids, float* out_scores, uint32_t* out_best_rank,
                                           uint32_t* out_sources) {
    const uint64_t total = offsets[n_lists];
    if (total == 0) return 0;
    uint64_t cap = 16;
    while (cap < 2 * total) cap *= 2;
    rrf_ent* tab = (rrf_ent*)calloc((size_t)cap, sizeof(rrf_ent));
    uint8_t* used = (uint8_t*)calloc((size_t)cap, 1);
    const int64_t kc = k > 0 ? k : 0;                       /* max(0, k) */
    uint64_t distinct = 0;
    for (uint32_t l = 0; l < n_lists; ++l) {
        if (!(weights[l] > 0.0f)) continue;                 /* guard list.weight > 0 */
        for (uint64_t p = offsets[l]; p < offsets[l + 1]; ++p) {
            const uint64_t id = ids[p];
            const uint64_t rank = p - offsets[l] + 1;
            const float contribution = weights[l] / (float)(kc + (int64_t)rank);
            uint64_t h = (id * 0x9E3779B97F4A7C15ull) & (cap - 1);
            while (used[h] && tab[h].id != id) h = (h + 1) & (cap - 1);
            if (!used[h]) { used[h] = 1; tab[h].id = id; tab[h].score = 0.0f; tab[h].best_rank = 0xffffffffu; tab[h].sources = 0; ++distinct; }
            tab[h].score += contribution;
            if ((uint32_t)rank < tab[h].best_rank) tab[h].best_rank = (uint32_t)rank;
            tab[h].sources |= 1u << (l & 31u);
        }
    }
    rrf_ent* out = (rrf_ent*)malloc((size_t)(distinct ? distinct : 1) * sizeof(rrf_ent));
    uint64_t m = 0;
    for (uint64_t h = 0; h < cap; ++h) if (used[h]) out[m++] = tab[h];
    qsort(out, (size_t)m, sizeof(rrf_ent), cmp_rrf);
    for (uint64_t i = 0; i < m; ++i) {
        out_ids[i] = out[i].id; out_scores[i] = out[i].score;
        if (out_best_rank) out_best_rank[i] = out[i].best_rank;
        if (out_sources) out_sources[i] = out[i].sources;
    }
    free(out); free(tab); free(used);
    return (int64_t)m;
}
