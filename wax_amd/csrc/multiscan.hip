// multiscan.hip — the exact f32 scan for SEVERAL queries in one pass over the store.
//
// Who needs it: the batched (bf16 MFMA) path answers a query exactly or not at all — a query whose exactness
// certificate fails (dense neighbourhoods: more rows inside the bf16 error band of the k-th neighbour than the candidate
// list holds; duplicated rows; an overflowed survivor segment) is re-run on the exact path. Until round 3 that was one
// scan_kernel launch per such query, serially: on clustered stores a 256-query batch went from 0.23 ms to ~10 ms. Here
// up to 16 of those queries share ONE stream of the f32 store: the rows a wave holds in registers (the same UNROLL x LOADS
// dwordx4 loads per lane as scan_kernel) are scored against every query of the group before the next chunk is
// fetched, so the HBM traffic of the fallback is rows x dims x 4 bytes per GROUP of queries instead of per query.
//
// Bit-identity with the single-query path is the point (results must equal nq calls of wax_hip_search): a row's
// distance is computed with scan_kernel's exact arithmetic — lane g of a GROUP-lane group owns float4s g, g + GROUP, ...,
// one fma chain per component over j, hsum (x + y) + (z + w), the DPP tree of group_sum<GROUP>, finish_distance — so
// (dims -> D4, GROUP) must mirror launch_scan's table. Only the loop order differs (queries inside rows).
//
// Layout per workgroup (4 waves, like scan_kernel):
//   * the group's queries sit in LDS as float4 [nq][D4] (a lane re-reads its LOADS float4s per query: conflict-free
//     ds_read_b128, the two half-waves of GROUP = 32 read the same addresses = broadcast), prefetched one query ahead;
//   * every wave keeps one WaveTopK-style list per query in LDS (CAP slots: a push offers at most 64 / GROUP
//     candidates, so CAP >= k + 64 / GROUP suffices — 64 slots for k <= 60) with its threshold and count beside it;
//     after warm-up a (row, query) pair costs one 64-bit compare, and the rare insert runs out of line;
//   * at the end the four lists of a query are rank-merged (topk.h) into partials[query][workgroup][k];
//     merge_keys_multi (kernels.hip) reduces them per query and attaches frame ids.
// VALU budget: ~35 instructions per (query, row-group) against ~950 cycles of HBM time per row-group and SIMD at
// 8 TB/s: about 12 queries ride on the stream for free, 16 cost ~1.3 passes — against 16 passes before.
#include "kernels.h"
#include "topk.h"

namespace wax {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4m;

enum { MS_COS = WAX_HIP_METRIC_COSINE, MS_DOT = WAX_HIP_METRIC_DOT, MS_L2 = WAX_HIP_METRIC_L2 };

// scan_kernel's arithmetic, restated verbatim (kernels.hip: finish_distance / accumulate / hsum) — any change there
// must be made here too; tests/test_parity_gpu.py::test_multi_query_exact_scan_is_bit_identical pins the pair.
template <int METRIC>
__device__ inline float ms_finish(float acc, float nrm, float q_norm) {
    float d;
    if (METRIC == MS_COS) {
        const float vn = sqrtf(nrm);
        const float sim = (vn > 1e-6f && q_norm > 1e-6f) ? acc / (vn * q_norm) : 0.0f;
        d = 1.0f - sim;
    } else if (METRIC == MS_DOT) {
        d = 1.0f - acc;
    } else {
        d = acc;
    }
    d = (d != d) ? __builtin_inff() : d;
    return d + 0.0f;
}
__device__ inline float ms_hsum(const f32x4& a) { return (a.x + a.y) + (a.z + a.w); }

// Per-(wave, query) selection state in LDS.
struct MsState {
    int64_t tau;   // only keys < tau can still enter the query's top-k (this wave's view)
    int cnt;       // live candidates in the list
    int pad;
};

// Cold path: append the passing lanes' keys to the (wave, query) list, prune when the next push might not fit.
// Returns the (possibly tightened) threshold. Out of line on purpose (see wave_prune).
template <int CAP>
__device__ __attribute__((noinline)) int64_t ms_insert(int64_t* list, MsState* st, int64_t key, bool pass, int k, int room) {
    wave_lds_fence();
    int cnt = st->cnt;
    const unsigned long long mask = __ballot(pass);
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (pass) list[cnt + before] = key;
    cnt += __popcll(mask);
    int64_t tau = st->tau;
    if (cnt > CAP - room) {
        const PruneOut r = wave_prune<CAP, true>((lds_i64*)list, cnt, k);
        cnt = r.cnt;
        tau = r.tau;
    }
    wave_lds_fence();
    if (lane_id() == 0) { st->cnt = cnt; st->tau = tau; }
    wave_lds_fence();
    return tau;
}

template <int D4, int GROUP, int METRIC, int UNROLL, int CAP>
__global__ __launch_bounds__(SCAN_THREADS) void scan_multi_kernel(ScanMultiArgs a) {
    constexpr int LOADS = D4 / GROUP;
    constexpr int RPW = WAVE / GROUP;
    constexpr int RPC = RPW * UNROLL;
    static_assert(D4 % GROUP == 0, "GROUP must divide D4");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nq = a.nq;
    f32x4* qs = reinterpret_cast<f32x4*>(smem);                                         // [nq][D4]
    int64_t* lists = reinterpret_cast<int64_t*>(smem + (size_t)nq * D4 * 16);           // [SCAN_WAVES][nq][CAP]
    MsState* state = reinterpret_cast<MsState*>(lists + (size_t)SCAN_WAVES * nq * CAP);  // [SCAN_WAVES][nq]
    float* qn_s = reinterpret_cast<float*>(state + SCAN_WAVES * nq);                     // [nq]
    int* counts = reinterpret_cast<int*>(qn_s + nq);                                     // [nq][SCAN_WAVES]

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int sub = lane / GROUP;
    const int gl = lane % GROUP;
    const bool owner = (gl == GROUP - 1);
    const uint32_t n = a.n_rows;
    const int k = a.k;

    // stage the group's queries (coalesced float4 loads), norms and list state
    for (uint32_t i = threadIdx.x; i < nq * (uint32_t)D4; i += SCAN_THREADS) {
        const uint32_t qi = i / (uint32_t)D4, c = i - qi * (uint32_t)D4;
        qs[i] = reinterpret_cast<const f32x4*>(a.queries)[(size_t)a.qlist[qi] * D4 + c];
    }
    if (threadIdx.x < nq) qn_s[threadIdx.x] = a.q_norm[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < SCAN_WAVES * nq; i += SCAN_THREADS) { state[i].tau = KEY_PAD; state[i].cnt = 0; }
    __syncthreads();

    const f32x4* __restrict__ store4 = reinterpret_cast<const f32x4*>(a.store);
    int64_t* my_lists = lists + (size_t)wave * nq * CAP;
    MsState* my_state = state + (size_t)wave * nq;
    const lds_f32x4m* qs_l = (const lds_f32x4m*)qs + gl;

    const uint32_t nchunks = (n + RPC - 1) / RPC;
    const uint32_t gwave = blockIdx.x * SCAN_WAVES + wave;
    const uint32_t nwaves = gridDim.x * SCAN_WAVES;

    for (uint32_t chunk = gwave; chunk < nchunks; chunk += nwaves) {
        const uint32_t rbase = chunk * RPC + sub;
        f32x4 v[UNROLL][LOADS];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t r = rbase + u * RPW;
            const uint32_t rc = r < n ? r : n - 1;   // clamp: tail lanes re-read the last row, result discarded
            const f32x4* p = store4 + (size_t)rc * D4 + gl;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) v[u][j] = __builtin_nontemporal_load(p + j * GROUP);
        }
        // ||v||^2 of every row-group once (query-independent): scan_kernel's nrm chain and reduction
        float m[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            m[u] = 0.f;
            if (METRIC == MS_COS) {
                f32x4 nrm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < LOADS; ++j) nrm = __builtin_elementwise_fma(v[u][j], v[u][j], nrm);
                m[u] = group_sum<GROUP>(ms_hsum(nrm));
            }
        }
        // queries inside rows; the next query's slice is fetched from LDS while this one is multiplied
        f32x4 qa[LOADS], qb[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) qa[j] = qs_l[j * GROUP];
        for (uint32_t qi = 0; qi < nq; ++qi) {
            const uint32_t qnext = qi + 1 < nq ? qi + 1 : qi;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) qb[j] = qs_l[(size_t)qnext * D4 + j * GROUP];
            int64_t tau = my_state[qi].tau;
            const float qn = qn_s[qi];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < LOADS; ++j) {
                    if (METRIC == MS_L2) {
                        const f32x4 e = qa[j] - v[u][j];
                        acc = __builtin_elementwise_fma(e, e, acc);
                    } else {
                        acc = __builtin_elementwise_fma(qa[j], v[u][j], acc);
                    }
                }
                const float s = group_sum<GROUP>(ms_hsum(acc));
                const float d = ms_finish<METRIC>(s, m[u], qn);
                const uint32_t r = rbase + u * RPW;
                const int64_t key = make_key(d, a.row_base + r);
                const bool pass = owner && (r < n) && (key < tau);
                if (__any(pass)) tau = ms_insert<CAP>(my_lists + (size_t)qi * CAP, my_state + qi, key, pass, k, RPW);
            }
#pragma unroll
            for (int j = 0; j < LOADS; ++j) qa[j] = qb[j];
        }
    }

    // per query: sort this wave's list, then rank-merge the workgroup's four lists into the query's partial row
    for (uint32_t qi = 0; qi < nq; ++qi) {
        wave_lds_fence();
        const int cnt = my_state[qi].cnt;
        const PruneOut r = wave_prune<CAP, true>((lds_i64*)(my_lists + (size_t)qi * CAP), cnt, k);
        if (lane == 0) counts[qi * SCAN_WAVES + wave] = r.cnt;
    }
    __syncthreads();
    for (uint32_t qi = 0; qi < nq; ++qi)
        block_rank_merge_impl((const lds_i64*)(lists + (size_t)qi * CAP), SCAN_WAVES, (int)(nq * CAP),
                              (const lds_i32*)(counts + qi * SCAN_WAVES), k,
                              a.partials + ((size_t)qi * gridDim.x + blockIdx.x) * k);
}

// ---------------------------------------------------------------------------
// launch table: must mirror launch_scan's (dims -> D4, GROUP); UNROLL = scan_kernel's default for the dimension.

size_t scan_multi_lds_bytes(uint32_t dims, int cap, uint32_t nq) {
    return (size_t)nq * dims * 4 + (size_t)SCAN_WAVES * nq * cap * 8 + (size_t)SCAN_WAVES * nq * sizeof(MsState) + (size_t)nq * 4 +
           (size_t)nq * SCAN_WAVES * 4 + 16;
}

static int ms_group_lanes(uint32_t dims) {
    switch (dims) {
        case 64: return 16;
        case 128: case 384: return 32;
        case 256: case 512: case 768: case 1024: case 1536: return 64;
        default: return 0;
    }
}

bool scan_multi_dims(uint32_t dims) { return ms_group_lanes(dims) != 0; }

int scan_multi_cap(uint32_t dims, int k) {
    const int group = ms_group_lanes(dims);
    if (group == 0 || k < 1 || k > FUSED_MAX_K) return 0;
    const int rpw = WAVE / group;
    return (k + rpw <= 64) ? 64 : 256;
}

// Queries per launch: as many as keep TWO workgroups per CU resident (LDS), at most 16.
uint32_t scan_multi_group(uint32_t dims, int k) {
    const int cap = scan_multi_cap(dims, k);
    if (cap == 0) return 0;
    uint32_t nq = 16;
    while (nq > 1 && scan_multi_lds_bytes(dims, cap, nq) > 78 * 1024) nq >>= 1;
    return scan_multi_lds_bytes(dims, cap, nq) <= 156 * 1024 ? nq : 0;
}

template <int D4, int GROUP, int UNROLL, int METRIC, int CAP>
static hipError_t ms_launch_one(const ScanMultiArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    if (smem > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&scan_multi_kernel<D4, GROUP, METRIC, UNROLL, CAP>), 156 * 1024, configured);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((scan_multi_kernel<D4, GROUP, METRIC, UNROLL, CAP>), dim3(grid), dim3(SCAN_THREADS), smem, st, a);
    return hipGetLastError();
}

template <int D4, int GROUP, int UNROLL>
static hipError_t ms_launch(const ScanMultiArgs& a, int metric, int cap, int grid, size_t smem, hipStream_t st) {
    switch (metric * 2 + (cap == 64 ? 0 : 1)) {
        case MS_COS * 2: return ms_launch_one<D4, GROUP, UNROLL, MS_COS, 64>(a, grid, smem, st);
        case MS_COS * 2 + 1: return ms_launch_one<D4, GROUP, UNROLL, MS_COS, 256>(a, grid, smem, st);
        case MS_DOT * 2: return ms_launch_one<D4, GROUP, UNROLL, MS_DOT, 64>(a, grid, smem, st);
        case MS_DOT * 2 + 1: return ms_launch_one<D4, GROUP, UNROLL, MS_DOT, 256>(a, grid, smem, st);
        case MS_L2 * 2: return ms_launch_one<D4, GROUP, UNROLL, MS_L2, 64>(a, grid, smem, st);
        case MS_L2 * 2 + 1: return ms_launch_one<D4, GROUP, UNROLL, MS_L2, 256>(a, grid, smem, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_multi(const ScanMultiArgs& a, int metric, int grid_cap, hipStream_t st, int* out_grid) {
    const int cap = scan_multi_cap(a.dims, a.k);
    if (cap == 0 || a.nq == 0 || a.nq > 16) return hipErrorInvalidValue;
    const size_t smem = scan_multi_lds_bytes(a.dims, cap, a.nq);
    if (smem > 156 * 1024) return hipErrorInvalidValue;
    const int grid = scan_grid_for(a.n_rows, a.dims, 0, grid_cap);   // variant 0 = the (UNROLL, rows per chunk) used below
    if (out_grid) *out_grid = grid;
    switch (a.dims) {   // (D4, GROUP, UNROLL) = launch_scan's default variant of the dimension
        case 64: return ms_launch<16, 16, 4>(a, metric, cap, grid, smem, st);
        case 128: return ms_launch<32, 32, 4>(a, metric, cap, grid, smem, st);
        case 256: return ms_launch<64, 64, 4>(a, metric, cap, grid, smem, st);
        case 384: return ms_launch<96, 32, 4>(a, metric, cap, grid, smem, st);
        case 512: return ms_launch<128, 64, 4>(a, metric, cap, grid, smem, st);
        case 768: return ms_launch<192, 64, 2>(a, metric, cap, grid, smem, st);
        case 1024: return ms_launch<256, 64, 3>(a, metric, cap, grid, smem, st);
        case 1536: return ms_launch<384, 64, 2>(a, metric, cap, grid, smem, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace wax
