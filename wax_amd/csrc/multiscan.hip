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

constexpr int MS_NQ = 16;      // queries per pass (a shorter last group is padded with idle slots)
constexpr int MS_U = 4;        // row-groups a wave holds in registers per chunk
constexpr int MS_M = MS_NQ * MS_U;   // partial sums per lane per chunk: value m = u * 16 + q

template <int CTRL>
__device__ inline float ms_dpp_move(float v) {   // v of the lane the DPP pattern names (every lane has a source)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// One halving level of the reduce-scatter: n values per lane -> n / 2. Of every pair (v[2i], v[2i+1]) a lane KEEPS one
// (sel: the odd one) and SENDS the other to its partner lane, which keeps the opposite one: out[i] = kept + partner's.
// Both lanes of a pair end up with the sum of the SAME two partial sums group_sum<> adds at this level (it adds them in
// both lanes and keeps both copies) — IEEE addition is commutative, so the bits are scan_kernel's. CTRL names the partner:
// xor 1, xor 2, 7 - i (row_half_mirror), 15 - i (row_mirror), exactly group_sum's patterns.
template <int CTRL, int N>
__device__ inline void ms_halve(float (&v)[MS_M], bool sel) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const float keep = sel ? b : a, send = sel ? a : b;
        v[i] = keep + ms_dpp_move<CTRL>(send);
    }
}
// Levels 5 / 6 (16-lane rows, 32-lane halves): v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of
// the first operand with the even rows (lower half) of the second, so even rows hold both parts of v[2i], odd rows both
// parts of v[2i+1]: one swap + one add per output, no select.
template <int N>
__device__ inline void ms_halve_rows16(float (&v)[MS_M]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}
template <int N>
__device__ inline void ms_halve_rows32(float (&v)[MS_M]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}

// The kernel. Per chunk a wave holds MS_U row-groups (64 / GROUP rows each) in registers and forms, for each of the 16
// queries, the lane-private partial dot products exactly as scan_kernel does (float4 fma chains over j, hsum). The
// cross-lane sums are where a one-query-at-a-time loop drowns: 5-6 DPP adds, the division of the cosine, the key and the
// threshold test for EVERY (query, row-group) pair, in all 64 lanes, for one useful lane — ~50 of ~80 VALU cycles per pair.
// Instead the 64 partial sums a lane holds per chunk go through a reduce-scatter over the GROUP lanes with the SAME
// summation tree (ms_halve): ~3 instructions per pair, and afterwards every lane owns complete sums of DIFFERENT
// (query, row) pairs — the distance, key and threshold test run once per 64 pairs instead of once per pair. ~37 VALU
// cycles per (query, row-group): 16 queries fit under the HBM time of the rows they share.
// Which pair a lane ends with: value m = u * 16 + q, level L keeps bit L-1 of m = the lane's selection bit s_L with
// s1 = b0^b2, s2 = b1^b2, s3 = b2^b3, s4 = b3 (b = lane bits; the XORs make mirror partners agree on the bits already
// fixed), s5 = b4, s6 = b5: q = s1 + 2 s2 + 4 s3 + 8 s4 is a per-lane constant, u comes from the higher bits.
template <int D4, int GROUP, int METRIC, int CAP>
__global__ __launch_bounds__(SCAN_THREADS) void scan_multi_kernel(ScanMultiArgs a) {
    constexpr int LOADS = D4 / GROUP;
    constexpr int RPW = WAVE / GROUP;
    constexpr int RPC = RPW * MS_U;
    constexpr int LEVELS = GROUP == 64 ? 6 : (GROUP == 32 ? 5 : 4);
    constexpr int OUT = MS_M >> LEVELS;          // complete sums per lane per chunk: 4 / 2 / 1
    static_assert(D4 % GROUP == 0 && (GROUP == 16 || GROUP == 32 || GROUP == 64), "GROUP");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nq = a.nq;
    f32x4* qs = reinterpret_cast<f32x4*>(smem);                                            // [16][D4] (slots >= nq: zeros)
    int64_t* lists = reinterpret_cast<int64_t*>(smem + (size_t)MS_NQ * D4 * 16);           // [SCAN_WAVES][16][CAP]
    MsState* state = reinterpret_cast<MsState*>(lists + (size_t)SCAN_WAVES * MS_NQ * CAP);  // [SCAN_WAVES][16]
    float* qn_s = reinterpret_cast<float*>(state + SCAN_WAVES * MS_NQ);                     // [16]
    int* counts = reinterpret_cast<int*>(qn_s + MS_NQ);                                     // [16][SCAN_WAVES]

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int sub = lane / GROUP;
    const int gl = lane % GROUP;
    const uint32_t n = a.n_rows;
    const int k = a.k;

    for (uint32_t i = threadIdx.x; i < (uint32_t)MS_NQ * (uint32_t)D4; i += SCAN_THREADS) {
        const uint32_t qi = i / (uint32_t)D4, c = i - qi * (uint32_t)D4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        qs[i] = qi < nq ? reinterpret_cast<const f32x4*>(a.queries)[(size_t)a.qlist[qi] * D4 + c] : zero;
    }
    if (threadIdx.x < MS_NQ) qn_s[threadIdx.x] = threadIdx.x < nq ? a.q_norm[threadIdx.x] : 0.f;
    for (uint32_t i = threadIdx.x; i < SCAN_WAVES * MS_NQ; i += SCAN_THREADS) { state[i].tau = KEY_PAD; state[i].cnt = 0; }
    __syncthreads();

    const f32x4* __restrict__ store4 = reinterpret_cast<const f32x4*>(a.store);
    int64_t* my_lists = lists + (size_t)wave * MS_NQ * CAP;
    MsState* my_state = state + (size_t)wave * MS_NQ;
    const lds_f32x4m* qs_l = (const lds_f32x4m*)qs + gl;

    // the lane's selection bits and its query
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
    const bool s1 = (b0 ^ b2) != 0, s2 = (b1 ^ b2) != 0, s3 = (b2 ^ b3) != 0, s4 = b3 != 0;
    const int q_lane = (s1 ? 1 : 0) | (s2 ? 2 : 0) | (s3 ? 4 : 0) | (s4 ? 8 : 0);
    const bool q_live = (uint32_t)q_lane < nq;
    const float qn_lane = qn_s[q_lane];
    int64_t tau_lane = KEY_PAD;                              // this wave's threshold for q_lane (kept current by the insert path)
    const int last_of_group = (lane & ~(GROUP - 1)) | (GROUP - 1);

    const uint32_t nchunks = (n + RPC - 1) / RPC;
    const uint32_t gwave = blockIdx.x * SCAN_WAVES + wave;
    const uint32_t nwaves = gridDim.x * SCAN_WAVES;

    // The rows of chunk c + 1 are requested BEFORE chunk c is scored (two register sets, the loop unrolled by two): a wave
    // computes ~4 400 VALU cycles per chunk — as long as the HBM round trip — and with two waves per SIMD nothing else would
    // cover that latency (PMC of the first version: waves parked 60 % of their cycles, VALU busy 45 %).
    auto load_chunk = [&](f32x4 (&v)[MS_U][LOADS], uint32_t chunk) {
        const uint32_t rbase = chunk * RPC + sub;
#pragma unroll
        for (int u = 0; u < MS_U; ++u) {
            const uint32_t r = rbase + u * RPW;
            const uint32_t rc = r < n ? r : n - 1;   // clamp: tail lanes re-read the last row, result discarded
            const f32x4* p = store4 + (size_t)rc * D4 + gl;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) v[u][j] = __builtin_nontemporal_load(p + j * GROUP);
        }
    };
    auto score_chunk = [&](const f32x4 (&v)[MS_U][LOADS], uint32_t chunk) {
        const uint32_t rbase = chunk * RPC + sub;
        // ||v||^2 per row-group, scan_kernel's chain and reduction, then handed to every lane of the group
        // (four named scalars, not an array: a `b4 ? nb[1] : nb[0]` on an array is rewritten into a load from a lane-indexed
        // stack copy — scratch traffic in the hot loop)
        auto row_norm = [&](const f32x4 (&vu)[LOADS]) -> float {
            if (METRIC != MS_COS) return 0.f;
            f32x4 nrm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < LOADS; ++j) nrm = __builtin_elementwise_fma(vu[j], vu[j], nrm);
            const float tot = group_sum<GROUP>(ms_hsum(nrm));                // valid in the group's last lane
            return __shfl(tot, last_of_group, 64);
        };
        const float nb0 = row_norm(v[0]), nb1 = row_norm(v[1]), nb2 = row_norm(v[2]), nb3 = row_norm(v[3]);
        static_assert(MS_U == 4, "four row-groups per chunk");
        // lane-private partial sums of all 64 (row-group, query) pairs; query slices come from LDS, one query ahead
        float part[MS_M];
        f32x4 qa[LOADS], qb[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) qa[j] = qs_l[j * GROUP];
#pragma unroll
        for (int qi = 0; qi < MS_NQ; ++qi) {
            const int qnext = qi + 1 < MS_NQ ? qi + 1 : qi;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) qb[j] = qs_l[(size_t)qnext * D4 + j * GROUP];
#pragma unroll
            for (int u = 0; u < MS_U; ++u) {
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
                part[u * MS_NQ + qi] = ms_hsum(acc);
            }
#pragma unroll
            for (int j = 0; j < LOADS; ++j) qa[j] = qb[j];
        }
        // reduce-scatter over the GROUP lanes with group_sum's tree
        ms_halve<0xB1, MS_M>(part, s1);           // quad_perm [1,0,3,2]
        ms_halve<0x4E, MS_M / 2>(part, s2);       // quad_perm [2,3,0,1]
        ms_halve<0x141, MS_M / 4>(part, s3);      // row_half_mirror
        ms_halve<0x140, MS_M / 8>(part, s4);      // row_mirror
        if (GROUP >= 32) ms_halve_rows16<MS_M / 16>(part);
        if (GROUP >= 64) ms_halve_rows32<MS_M / 32>(part);
        // every lane now owns OUT complete sums, all of query q_lane: value index m = j << LEVELS | (selection bits)
#pragma unroll
        for (int j = 0; j < OUT; ++j) {
            int u;
            float nrm;
            if (GROUP == 16) { u = j; nrm = j == 0 ? nb0 : (j == 1 ? nb1 : (j == 2 ? nb2 : nb3)); }
            else if (GROUP == 32) { u = b4 + 2 * j; nrm = j == 0 ? (b4 ? nb1 : nb0) : (b4 ? nb3 : nb2); }
            else { u = b4 + 2 * b5; nrm = b5 ? (b4 ? nb3 : nb2) : (b4 ? nb1 : nb0); }
            const float d = ms_finish<METRIC>(part[j], nrm, qn_lane);
            const uint32_t r = rbase + (uint32_t)u * RPW;
            const int64_t key = make_key(d, a.row_base + r);
            const bool pass = q_live && (r < n) && (key < tau_lane);
            unsigned long long todo = __ballot(pass);
            while (todo != 0ull) {                               // rare after warm-up: one list at a time
                const int L = (int)__builtin_ctzll(todo);
                const int qL = __builtin_amdgcn_readlane(q_lane, L);
                const bool same = pass && (q_lane == qL);
                // append inline (a few DS operations); only the prune of a full list runs out of line
                lds_i64* list = (lds_i64*)(my_lists + (size_t)qL * CAP);
                MsState* stq = my_state + qL;
                wave_lds_fence();
                int cnt = __builtin_amdgcn_readfirstlane(stq->cnt);
                const unsigned long long mask = __ballot(same);
                const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (same) list[cnt + before] = key;
                cnt += __popcll(mask);
                if (cnt > CAP - 4) {                             // the next push (<= 4 candidates per list) might not fit
                    const PruneOut pr = wave_prune<CAP, true>(list, cnt, k);
                    cnt = pr.cnt;
                    if (q_lane == qL) tau_lane = pr.tau;
                }
                wave_lds_fence();
                if (lane == 0) stq->cnt = cnt;
                todo &= ~mask;
            }
        }
    };
    {
        f32x4 va[MS_U][LOADS], vb[MS_U][LOADS];
        // The prefetch is UNCONDITIONAL (past the end it re-requests the current chunk: L2 hits, discarded): behind a branch
        // the compiler no longer knows how many requests are outstanding and waits vmcnt(0) for the current set — which
        // drains the prefetch it was meant to overlap.
        uint32_t chunk = gwave;
        if (chunk < nchunks) {
            load_chunk(va, chunk);
            for (;;) {
                uint32_t nxt = chunk + nwaves;
                load_chunk(vb, nxt < nchunks ? nxt : chunk);
                __builtin_amdgcn_sched_barrier(0);    // the requests go out before the first use of the current set
                score_chunk(va, chunk);
                chunk = nxt;
                if (chunk >= nchunks) break;
                nxt = chunk + nwaves;
                load_chunk(va, nxt < nchunks ? nxt : chunk);
                __builtin_amdgcn_sched_barrier(0);
                score_chunk(vb, chunk);
                chunk = nxt;
                if (chunk >= nchunks) break;
            }
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
        block_rank_merge_impl((const lds_i64*)(lists + (size_t)qi * CAP), SCAN_WAVES, (int)(MS_NQ * CAP),
                              (const lds_i32*)(counts + qi * SCAN_WAVES), k,
                              a.partials + ((size_t)qi * gridDim.x + blockIdx.x) * k);
}

// ---------------------------------------------------------------------------
// launch table: must mirror launch_scan's (dims -> D4, GROUP).

size_t scan_multi_lds_bytes(uint32_t dims, int cap) {
    return (size_t)MS_NQ * dims * 4 + (size_t)SCAN_WAVES * MS_NQ * cap * 8 + (size_t)SCAN_WAVES * MS_NQ * sizeof(MsState) + (size_t)MS_NQ * 4 +
           (size_t)MS_NQ * SCAN_WAVES * 4 + 16;
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

// A push offers at most 4 candidates to one list (the 4 lanes that share a query), so CAP >= k + 4.
int scan_multi_cap(uint32_t dims, int k) {
    if (ms_group_lanes(dims) == 0 || k < 1 || k > FUSED_MAX_K) return 0;
    return (k + 4 <= 64) ? 64 : 256;
}

// Queries per launch: 16, where the group's queries and lists fit in LDS (k > 60 needs 256-slot lists: 16 queries x 4 waves x
// 2 KB = 128 KB beside the query block — only the smaller dimensions).
uint32_t scan_multi_group(uint32_t dims, int k) {
    const int cap = scan_multi_cap(dims, k);
    if (cap == 0) return 0;
    return scan_multi_lds_bytes(dims, cap) <= 160 * 1024 ? (uint32_t)MS_NQ : 0u;
}

// rows a wave consumes per chunk (MS_U row-groups)
static int ms_rows_per_chunk(uint32_t dims) { return (WAVE / ms_group_lanes(dims)) * MS_U; }

int scan_multi_grid(uint32_t n_rows, uint32_t dims, int grid_cap) {
    if (grid_cap <= 0) grid_cap = 512;
    if (grid_cap > MAX_GRID_BLOCKS) grid_cap = MAX_GRID_BLOCKS;
    const uint64_t nchunks = ((uint64_t)n_rows + ms_rows_per_chunk(dims) - 1) / ms_rows_per_chunk(dims);
    const uint64_t max_waves = (uint64_t)grid_cap * SCAN_WAVES;
    uint64_t waves = nchunks;
    if (nchunks > max_waves) {   // balance the grid-stride loop (as scan_grid_for does)
        const uint64_t iters = (nchunks + max_waves - 1) / max_waves;
        waves = (nchunks + iters - 1) / iters;
    }
    uint64_t blocks = (waves + SCAN_WAVES - 1) / SCAN_WAVES;
    if (blocks < 1) blocks = 1;
    if (blocks > (uint64_t)grid_cap) blocks = grid_cap;
    return (int)blocks;
}

template <int D4, int GROUP, int METRIC, int CAP>
static hipError_t ms_launch_one(const ScanMultiArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    if (smem > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&scan_multi_kernel<D4, GROUP, METRIC, CAP>), 160 * 1024, configured);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((scan_multi_kernel<D4, GROUP, METRIC, CAP>), dim3(grid), dim3(SCAN_THREADS), smem, st, a);
    return hipGetLastError();
}

template <int D4, int GROUP>
static hipError_t ms_launch(const ScanMultiArgs& a, int metric, int cap, int grid, size_t smem, hipStream_t st) {
    switch (metric * 2 + (cap == 64 ? 0 : 1)) {
        case MS_COS * 2: return ms_launch_one<D4, GROUP, MS_COS, 64>(a, grid, smem, st);
        case MS_COS * 2 + 1: return ms_launch_one<D4, GROUP, MS_COS, 256>(a, grid, smem, st);
        case MS_DOT * 2: return ms_launch_one<D4, GROUP, MS_DOT, 64>(a, grid, smem, st);
        case MS_DOT * 2 + 1: return ms_launch_one<D4, GROUP, MS_DOT, 256>(a, grid, smem, st);
        case MS_L2 * 2: return ms_launch_one<D4, GROUP, MS_L2, 64>(a, grid, smem, st);
        case MS_L2 * 2 + 1: return ms_launch_one<D4, GROUP, MS_L2, 256>(a, grid, smem, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_multi(const ScanMultiArgs& a, int metric, int grid_cap, hipStream_t st, int* out_grid) {
    const int cap = scan_multi_cap(a.dims, a.k);
    if (cap == 0 || a.nq == 0 || a.nq > (uint32_t)MS_NQ) return hipErrorInvalidValue;
    const size_t smem = scan_multi_lds_bytes(a.dims, cap);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    const int grid = scan_multi_grid(a.n_rows, a.dims, grid_cap);
    if (out_grid) *out_grid = grid;
    switch (a.dims) {   // (D4, GROUP) = launch_scan's table
        case 64: return ms_launch<16, 16>(a, metric, cap, grid, smem, st);
        case 128: return ms_launch<32, 32>(a, metric, cap, grid, smem, st);
        case 256: return ms_launch<64, 64>(a, metric, cap, grid, smem, st);
        case 384: return ms_launch<96, 32>(a, metric, cap, grid, smem, st);
        case 512: return ms_launch<128, 64>(a, metric, cap, grid, smem, st);
        case 768: return ms_launch<192, 64>(a, metric, cap, grid, smem, st);
        case 1024: return ms_launch<256, 64>(a, metric, cap, grid, smem, st);
        case 1536: return ms_launch<384, 64>(a, metric, cap, grid, smem, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace wax
