// kernels.hip — hand-written CDNA4 (gfx950) kernels for the Wax vector scan + top-k path.
//
// Hot kernel: scan_kernel — ONE fused pass that streams the row-major f32 store
// from HBM with fully coalesced 16-byte loads, reduces each row's dot / norm
// with DPP cross-lane adds, and keeps a running per-wave top-k behind a
// threshold, so the algorithmic traffic is exactly n_rows*dims*4 bytes and no
// n-float distance buffer is written (the reference writes one and re-reads it
// through up to six reduction launches: CosineDistance.metal:233-328 +
// TopKReduction.metal:103-167, driven by MetalVectorEngine.swift:483-575).
//
// Layout / mapping (MI355X-first, not a translation of the Metal one-thread-
// per-row scheme, which would make a 64-wide wave stride 1536 B between lanes):
//   * a row of D floats is D4 = D/4 float4s; GROUP lanes (16/32/64) cooperate on
//     a row, each lane owning LOADS = D4/GROUP float4s at stride GROUP, so a
//     wave-wide dwordx4 load covers 64/GROUP consecutive rows in GROUP*16-byte
//     (>= 256 B, 128-B-line aligned) contiguous runs — every fetched line is
//     fully used;
//   * the query slice a lane needs is loop-invariant => it lives in VGPRs
//     (LOADS float4s); no LDS traffic at all on the streaming path;
//   * each wave keeps UNROLL row-groups (UNROLL*LOADS dwordx4 loads) in flight;
//   * waves walk the store grid-strided by chunk so that at any instant the whole
//     chip reads one contiguous, moving window (TLB- and DRAM-page-friendly);
//   * every row's summation order is fixed by (GROUP, LOADS) alone, so a row's
//     distance is bit-identical wherever it sits — on any shard, any GPU count.
#include <cstddef>
#include <cstring>

#include "kernels.h"
#include "topk.h"

namespace wax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { M_COS = WAX_HIP_METRIC_COSINE, M_DOT = WAX_HIP_METRIC_DOT, M_L2 = WAX_HIP_METRIC_L2 };

template <bool NT>
__device__ inline f32x4 ld16(const f32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// a3 + a6 (distance side): CosineDistance.metal:321-325 rule `sqrt(m) > 1e-6 ? dot/sqrt(m) : 0`,
// extended to the true cosine the CPU path computes (divide by ||q|| as well;
// SURVEY.md §7 "Query-norm semantics"); dot / l2 use USearch's ip / l2sq distances
// (VectorMetric.swift:21-30). NaN -> +inf so it sorts last and is dropped on the host
// like MetalVectorEngine.swift:597; "+ 0.0f" folds -0 into +0.
template <int METRIC>
__device__ inline float finish_distance(float acc, float nrm, float q_norm) {
    float d;
    if (METRIC == M_COS) {
        const float vn = sqrtf(nrm);
        const float sim = (vn > 1e-6f && q_norm > 1e-6f) ? acc / (vn * q_norm) : 0.0f;
        d = 1.0f - sim;
    } else if (METRIC == M_DOT) {
        d = 1.0f - acc;
    } else {
        d = acc;
    }
    d = (d != d) ? __builtin_inff() : d;
    return d + 0.0f;
}

template <int METRIC>
__device__ inline void accumulate(const f32x4& q, const f32x4& v, f32x4& acc, f32x4& nrm) {
    if (METRIC == M_L2) {
        const f32x4 e = q - v;
        acc = __builtin_elementwise_fma(e, e, acc);
    } else {
        acc = __builtin_elementwise_fma(q, v, acc);
        if (METRIC == M_COS) nrm = __builtin_elementwise_fma(v, v, nrm);
    }
}

__device__ inline float hsum(const f32x4& a) { return (a.x + a.y) + (a.z + a.w); }

// Tail of the fused scan kernels: the workgroup's k best keys are in `fin` (LDS, ascending, KEY_PAD padded).
// Ordinary launch: store them as this workgroup's partial list. Fused-merge launch (a.merge_out != nullptr): publish the
// list write-through (8-byte agent-scope stores = `sc1`: in L2 / HBM once the wave's vmcnt drains — no release fence),
// take a ticket; the last arriver re-reads every list with agent-scope loads (served by L2, never a stale L1 line), selects
// the global k with the same wave lists + rank merge, attaches frame ids and writes the kpad hits
// (cdna_hip_programming.md §6 Guideline 16, counter form; *arrive is re-armed by the last arriver).
// Why not an acq_rel ticket: at agent scope a release on this part writes back the XCD's whole L2 (buffer_wbl2 sc1) and an acquire
// invalidates it — microseconds on a kernel whose point is to be a few microseconds. What is ordered here is exactly what must be:
// the k keys are stored write-through with agent-scope atomics (they are in memory when vmcnt reaches 0, which every storing wave
// waits for before the workgroup barrier in front of the ticket), the ticket itself is an agent-scope RMW, and the last arriver reads
// the lists with agent-scope atomic loads that cannot be served from a stale line. Every access that takes part in the hand-over is
// an atomic of agent scope; only the fence instructions a release / acquire pair would add around them are left out.
// ---- last arriver, small k: a k-way merge of the per-workgroup lists instead of streaming them through the wave lists -----------
// The lists are sorted, so the global best key is the smallest of the lists' HEADS. Thread t owns lists t, t + 256, ... (LISTS of
// them: 1 for the small grids, 2 up to SCAN_KWAY_MERGE_GRID) and keeps the next four keys of each in registers; k rounds of
// {workgroup-wide minimum of the heads by DPP + one LDS exchange, the owner of the minimum emits it and advances that list} produce
// the k best in order. A round is a few hundred cycles and one barrier whatever the number of lists; the wave-list path (157 lists,
// k = 10: eight 64-key pushes with two rank-sort prunes per wave, a final prune, a rank merge) took ~10 of the ~20 us a 10K-row
// query spends in this kernel, and grows with the grid. Keys are unique (distinct rows) except KEY_PAD, which nobody "wins".
template <int CTRL>
__device__ inline int64_t dpp_min_i64(int64_t v) {
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(v >> 32), (int)(v >> 32), CTRL, 0xF, 0xF, false);
    const int64_t o = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    return o < v ? o : v;
}
__device__ inline int64_t readlane_i64(int64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <int LISTS>
__device__ __attribute__((noinline)) void kway_merge(const int64_t* partials, int nlists, int k, int64_t* fin, int64_t* xch) {
    const int t = (int)threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    int64_t h[LISTS][4];
    int taken[LISTS];
#pragma unroll
    for (int j = 0; j < LISTS; ++j) {
        const int64_t* lst = partials + (size_t)(t + j * SCAN_THREADS) * k;
        taken[j] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            h[j][i] = (t + j * SCAN_THREADS < nlists && i < k) ? __hip_atomic_load(lst + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_PAD;
    }
    for (int r = 0; r < k; ++r) {
        int64_t head[LISTS];
#pragma unroll
        for (int j = 0; j < LISTS; ++j) {
            const int sel = taken[j] & 3;
            head[j] = sel == 0 ? h[j][0] : sel == 1 ? h[j][1] : sel == 2 ? h[j][2] : h[j][3];
        }
        int64_t mine = head[0];
#pragma unroll
        for (int j = 1; j < LISTS; ++j) mine = head[j] < mine ? head[j] : mine;
        int64_t v = mine;
        v = dpp_min_i64<0xB1>(v);       // quad_perm [1,0,3,2]
        v = dpp_min_i64<0x4E>(v);       // quad_perm [2,3,0,1]
        v = dpp_min_i64<0x141>(v);      // row_half_mirror
        v = dpp_min_i64<0x140>(v);      // row_mirror: every lane holds its row's (16 lanes) minimum
        int64_t w = readlane_i64(v, 0);
        const int64_t w1 = readlane_i64(v, 16), w2 = readlane_i64(v, 32), w3 = readlane_i64(v, 48);
        w = w1 < w ? w1 : w; w = w2 < w ? w2 : w; w = w3 < w ? w3 : w;
        int64_t* slot = xch + (r & 1) * SCAN_WAVES;            // two sets of slots: one barrier per round
        if (lane == 0) slot[wave] = w;
        __syncthreads();
        int64_t best = slot[0];
#pragma unroll
        for (int i = 1; i < SCAN_WAVES; ++i) best = slot[i] < best ? slot[i] : best;
        if (best == KEY_PAD) {                                  // every list is exhausted: pad the rest
            if (t == 0) fin[r] = KEY_PAD;
            continue;
        }
        if (mine == best) {
            fin[r] = best;
#pragma unroll
            for (int j = 0; j < LISTS; ++j) {
                if (head[j] != best) continue;
                ++taken[j];
                if ((taken[j] & 3) == 0) {                       // the next four keys of this list (a run of neighbours in one workgroup's rows)
                    const int64_t* lst = partials + (size_t)(t + j * SCAN_THREADS) * k + taken[j];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        h[j][i] = (taken[j] + i < k) ? __hip_atomic_load(lst + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_PAD;
                }
            }
        }
    }
    __syncthreads();
}

template <int CAP>
__device__ inline void scan_epilogue(const ScanArgs& a, int64_t* lds, int* counts, int64_t* fin) {
    const int k = a.k;
    int64_t* mine = a.partials + (size_t)blockIdx.x * k;
    if (a.merge_out == nullptr) {
        for (int t = (int)threadIdx.x; t < k; t += SCAN_THREADS) mine[t] = fin[t];
        return;
    }
    for (int t = (int)threadIdx.x; t < k; t += SCAN_THREADS)
        __hip_atomic_store(mine + t, fin[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the ticket
    __syncthreads();
    if (threadIdx.x == 0) counts[0] = (int)__hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if ((unsigned)counts[0] != gridDim.x - 1u) return;
    __syncthreads();                                     // everyone has read the ticket before counts[] is reused
    static_assert(SCAN_KWAY_MERGE_GRID <= 2 * SCAN_THREADS, "two lists per thread");
    if (k <= SCAN_KWAY_MAX_K && gridDim.x <= (unsigned)SCAN_KWAY_MERGE_GRID && !a.no_kway) {
        // (the wave lists at the start of `lds` are free: 8 exchange slots)
        if (gridDim.x <= (unsigned)SCAN_THREADS) kway_merge<1>(a.partials, (int)gridDim.x, k, fin, lds);
        else kway_merge<2>(a.partials, (int)gridDim.x, k, fin, lds);
    } else {
        const int lane = lane_id();
        const int wave = (int)(threadIdx.x >> 6);
        WaveTopK<CAP> tk;
        tk.init(lds + wave * CAP, k);
        const uint32_t total = gridDim.x * (uint32_t)k;
        for (uint32_t base = 0; base < total; base += SCAN_THREADS * 4) {
            int64_t keys[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t i = base + r * SCAN_THREADS + threadIdx.x;
                keys[r] = (i < total) ? __hip_atomic_load(a.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_PAD;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) tk.push_wide(keys[r], keys[r] != KEY_PAD);
        }
        tk.finalize();
        if (lane == 0) counts[wave] = tk.cnt;
        __syncthreads();
        block_rank_merge<SCAN_WAVES>(lds, CAP, counts, k, fin);
        __syncthreads();
    }
    for (int t = (int)threadIdx.x; t < a.kpad; t += SCAN_THREADS) {
        wax_hip_hit h;
        h.key = (t < k) ? fin[t] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - a.row_base;
            h.frame_id = (a.ids != nullptr && local < a.n_rows) ? a.ids[local] : (uint64_t)key_row(h.key);
        }
        a.merge_out[t] = h;
    }
    if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.done_flag != nullptr) {
        __threadfence_system();                              // this thread's hits have reached host memory
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.done_flag, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------
// Fused scan + select, compile-time dims.
template <int D4, int GROUP, int METRIC, int UNROLL, bool NT, int CAP, bool WRITE_DIST>
__device__ __forceinline__ void scan_body(const ScanArgs& a, const f32x4* __restrict__ q4);

template <int D4, int GROUP, int METRIC, int UNROLL, bool NT, int CAP, bool WRITE_DIST>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(ScanArgs a) {
    scan_body<D4, GROUP, METRIC, UNROLL, NT, CAP, WRITE_DIST>(a, reinterpret_cast<const f32x4*>(a.query));
}

// The same kernel with the QUERY IN THE KERNEL ARGUMENTS (ScanArgsQ: the scan arguments followed by dims floats, <= 4 KB of
// kernarg). A query then reaches the GPU with the launch packet itself — no upload copy in front of the scan, one packet
// less on the stream per query — which is what a launch-latency-bound store (10K rows: the scan is ~5 us) is made of.
// The lanes read their slice straight from the kernarg segment (a vector load per float4, like the HBM copy they replace).
template <int D4, int GROUP, int METRIC, int UNROLL, bool NT, int CAP>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel_qarg(ScanArgsQ<D4 * 4> aq) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();   // constant -> generic address space: still plain global loads
    const f32x4* q4 = reinterpret_cast<const f32x4*>(ka + offsetof(ScanArgsQ<D4 * 4>, q));
    scan_body<D4, GROUP, METRIC, UNROLL, NT, CAP, false>(aq.a, q4);
}

template <int D4, int GROUP, int METRIC, int UNROLL, bool NT, int CAP, bool WRITE_DIST>
__device__ __forceinline__ void scan_body(const ScanArgs& a, const f32x4* __restrict__ q4) {
    constexpr int LOADS = D4 / GROUP;       // float4s per lane per row
    constexpr int RPW = WAVE / GROUP;       // rows per wave-wide load
    constexpr int RPC = RPW * UNROLL;       // rows per wave per iteration
    static_assert(D4 % GROUP == 0, "GROUP must divide D4");
    if (WRITE_DIST && a.gate != nullptr && *a.gate == 0u) return;   // the short selection answered this query

    __shared__ int64_t lds[WRITE_DIST ? 1 : SCAN_WAVES * CAP + SCAN_WAVES + FUSED_MAX_K];

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int sub = lane / GROUP;
    const int gl = lane % GROUP;
    const bool owner = (gl == GROUP - 1);
    const uint32_t n = a.n_rows;

    const f32x4* __restrict__ store4 = reinterpret_cast<const f32x4*>(a.store);

    f32x4 q[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) q[j] = q4[gl + j * GROUP];

    WaveTopK<CAP> tk;
    if (!WRITE_DIST) tk.init(lds + wave * CAP, a.k);

    const uint32_t nchunks = (n + RPC - 1) / RPC;
    const uint32_t gwave = blockIdx.x * SCAN_WAVES + wave;
    const uint32_t nwaves = gridDim.x * SCAN_WAVES;

    for (uint32_t chunk = gwave; chunk < nchunks; chunk += nwaves) {
        const uint32_t rbase = chunk * RPC + sub;
        if (!WRITE_DIST) tk.make_room(RPC);  // one prune site per iteration keeps the streaming loop small
        f32x4 v[UNROLL][LOADS];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t r = rbase + u * RPW;
            const uint32_t rc = r < n ? r : n - 1;  // clamp: tail lanes re-read the last row, result discarded
            const f32x4* p = store4 + (size_t)rc * D4 + gl;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) v[u][j] = ld16<NT>(p + j * GROUP);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < LOADS; ++j) accumulate<METRIC>(q[j], v[u][j], acc, nrm);
            float s = group_sum<GROUP>(hsum(acc));
            float m = 0.f;
            if (METRIC == M_COS) m = group_sum<GROUP>(hsum(nrm));
            const float d = finish_distance<METRIC>(s, m, a.q_norm);
            const uint32_t r = rbase + u * RPW;
            const bool valid = owner && (r < n);
            if (WRITE_DIST) {
                if (valid) a.dist_out[r] = d;
            } else {
                tk.push(make_key(d, a.row_base + r), valid);
            }
        }
    }

    if (!WRITE_DIST) {
        int* counts = reinterpret_cast<int*>(lds + SCAN_WAVES * CAP);
        int64_t* fin = lds + SCAN_WAVES * CAP + SCAN_WAVES;
        tk.finalize();
        if (lane == 0) counts[wave] = tk.cnt;
        __syncthreads();
        block_rank_merge<SCAN_WAVES>(lds, CAP, counts, a.k, fin);
        __syncthreads();
        scan_epilogue<CAP>(a, lds, counts, fin);
    }
}

// ---------------------------------------------------------------------------
// Any-dims fallback (D not in the specialised list, including D % 4 != 0 and the
// 2-d / 4-d toy corpora of the reference tests): one wave per row, lanes stride
// the row; float4 loads when D % 4 == 0, scalar otherwise. Query read through L1/L2.
template <int METRIC, int CAP, bool WRITE_DIST>
__global__ __launch_bounds__(SCAN_THREADS) void scan_generic_kernel(ScanArgs a) {
    __shared__ int64_t lds[WRITE_DIST ? 1 : SCAN_WAVES * CAP + SCAN_WAVES + FUSED_MAX_K];
    if (WRITE_DIST && a.gate != nullptr && *a.gate == 0u) return;   // the short selection answered this query
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t n = a.n_rows, D = a.dims;
    const bool vec4 = (D & 3u) == 0;
    const uint32_t D4 = D >> 2;

    WaveTopK<CAP> tk;
    if (!WRITE_DIST) tk.init(lds + wave * CAP, a.k);

    const uint32_t gwave = blockIdx.x * SCAN_WAVES + wave;
    const uint32_t nwaves = gridDim.x * SCAN_WAVES;
    for (uint32_t r = gwave; r < n; r += nwaves) {
        if (!WRITE_DIST) tk.make_room(1);
        const float* row = a.store + (size_t)r * D;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
        if (vec4) {
            const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
            const f32x4* q4 = reinterpret_cast<const f32x4*>(a.query);
            for (uint32_t c = lane; c < D4; c += WAVE) accumulate<METRIC>(q4[c], row4[c], acc, nrm);
        } else {
            for (uint32_t c = lane; c < D; c += WAVE) {
                const f32x4 qq = {a.query[c], 0.f, 0.f, 0.f};
                const f32x4 vv = {row[c], 0.f, 0.f, 0.f};
                accumulate<METRIC>(qq, vv, acc, nrm);
            }
        }
        float s = group_sum<64>(hsum(acc));
        float m = 0.f;
        if (METRIC == M_COS) m = group_sum<64>(hsum(nrm));
        const float d = finish_distance<METRIC>(s, m, a.q_norm);
        const bool valid = (lane == WAVE - 1);
        if (WRITE_DIST) {
            if (valid) a.dist_out[r] = d;
        } else {
            tk.push(make_key(d, a.row_base + r), valid);
        }
    }
    if (!WRITE_DIST) {
        int* counts = reinterpret_cast<int*>(lds + SCAN_WAVES * CAP);
        int64_t* fin = lds + SCAN_WAVES * CAP + SCAN_WAVES;
        tk.finalize();
        if (lane == 0) counts[wave] = tk.cnt;
        __syncthreads();
        block_rank_merge<SCAN_WAVES>(lds, CAP, counts, a.k, fin);
        __syncthreads();
        scan_epilogue<CAP>(a, lds, counts, fin);
    }
}

// ---------------------------------------------------------------------------
// variant registry

struct DimSpec { uint32_t dims; int group; };
static const DimSpec kDimSpecs[] = {
    {64, 16}, {128, 32}, {256, 64}, {384, 32}, {512, 64}, {768, 64}, {1024, 64}, {1536, 64},
};

struct VariantSpec { int unroll; int nt; };
// variant 0 is the production default; the others exist for on-device sweeps (tools/sweep.py).
static const VariantSpec kVariants384[] = {{4, 1}, {4, 0}, {2, 1}, {2, 0}, {8, 1}, {8, 0}, {1, 1}};
static const VariantSpec kVariants768[] = {{2, 1}, {2, 0}, {4, 1}, {1, 1}};

static int default_unroll(uint32_t dims, int group) {
    const int loads = (int)(dims / 4) / group;
    int u = 12 / loads;
    if (u < 1) u = 1;
    if (u > 4) u = 4;
    return u;
}

static const DimSpec* find_dim(uint32_t dims) {
    for (const auto& s : kDimSpecs)
        if (s.dims == dims) return &s;
    return nullptr;
}

int scan_variant_count(uint32_t dims) {
    if (dims == 384) return (int)(sizeof(kVariants384) / sizeof(kVariants384[0]));
    if (dims == 768) return (int)(sizeof(kVariants768) / sizeof(kVariants768[0]));
    return 1;
}

bool scan_variant_info(uint32_t dims, int variant, ScanVariantInfo* out) {
    if (variant < 0 || variant >= scan_variant_count(dims)) return false;
    const DimSpec* s = find_dim(dims);
    if (!s) {
        *out = ScanVariantInfo{1, 0, 64, 1, 0};
        return true;
    }
    VariantSpec v;
    if (dims == 384) v = kVariants384[variant];
    else if (dims == 768) v = kVariants768[variant];
    else v = VariantSpec{default_unroll(dims, s->group), 1};
    *out = ScanVariantInfo{v.unroll, v.nt, s->group, (WAVE / s->group) * v.unroll, 1};
    return true;
}

int scan_grid_for(uint32_t n_rows, uint32_t dims, int variant, int grid_cap) {
    ScanVariantInfo info;
    if (!scan_variant_info(dims, variant, &info)) scan_variant_info(dims, 0, &info);
    // Default 512 workgroups = 2 per CU (8 of the 20 wave slots the kernel's 82 VGPRs allow). The
    // on-device sweeps (profiles/r01/*sweep.json) show 512..8192 workgroups within 1 % of each other
    // at 10M x 384 when the kernel runs alone, but a grid that fills the chip (1024 = 4 per CU) loses
    // 8 % as soon as anything co-runs (the merge kernel, a copy, an RCCL kernel): the co-runner displaces
    // a few long-lived scan workgroups, which then start late and finish alone at single-workgroup
    // bandwidth. Half occupancy keeps every scan workgroup resident from the first cycle.
    if (grid_cap <= 0) grid_cap = 512;
    if (grid_cap > MAX_GRID_BLOCKS) grid_cap = MAX_GRID_BLOCKS;
    const uint64_t nchunks = ((uint64_t)n_rows + info.rows_per_chunk - 1) / info.rows_per_chunk;
    const uint64_t max_waves = (uint64_t)grid_cap * SCAN_WAVES;
    uint64_t waves = nchunks;
    const uint64_t small_waves = (uint64_t)SCAN_FUSE_MERGE_GRID * SCAN_WAVES;   // 640 waves = 160 workgroups
    if (nchunks >= 64 && nchunks <= 4 * small_waves && (uint64_t)grid_cap >= (uint64_t)SCAN_FUSE_MERGE_GRID) {
        // Small stores (<= 20 K rows at 8 rows per chunk): as many chunks per wave (1 .. 4) as keep the grid at <= 160 workgroups —
        // few enough partial lists for the last-arriving workgroup to do the final merge itself for every k the fused kernels serve,
        // and the best grid measured at every such size: blocking call, 384-d, top-10 (sessions r04_s22 / s23) —
        //    5K rows: 157 workgroups 19.8 us, 79 (two chunks per wave) 21.0     10K: 157 (two chunks) 21.6, 313 (one) 23.3
        //   20K rows: 157 (four chunks) 24.2, 313 (two) 25.6                     40K: 157 30.8, 417 (default rule) 29.6: the rule ends here
        const uint64_t iters = (nchunks + small_waves - 1) / small_waves;
        waves = (nchunks + iters - 1) / iters;
    } else if (nchunks > max_waves) {
        // Balance the grid-stride loop: every wave runs the same number of iterations (+-1 chunk
        // in total) instead of leaving a mostly idle last iteration (26 % idle at 1M x 384).
        const uint64_t iters = (nchunks + max_waves - 1) / max_waves;
        waves = (nchunks + iters - 1) / iters;
    }
    uint64_t blocks = (waves + SCAN_WAVES - 1) / SCAN_WAVES;
    if (blocks < 1) blocks = 1;
    if (blocks > (uint64_t)grid_cap) blocks = grid_cap;
    return (int)blocks;
}

template <int D4, int GROUP, int UNROLL, bool NT, int METRIC>
static hipError_t launch_metric(const ScanArgs& a, int cap, bool write_dist, int grid, hipStream_t st) {
    if (write_dist) {
        launch_kernel((scan_kernel<D4, GROUP, METRIC, UNROLL, NT, 128, true>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    } else if (cap <= 128) {
        launch_kernel((scan_kernel<D4, GROUP, METRIC, UNROLL, NT, 128, false>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    } else {
        launch_kernel((scan_kernel<D4, GROUP, METRIC, UNROLL, NT, 256, false>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    }
    return hipGetLastError();
}

// query in the kernel arguments (args.query_host != nullptr; fused path only)
template <int D4, int GROUP, int UNROLL, bool NT, int METRIC>
static hipError_t launch_qarg_metric(const ScanArgs& a, int cap, int grid, hipStream_t st) {
    ScanArgsQ<D4 * 4> aq;
    aq.a = a;
    std::memcpy(aq.q, a.query_host, sizeof(aq.q));
    aq.a.query = nullptr;
    if (cap <= 128) launch_kernel((scan_kernel_qarg<D4, GROUP, METRIC, UNROLL, NT, 128>), dim3(grid), dim3(SCAN_THREADS), 0, st, aq);
    else launch_kernel((scan_kernel_qarg<D4, GROUP, METRIC, UNROLL, NT, 256>), dim3(grid), dim3(SCAN_THREADS), 0, st, aq);
    return hipGetLastError();
}
template <int D4, int GROUP, int UNROLL, bool NT>
static hipError_t launch_qarg(const ScanArgs& a, int metric, int cap, int grid, hipStream_t st) {
    switch (metric) {
        case M_COS: return launch_qarg_metric<D4, GROUP, UNROLL, NT, M_COS>(a, cap, grid, st);
        case M_DOT: return launch_qarg_metric<D4, GROUP, UNROLL, NT, M_DOT>(a, cap, grid, st);
        case M_L2: return launch_qarg_metric<D4, GROUP, UNROLL, NT, M_L2>(a, cap, grid, st);
    }
    return hipErrorInvalidValue;
}

template <int D4, int GROUP, int UNROLL, bool NT>
static hipError_t launch_full(const ScanArgs& a, int metric, int cap, bool write_dist, int grid, hipStream_t st) {
    switch (metric) {
        case M_COS: return launch_metric<D4, GROUP, UNROLL, NT, M_COS>(a, cap, write_dist, grid, st);
        case M_DOT: return launch_metric<D4, GROUP, UNROLL, NT, M_DOT>(a, cap, write_dist, grid, st);
        case M_L2: return launch_metric<D4, GROUP, UNROLL, NT, M_L2>(a, cap, write_dist, grid, st);
    }
    return hipErrorInvalidValue;
}

// sweep-only variants: cosine, k <= 64, fused path; anything else falls back to variant 0
template <int D4, int GROUP, int UNROLL, bool NT>
static hipError_t launch_sweep(const ScanArgs& a, int grid, hipStream_t st) {
    launch_kernel((scan_kernel<D4, GROUP, M_COS, UNROLL, NT, 128, false>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_generic_metric(const ScanArgs& a, int cap, bool write_dist, int grid, hipStream_t st) {
    if (write_dist) {
        launch_kernel((scan_generic_kernel<METRIC, 128, true>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    } else if (cap <= 128) {
        launch_kernel((scan_generic_kernel<METRIC, 128, false>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    } else {
        launch_kernel((scan_generic_kernel<METRIC, 256, false>), dim3(grid), dim3(SCAN_THREADS), 0, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs& args, int metric, int variant, int cap, bool write_dist, int grid_cap,
                       hipStream_t st, int* out_grid, bool* out_merged) {
    if (variant < 0 || variant >= scan_variant_count(args.dims)) variant = 0;
    const bool sweepable = (metric == M_COS && cap <= 128 && !write_dist);
    if (!sweepable) variant = 0;
    const int grid = scan_grid_for(args.n_rows, args.dims, variant, grid_cap);
    if (out_grid) *out_grid = grid;
    ScanArgs a = args;
    const bool fuse = !write_dist && a.merge_out != nullptr && a.arrive != nullptr && scan_merges_in_kernel(grid, a.k, a.no_kway == 0, a.n_rows, a.dims) && a.kpad >= a.k;
    const bool small = a.plain_loads != 0;             // a store that lives in the caches between queries (see below; the caller decides)
    if (!fuse) { a.merge_out = nullptr; a.arrive = nullptr; a.done_flag = nullptr; }
    if (out_merged) *out_merged = fuse;
    // query in the kernel arguments: the BASELINE dimensions, default variant, fused path (the caller decides when — launch_scan
    // only refuses what it has no kernel for, by falling through to the pointer form, which needs args.query)
    // Small stores (a.plain_loads: the caller's rule) read their rows with ordinary loads instead of the streaming (non-temporal)
    // loads of the large-store kernels: 0.7 - 0.9 us per query faster up to ~30 MB of rows, equal to 230 MB, slower beyond
    // (profiles/HISTORY.md).
    if (a.query_host != nullptr && !write_dist && variant == 0) {
        if (a.dims == 384) return (fuse && small) ? launch_qarg<96, 32, 4, false>(a, metric, cap, grid, st) : launch_qarg<96, 32, 4, true>(a, metric, cap, grid, st);
        if (a.dims == 768) return (fuse && small) ? launch_qarg<192, 64, 2, false>(a, metric, cap, grid, st) : launch_qarg<192, 64, 2, true>(a, metric, cap, grid, st);
    }
    if (a.query == nullptr) return hipErrorInvalidValue;
    switch (a.dims) {
        case 64: return launch_full<16, 16, 4, true>(a, metric, cap, write_dist, grid, st);
        case 128: return launch_full<32, 32, 4, true>(a, metric, cap, write_dist, grid, st);
        case 256: return launch_full<64, 64, 4, true>(a, metric, cap, write_dist, grid, st);
        case 384:
            switch (variant) {
                case 1: return launch_sweep<96, 32, 4, false>(a, grid, st);
                case 2: return launch_sweep<96, 32, 2, true>(a, grid, st);
                case 3: return launch_sweep<96, 32, 2, false>(a, grid, st);
                case 4: return launch_sweep<96, 32, 8, true>(a, grid, st);
                case 5: return launch_sweep<96, 32, 8, false>(a, grid, st);
                case 6: return launch_sweep<96, 32, 1, true>(a, grid, st);
                default: return launch_full<96, 32, 4, true>(a, metric, cap, write_dist, grid, st);
            }
        case 512: return launch_full<128, 64, 4, true>(a, metric, cap, write_dist, grid, st);
        case 768:
            switch (variant) {
                case 1: return launch_sweep<192, 64, 2, false>(a, grid, st);
                case 2: return launch_sweep<192, 64, 4, true>(a, grid, st);
                case 3: return launch_sweep<192, 64, 1, true>(a, grid, st);
                default: return launch_full<192, 64, 2, true>(a, metric, cap, write_dist, grid, st);
            }
        case 1024: return launch_full<256, 64, 3, true>(a, metric, cap, write_dist, grid, st);
        case 1536: return launch_full<384, 64, 2, true>(a, metric, cap, write_dist, grid, st);
        default: break;
    }
    switch (metric) {
        case M_COS: return launch_generic_metric<M_COS>(a, cap, write_dist, grid, st);
        case M_DOT: return launch_generic_metric<M_DOT>(a, cap, write_dist, grid, st);
        case M_L2: return launch_generic_metric<M_L2>(a, cap, write_dist, grid, st);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Final merge: `n_lists` per-workgroup lists of k ascending keys -> k hits, one launch, one workgroup
// (a4's iterated topKReduceEntries passes, TopKReduction.metal:136-167).
//   stage 1: top-k over the list HEADS only (one key per list). With T = the k-th smallest head,
//            a list whose head is > T cannot contribute: k lists already hold k keys <= T.
//   stage 2: the <= k surviving lists (k*k keys) go through the same wave-private selection,
//            the workgroup rank-merges, and frame ids are looked up.
// 10 240 keys (1024 workgroups, k = 10) cost ~1 push + 1 prune per wave instead of ~10 + 4.
template <int CAP>
__global__ __launch_bounds__(MERGE_THREADS) void merge_keys_kernel(const int64_t* __restrict__ in, uint32_t n_lists,
                                                                   int k, int kpad,
                                                                   const uint64_t* __restrict__ ids,
                                                                   uint32_t row_base, uint32_t n_rows,
                                                                   wax_hip_hit* __restrict__ out,
                                                                   const uint32_t* __restrict__ qlist, uint32_t out_stride,
                                                                   const uint32_t* __restrict__ gate) {
    __shared__ int64_t lds[MERGE_WAVES * CAP + MERGE_WAVES + 2 * FUSED_MAX_K + 1];
    if (gate != nullptr && *gate == 0u) return;               // the short selection in front of this launch answered
    // one workgroup per query (launch_merge_keys_multi); the single-query launch has one workgroup and out_stride = 0
    in += (size_t)blockIdx.x * n_lists * (uint32_t)k;
    out += (size_t)(qlist ? qlist[blockIdx.x] : blockIdx.x) * out_stride;
    int* counts = reinterpret_cast<int*>(lds + MERGE_WAVES * CAP);
    int64_t* fin = lds + MERGE_WAVES * CAP + MERGE_WAVES;
    int* sel = reinterpret_cast<int*>(fin + FUSED_MAX_K);           // FUSED_MAX_K list indices
    int* sel_count = reinterpret_cast<int*>(fin + 2 * FUSED_MAX_K);
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    constexpr int MERGE_LOADS = 4;  // independent loads in flight per lane before the first push

    WaveTopK<CAP> tk;
    tk.init(lds + wave * CAP, k);
    for (uint32_t base = 0; base < n_lists; base += MERGE_THREADS * MERGE_LOADS) {
        int64_t keys[MERGE_LOADS];
#pragma unroll
        for (int r = 0; r < MERGE_LOADS; ++r) {
            const uint32_t b = base + r * MERGE_THREADS + threadIdx.x;
            keys[r] = (b < n_lists) ? in[(size_t)b * k] : KEY_PAD;
        }
#pragma unroll
        for (int r = 0; r < MERGE_LOADS; ++r) {
            tk.push_wide(keys[r], keys[r] != KEY_PAD);
        }
    }
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    if (threadIdx.x == 0) *sel_count = 0;
    __syncthreads();
    block_rank_merge<MERGE_WAVES>(lds, CAP, counts, k, fin);
    __syncthreads();
    const int64_t head_cut = fin[k - 1];  // KEY_PAD if fewer than k non-empty lists: keep them all
    for (uint32_t b = threadIdx.x; b < n_lists; b += MERGE_THREADS) {
        const int64_t h = in[(size_t)b * k];
        if (h != KEY_PAD && h <= head_cut) {
            const int pos = atomicAdd(sel_count, 1);
            if (pos < FUSED_MAX_K) sel[pos] = (int)b;
        }
    }
    __syncthreads();
    const int nsel = *sel_count < k ? *sel_count : k;   // keys are unique => at most k lists survive
    const uint32_t total = (uint32_t)nsel * (uint32_t)k;
    __syncthreads();                                     // everyone has read fin/sel_count before they are reused
    tk.init(lds + wave * CAP, k);
    for (uint32_t base = 0; base < total; base += MERGE_THREADS * MERGE_LOADS) {
        int64_t keys[MERGE_LOADS];
#pragma unroll
        for (int r = 0; r < MERGE_LOADS; ++r) {
            const uint32_t i = base + r * MERGE_THREADS + threadIdx.x;
            keys[r] = KEY_PAD;
            if (i < total) {
                const uint32_t li = i / (uint32_t)k;
                keys[r] = in[(size_t)sel[li] * k + (i - li * (uint32_t)k)];
            }
        }
#pragma unroll
        for (int r = 0; r < MERGE_LOADS; ++r) {
            tk.push_wide(keys[r], keys[r] != KEY_PAD);
        }
    }
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    __syncthreads();
    block_rank_merge<MERGE_WAVES>(lds, CAP, counts, k, fin);
    __syncthreads();
    for (int t = (int)threadIdx.x; t < kpad; t += MERGE_THREADS) {
        wax_hip_hit h;
        h.key = (t < k) ? fin[t] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - row_base;
            h.frame_id = (ids != nullptr && local < n_rows) ? ids[local] : (uint64_t)key_row(h.key);
        }
        out[t] = h;
    }
}

hipError_t launch_merge_keys(const int64_t* d_in, uint32_t n_in, int k, int kpad, const uint64_t* d_ids,
                             uint32_t row_base, uint32_t n_rows, wax_hip_hit* d_out, int cap, hipStream_t st, const uint32_t* gate) {
    if (k > FUSED_MAX_K || k < 1 || kpad < k || n_in % (uint32_t)k != 0) return hipErrorInvalidValue;
    const uint32_t n_lists = n_in / (uint32_t)k;
    if (cap <= 128)
        hipLaunchKernelGGL((merge_keys_kernel<128>), dim3(1), dim3(MERGE_THREADS), 0, st, d_in, n_lists, k, kpad, d_ids,
                           row_base, n_rows, d_out, (const uint32_t*)nullptr, 0u, gate);
    else
        hipLaunchKernelGGL((merge_keys_kernel<256>), dim3(1), dim3(MERGE_THREADS), 0, st, d_in, n_lists, k, kpad, d_ids,
                           row_base, n_rows, d_out, (const uint32_t*)nullptr, 0u, gate);
    return hipGetLastError();
}

hipError_t launch_merge_keys_multi(const int64_t* d_in, uint32_t n_lists, int k, const uint64_t* d_ids, uint32_t row_base,
                                   uint32_t n_rows, wax_hip_hit* d_out_base, uint32_t out_stride, const uint32_t* d_qlist,
                                   uint32_t nq, hipStream_t st) {
    if (k > FUSED_MAX_K || k < 1 || out_stride < (uint32_t)k || nq == 0 || n_lists == 0) return hipErrorInvalidValue;
    if (k <= 64)
        hipLaunchKernelGGL((merge_keys_kernel<128>), dim3(nq), dim3(MERGE_THREADS), 0, st, d_in, n_lists, k, (int)out_stride, d_ids,
                           row_base, n_rows, d_out_base, d_qlist, out_stride, (const uint32_t*)nullptr);
    else
        hipLaunchKernelGGL((merge_keys_kernel<256>), dim3(nq), dim3(MERGE_THREADS), 0, st, d_in, n_lists, k, (int)out_stride, d_ids,
                           row_base, n_rows, d_out_base, d_qlist, out_stride, (const uint32_t*)nullptr);
    return hipGetLastError();
}

// Gathered shard hits -> global top-k (SURVEY.md §8e "merge G*k -> k"). Select on keys, then
// every input hit binary-searches the sorted winners to deposit its frame id.
// One workgroup per query: blockIdx.x = q of nq. Shard s's list for query q is in[(s * nq + q) * kin .. + kin),
// out[q * k .. + k) receives the merged top-k (nq = 1, kin = n: one flat list, the single-query exchange).
__global__ __launch_bounds__(MERGE_THREADS) void merge_hits_kernel(const wax_hip_hit* __restrict__ in, uint32_t n_shards,
                                                                   uint32_t nq, uint32_t kin, int k,
                                                                   wax_hip_hit* __restrict__ out_all, uint32_t out_stride) {
    constexpr int CAP = 256;
    __shared__ int64_t lds[MERGE_WAVES * CAP + MERGE_WAVES + FUSED_MAX_K];
    int* counts = reinterpret_cast<int*>(lds + MERGE_WAVES * CAP);
    int64_t* fin = lds + MERGE_WAVES * CAP + MERGE_WAVES;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t q = blockIdx.x;
    const uint32_t n = n_shards * kin;
    wax_hip_hit* __restrict__ out = out_all + (size_t)q * out_stride;   // rows out_stride hits wide, padded past k
    auto src = [&](uint32_t i) -> const wax_hip_hit* {
        const uint32_t s = i / kin, j = i - s * kin;
        return in + ((size_t)s * nq + q) * kin + j;
    };
    WaveTopK<CAP, false> tk;  // foreign keys: a misconfigured shard layout may offer a row twice
    tk.init(lds + wave * CAP, k);
    for (uint32_t base = wave * WAVE; base < n; base += MERGE_THREADS) {
        const uint32_t i = base + lane;
        const bool inb = i < n;
        const int64_t key = inb ? src(i)->key : KEY_PAD;
        tk.push_wide(key, inb && key != KEY_PAD);
    }
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    __syncthreads();
    block_rank_merge<MERGE_WAVES>(lds, CAP, counts, k, fin);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < out_stride; t += MERGE_THREADS)
        if (t >= (uint32_t)k || fin[t] == KEY_PAD) out[t] = wax_hip_hit{KEY_PAD, ID_PAD};
    for (uint32_t i = threadIdx.x; i < n; i += MERGE_THREADS) {
        const wax_hip_hit h = *src(i);
        if (h.key == KEY_PAD) continue;
        int lo = 0, hi = k;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (fin[mid] < h.key) lo = mid + 1; else hi = mid;
        }
        if (lo < k && fin[lo] == h.key) out[lo] = h;
    }
}

hipError_t launch_merge_hits(const wax_hip_hit* d_in, uint32_t n, int k, wax_hip_hit* d_out, hipStream_t st) {
    if (k > FUSED_MAX_K || k < 1 || n > 16384) return hipErrorInvalidValue;
    if (n == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_hits_kernel, dim3(1), dim3(MERGE_THREADS), 0, st, d_in, 1u, 1u, n, k, d_out, (uint32_t)k);
    return hipGetLastError();
}

hipError_t launch_merge_batch_hits(const wax_hip_hit* d_in, uint32_t n_shards, uint32_t nq, uint32_t kin, int k,
                                   wax_hip_hit* d_out, hipStream_t st, uint32_t out_stride) {
    if (out_stride == 0) out_stride = (uint32_t)k;
    if (k > FUSED_MAX_K || k < 1 || nq == 0 || n_shards == 0 || kin == 0 || (uint64_t)n_shards * kin > 16384 || out_stride < (uint32_t)k)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_hits_kernel, dim3(nq), dim3(MERGE_THREADS), 0, st, d_in, n_shards, nq, kin, k, d_out, out_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// General selection (k up to 10 000): exact k-th smallest 64-bit key by MSB-first 8-bit
// radix select over (ordered distance : row), then compaction, rank sort and id lookup.
// Serves MetalVectorEngine's "k > 256 => CPU heap" branch (MetalVectorEngine.swift:452-455,
// 614-625) without leaving the device.

__device__ inline uint64_t ukey_of(float d, uint32_t row) {
    return (uint64_t)make_key(d, row) ^ 0x8000000000000000ull;  // unsigned-ordered
}

// Round 6: the pick is fused into the histogram pass (its last-arriving workgroup picks and re-arms histogram and ticket), there is no
// init launch, a wave whose keys all fall into one bin — the rule in the first passes, distances of one store share their leading bits —
// adds once instead of 64 times, and a pass that has nothing left to decide returns at once: after the fourth pass the distance of the
// k-th key is known, and when every key with that distance is needed (always, unless equal distances straddle rank k) the threshold is
// final — the four row passes are four empty launches. 8 + 3 launches behind the distance pass, four of them real passes over the
// distances, instead of 20 launches and eight passes. (11-bit digits — six passes, three real — were measured and lost: every
// workgroup flushes up to 2 048 bins with global atomics, 4 M atomics per pass at 10M rows: profiles/r06/e_general_selection_*.)
constexpr int SEL_PASSES = 8;
constexpr int SEL_BINS = 256;
// state[0] = prefix (the digits decided so far, right-aligned), state[1] = rank still to find among the keys that carry the prefix,
// state[2] = 1 once state[0] is the final threshold (the exact k-th smallest unsigned key, or the largest key of its tie group when
// the whole group is needed). hist[SEL_BINS] = arrival ticket of the pass; *counter = slots handed out by the compaction.
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ dist, uint32_t n,
                                                          uint32_t row_base, int pass, uint32_t k,
                                                          uint64_t* __restrict__ state,
                                                          uint32_t* __restrict__ hist, uint32_t* __restrict__ counter,
                                                          const uint32_t* __restrict__ gate) {
    __shared__ uint32_t h[SEL_BINS];
    __shared__ uint32_t part[SEL_BINS];
    __shared__ uint32_t last_s;
    if (gate != nullptr && *gate == 0u) return;               // the short selection answered this query
    const bool first = pass == 0;
    if (!first && state[2] != 0) return;                      // nothing left to decide
    const uint64_t prefix = first ? 0ull : state[0];
    const uint64_t rem = first ? (uint64_t)k : state[1];      // (read here, by everyone: the picking thread overwrites it at the end)
    h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    auto count_one = [&](float d, uint32_t i, bool live) {
        const uint64_t u = ukey_of(d, row_base + i);
        const bool match = live && (first || ((u >> (shift + 8)) == prefix));
        const uint32_t bin = match ? (uint32_t)(u >> shift) & 0xffu : 0xFFFFFFFFu;
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
        if (__ballot(bin != b0) == 0ull) {                    // the whole wave in one bin (or nobody matches): one LDS add, not 64
            if (b0 != 0xFFFFFFFFu) {
                const unsigned long long act = __ballot(true);
                if (lane_id() == (int)__builtin_ctzll(act)) atomicAdd(&h[b0], (uint32_t)__builtin_popcountll(act));
            }
        } else if (match) {
            atomicAdd(&h[bin], 1u);
        }
    };
    // four consecutive distances per thread and step (one 16-byte load; the distance array is 16-byte aligned: hipMalloc), two steps in
    // flight: a thread's chain of dependent global loads was the whole cost of a pass (19 round trips at 10M rows)
    const uint32_t n4 = n >> 2;
    const f32x4* __restrict__ dist4 = reinterpret_cast<const f32x4*>(dist);
    const uint32_t stride = gridDim.x * 256;
    uint32_t i4 = blockIdx.x * 256 + threadIdx.x;
    for (; i4 + 3 * stride < n4; i4 += 4 * stride) {          // four loads in flight: at 10M rows a thread's whole share in ~1 round trip
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = dist4[i4 + j * stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t j4 = i4 + j * stride;
            count_one(v[j].x, 4 * j4, true); count_one(v[j].y, 4 * j4 + 1, true); count_one(v[j].z, 4 * j4 + 2, true); count_one(v[j].w, 4 * j4 + 3, true);
        }
    }
    for (; i4 + stride < n4; i4 += 2 * stride) {
        const f32x4 a = dist4[i4], b = dist4[i4 + stride];
        count_one(a.x, 4 * i4, true); count_one(a.y, 4 * i4 + 1, true); count_one(a.z, 4 * i4 + 2, true); count_one(a.w, 4 * i4 + 3, true);
        const uint32_t j4 = i4 + stride;
        count_one(b.x, 4 * j4, true); count_one(b.y, 4 * j4 + 1, true); count_one(b.z, 4 * j4 + 2, true); count_one(b.w, 4 * j4 + 3, true);
    }
    for (; i4 < n4; i4 += stride) {                           // (lanes leave these loops at different trips: the ballots above see the active ones)
        const f32x4 a = dist4[i4];
        count_one(a.x, 4 * i4, true); count_one(a.y, 4 * i4 + 1, true); count_one(a.z, 4 * i4 + 2, true); count_one(a.w, 4 * i4 + 3, true);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {                // the last n % 4 distances
        const uint32_t i = (n4 << 2) + threadIdx.x;
        const bool live = i < n;
        count_one(live ? dist[i] : 0.f, i, live);
    }
    __syncthreads();
    // last arriver picks. The histogram is only ever touched by device-scope atomics (they execute where every XCD sees them). The adds
    // are RETURNING ones and their results are consumed (stored to LDS) before the barrier: a result in a register is the only proof
    // that the add has been performed — a fire-and-forget add could still be on its way when the last arriver empties the bins. (No
    // fence: an agent-scope release writes back the XCD's L2, and 2 048 workgroups doing so one after the other cost 200 - 450 us per
    // pass: profiles/r06/e_general_selection_*.)
    uint32_t seen = 0;
    if (h[threadIdx.x]) seen = __hip_atomic_fetch_add(&hist[threadIdx.x], h[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    part[threadIdx.x] = seen;
    __syncthreads();
    if (threadIdx.x == 0) last_s = __hip_atomic_fetch_add(&hist[SEL_BINS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last_s == 0u) return;
    // (a returning atomic: the value comes from wherever device-scope atomics execute, never from a stale line of this XCD's L2; the
    // exchange also re-arms the bin for the next pass)
    const uint32_t mine = __hip_atomic_exchange(&hist[threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {                                    // exclusive scan of the 256 bins
        uint32_t run = 0;
        for (int t = 0; t < SEL_BINS; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    const uint32_t cum = part[threadIdx.x];
    if ((uint64_t)cum < rem && (uint64_t)cum + mine >= rem) {   // exactly one thread: its bin holds the k-th key
        const uint64_t np = (prefix << 8) | (uint64_t)threadIdx.x;
        const uint64_t left = rem - cum;                       // rank inside the bin, 1 .. mine
        const bool all_needed = left == (uint64_t)mine;        // every key of the bin is among the k smallest: no finer digit matters
        if (pass == SEL_PASSES - 1 || all_needed) {
            state[0] = shift == 0 ? np : ((np << shift) | ((1ull << shift) - 1ull));
            state[2] = 1;
        } else {
            state[0] = np;
            state[2] = 0;
        }
        state[1] = left;
        if (first) *counter = 0;
    }
    if (threadIdx.x == 0) (void)__hip_atomic_exchange(&hist[SEL_BINS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void select_compact_kernel(const float* __restrict__ dist, uint32_t n,
                                                             uint32_t row_base, const uint64_t* __restrict__ state,
                                                             uint32_t* counter, int64_t* __restrict__ out,
                                                             uint32_t kmax, const uint32_t* __restrict__ gate) {
    if (gate != nullptr && *gate == 0u) return;
    const uint64_t thr = state[0];  // the exact k-th smallest unsigned key (or the largest key of its tie group when all of it is needed)
    auto take = [&](float d, uint32_t i) {
        const uint64_t u = ukey_of(d, row_base + i);
        if (u <= thr) {
            const uint32_t pos = atomicAdd(counter, 1u);
            if (pos < kmax) out[pos] = (int64_t)(u ^ 0x8000000000000000ull);
        }
    };
    const uint32_t n4 = n >> 2, stride = gridDim.x * 256;
    const f32x4* __restrict__ dist4 = reinterpret_cast<const f32x4*>(dist);
    uint32_t i4 = blockIdx.x * 256 + threadIdx.x;
    for (; i4 + stride < n4; i4 += 2 * stride) {              // 16-byte loads, two in flight (see select_hist_kernel)
        const f32x4 a = dist4[i4], b = dist4[i4 + stride];
        const uint32_t j4 = i4 + stride;
        take(a.x, 4 * i4); take(a.y, 4 * i4 + 1); take(a.z, 4 * i4 + 2); take(a.w, 4 * i4 + 3);
        take(b.x, 4 * j4); take(b.y, 4 * j4 + 1); take(b.z, 4 * j4 + 2); take(b.w, 4 * j4 + 3);
    }
    for (; i4 < n4; i4 += stride) {
        const f32x4 a = dist4[i4];
        take(a.x, 4 * i4); take(a.y, 4 * i4 + 1); take(a.z, 4 * i4 + 2); take(a.w, 4 * i4 + 3);
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const uint32_t i = (n4 << 2) + threadIdx.x;
        if (i < n) take(dist[i], i);
    }
}

__global__ __launch_bounds__(256) void rank_sort_kernel(const int64_t* __restrict__ in, int k,
                                                        int64_t* __restrict__ out, const uint32_t* __restrict__ gate) {
    __shared__ int64_t tile[256];
    if (gate != nullptr && *gate == 0u) return;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    const int64_t mine = (i < k) ? in[i] : KEY_PAD;
    int rank = 0;
    for (int base = 0; base < k; base += 256) {
        const int j = base + (int)threadIdx.x;
        tile[threadIdx.x] = (j < k) ? in[j] : KEY_PAD;
        __syncthreads();
        const int lim = (k - base) < 256 ? (k - base) : 256;
        for (int t = 0; t < lim; ++t) rank += (tile[t] < mine) ? 1 : 0;
        __syncthreads();
    }
    if (i < k) out[rank] = mine;
}

__global__ __launch_bounds__(256) void keys_to_hits_kernel(const int64_t* __restrict__ keys, int k, int kpad,
                                                           const uint64_t* __restrict__ ids, uint32_t row_base,
                                                           uint32_t n_rows, wax_hip_hit* __restrict__ out,
                                                           const uint32_t* __restrict__ gate) {
    const int t = (int)(blockIdx.x * 256 + threadIdx.x);
    if (t >= kpad) return;
    if (gate != nullptr && *gate == 0u) return;
    wax_hip_hit h;
    h.key = (t < k) ? keys[t] : KEY_PAD;
    h.frame_id = ID_PAD;
    if (h.key != KEY_PAD) {
        const uint32_t local = key_row(h.key) - row_base;
        h.frame_id = (ids != nullptr && local < n_rows) ? ids[local] : (uint64_t)key_row(h.key);
    }
    out[t] = h;
}

hipError_t launch_select_general(const float* d_dist, uint32_t n_rows, uint32_t row_base, int k, int kpad,
                                 const uint64_t* d_ids, const SelectWork& w, wax_hip_hit* d_out, hipStream_t st, const uint32_t* gate) {
    if (k < 1 || (uint32_t)k > n_rows || k > WAX_HIP_MAX_RESULTS || kpad < k) return hipErrorInvalidValue;
    int grid = (int)((n_rows + 255) / 256);
    if (grid > 2048) grid = 2048;
    for (int pass = 0; pass < SEL_PASSES; ++pass)
        hipLaunchKernelGGL(select_hist_kernel, dim3(grid), dim3(256), 0, st, d_dist, n_rows, row_base, pass, (uint32_t)k, w.state,
                           w.hist, w.counter, gate);
    hipLaunchKernelGGL(select_compact_kernel, dim3(grid), dim3(256), 0, st, d_dist, n_rows, row_base, w.state,
                       w.counter, w.keys_a, (uint32_t)k, gate);
    hipLaunchKernelGGL(rank_sort_kernel, dim3((k + 255) / 256), dim3(256), 0, st, w.keys_a, k, w.keys_b, gate);
    hipLaunchKernelGGL(keys_to_hits_kernel, dim3((kpad + 255) / 256), dim3(256), 0, st, w.keys_b, k, kpad, d_ids,
                       row_base, n_rows, d_out, gate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Short selection (kernels.h: launch_select_short). One workgroup; the candidate lists are L2-resident (the scan in front of it
// wrote them) and ascending. Steps:
//   (1) S = the first p entries of every list (p = about twice the k / lists a list holds of the answer on average), sorted in LDS:
//       its k-th smallest, U, is an upper bound of the k-th smallest candidate (S is a subset with at least k members) and a tight one —
//       on a store whose best rows are spread over the workgroups it lies a few per cent of the candidates above the true k-th;
//       unlike "the largest head" it stays tight when some workgroups hold no good row at all (sorted, clustered corpora);
//   (2) every list's prefix <= U into LDS, eight independent loads at a time (a thread walking its list one load after the other
//       would wait for L2 once per key): at least k keys, usually k + a few per cent;
//   (3) bitonic sort in LDS: the first k are the answer;
//   (4) certificate (per_list < k only): a full list dropped only keys above its last entry;
//   (5) hits.
// More prefix keys than the LDS buffer holds (the k best bunched in a few lists, deeper than p), fewer than k live candidates: the
// launch fails (*flags = 1, nothing written) and the gated path behind it answers.
constexpr int SHORT_THREADS = 1024;
constexpr uint32_t SHORT_CAP = 16384;            // LDS buffer, keys (128 KB)
// Bitonic sort of n_pow2 >= 2048 keys in LDS by SHORT_THREADS threads. A wave owns n_pow2 / 16 consecutive keys: every step whose
// stride stays inside that segment needs no workgroup barrier, only the wave's own LDS order (at 2 048 keys 56 of the 66 steps) —
// with one compare-exchange per thread and step the barriers were most of the kernel.
__device__ inline void short_sort(int64_t* buf, uint32_t n_pow2, uint32_t tid) {
    const uint32_t seg = n_pow2 / (SHORT_THREADS / 64);          // keys per wave, a power of two >= 128
    const uint32_t wave = tid >> 6, lane = tid & 63u, base = wave * seg;
    for (uint32_t size = 2; size <= n_pow2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride < seg) {                                    // inside the wave's own segment
                for (uint32_t j = lane; j < (seg >> 1); j += 64) {
                    const uint32_t lo = base + ((j / stride) * (stride << 1)) + (j % stride), hi = lo + stride;
                    const int64_t a = buf[lo], b = buf[hi];
                    const bool up = (lo & size) == 0u;
                    if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
                }
                wave_lds_fence();
            } else {
                for (uint32_t i = tid; i < (n_pow2 >> 1); i += SHORT_THREADS) {
                    const uint32_t lo = ((i / stride) * (stride << 1)) + (i % stride), hi = lo + stride;
                    const int64_t a = buf[lo], b = buf[hi];
                    const bool up = (lo & size) == 0u;
                    if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
                }
                __syncthreads();
            }
        }
        if (size >= seg) __syncthreads();                          // the next size starts with a stride that leaves the segments
    }
    __syncthreads();
}
__host__ __device__ inline uint32_t short_depth(uint32_t k, uint32_t lists, uint32_t per_list) {
    uint32_t p = (2u * k + lists - 1) / lists;
    p = p < 4u ? 4u : p;
    return p > per_list ? per_list : p;
}
__global__ __launch_bounds__(SHORT_THREADS) void select_short_kernel(const int64_t* __restrict__ cand, uint32_t lists, uint32_t per_list,
                                                                     uint32_t k, uint32_t kpad, uint32_t cap,
                                                                     const uint64_t* __restrict__ ids, uint32_t row_base, uint32_t n_rows,
                                                                     wax_hip_hit* __restrict__ out, uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) int64_t buf[];   // [cap] (a power of two, >= lists * p, <= SHORT_CAP)
    __shared__ uint32_t s_fill, s_fail;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_fail = 0u; s_fill = 0u; }
    // (1)
    const uint32_t p = short_depth(k, lists, per_list), ns = lists * p;   // ns <= cap (launch_select_short)
    uint32_t sn = 2048;
    while (sn < ns) sn <<= 1;
    for (uint32_t i = tid; i < sn; i += SHORT_THREADS) buf[i] = i < ns ? cand[(size_t)(i / p) * per_list + (i % p)] : KEY_PAD;
    __syncthreads();
    short_sort(buf, sn, tid);
    const int64_t bound = k <= ns ? buf[k - 1] : KEY_PAD;      // KEY_PAD: fewer than k live entries in S — every live candidate is taken
    __syncthreads();
    // (2)
    for (uint32_t w = tid; w < lists; w += SHORT_THREADS) {
        const int64_t* __restrict__ lw = cand + (size_t)w * per_list;
        bool more = true;
        for (uint32_t base = 0; base < per_list && more; base += 8) {
            int64_t v[8];
#pragma unroll
            for (uint32_t t = 0; t < 8; ++t) v[t] = base + t < per_list ? lw[base + t] : KEY_PAD;
#pragma unroll
            for (uint32_t t = 0; t < 8; ++t) {
                if (more && v[t] != KEY_PAD && v[t] <= bound) {
                    const uint32_t pos = atomicAdd(&s_fill, 1u);
                    if (pos < cap) buf[pos] = v[t];
                } else {
                    more = false;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t c = s_fill;
    if ((c < k || c > cap) && tid == 0) s_fail = 1u;
    uint32_t sort_n = 2048;
    while (sort_n < c && sort_n < cap) sort_n <<= 1;
    for (uint32_t i = c + tid; i < sort_n; i += SHORT_THREADS) buf[i] = KEY_PAD;
    __syncthreads();
    if (s_fail == 0u) {
        short_sort(buf, sort_n, tid);                          // (3)
        if (per_list < k) {                                    // (4)
            const int64_t kth = buf[k - 1];
            uint32_t bad = 0;
            for (uint32_t w = tid; w < lists; w += SHORT_THREADS) {
                const int64_t last = cand[(size_t)w * per_list + per_list - 1];
                if (last != KEY_PAD && last < kth) bad = 1u;
            }
            if (bad) s_fail = 1u;
        }
        __syncthreads();
    }
    const bool fail = s_fail != 0u;
    if (!fail) {
        for (uint32_t t = tid; t < kpad; t += SHORT_THREADS) {
            wax_hip_hit hit;
            hit.key = t < k ? buf[t] : KEY_PAD;
            hit.frame_id = ID_PAD;
            if (hit.key != KEY_PAD) {
                const uint32_t local = key_row(hit.key) - row_base;
                hit.frame_id = (ids != nullptr && local < n_rows) ? ids[local] : (uint64_t)key_row(hit.key);
            }
            out[t] = hit;
        }
    }
    if (tid == 0) { flags[0] = fail ? 1u : 0u; flags[1] += fail ? 1u : 0u; flags[2] += 1u; }
}

// whether the short selection is worth trying: S must fit the LDS buffer, and for k > per_list the lists must be able to hold the
// answer with room to spare (k <= a third of a list per list on average)
bool select_short_viable(int k, int lists, int per_list) {
    if (lists <= 0 || k < 1 || per_list < 1) return false;
    if ((uint64_t)lists * short_depth((uint32_t)k, (uint32_t)lists, (uint32_t)per_list) > SHORT_CAP) return false;
    return k <= per_list || (int64_t)k * 3 <= (int64_t)lists * per_list;
}

hipError_t launch_select_short(const int64_t* d_cand, uint32_t lists, uint32_t per_list, int k, int kpad, const uint64_t* d_ids,
                               uint32_t row_base, uint32_t n_rows, uint32_t* d_flags, wax_hip_hit* d_out, hipStream_t st) {
    if (k < 1 || k > WAX_HIP_MAX_RESULTS || kpad < k || d_flags == nullptr || !select_short_viable(k, (int)lists, (int)per_list))
        return hipErrorInvalidValue;
    // LDS buffer: room for S and for four times the answer (prefixes are the answer plus a few per cent on ordinary stores), a power of
    // two, at most 128 KB — a small one finds a CU beside the filtering GEMM's workgroups (100 KB at 384-d) when batches are in flight;
    // prefixes that overflow it leave the answer to the gated launches behind. (Dynamic LDS beyond 64 KB: like the GEMM's tile
    // buffers, no attribute needed on this runtime.)
    const uint32_t ns = lists * short_depth((uint32_t)k, lists, per_list);
    uint32_t cap = 2048;
    while (cap < SHORT_CAP && (cap < ns || cap < 4u * (uint32_t)k)) cap <<= 1;
    hipLaunchKernelGGL(select_short_kernel, dim3(1), dim3(SHORT_THREADS), (size_t)cap * sizeof(int64_t), st, d_cand, lists, per_list,
                       (uint32_t)k, (uint32_t)kpad, cap, d_ids, row_base, n_rows, d_out, d_flags);
    return hipGetLastError();
}

hipError_t alloc_select_work(SelectWork* w) {
    hipError_t e = hipMalloc(&w->hist, (SEL_BINS + 1) * sizeof(uint32_t));           // bins + the arrival ticket
    if (e == hipSuccess) e = hipMemset(w->hist, 0, (SEL_BINS + 1) * sizeof(uint32_t));   // every pass leaves them zero again
    if (e == hipSuccess) e = hipMalloc(&w->state, 8 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(w->state, 0, 8 * sizeof(uint64_t));
    if (e == hipSuccess) w->flags = reinterpret_cast<uint32_t*>(w->state + 4);
    if (e == hipSuccess) e = hipMalloc(&w->counter, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&w->keys_a, (size_t)WAX_HIP_MAX_RESULTS * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&w->keys_b, (size_t)WAX_HIP_MAX_RESULTS * sizeof(int64_t));
    // hipMemset on device memory may return before the fill has run, and the engine's streams are not ordered behind the null stream:
    // without this wait the first selection of a fresh workspace can meet a histogram that still holds whatever the allocation held
    // (seen: wrong first answers of a sharded handle's shards, whose workspaces reuse freed memory).
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}
void free_select_work(SelectWork* w) {
    (void)hipFree(w->hist); (void)hipFree(w->state); (void)hipFree(w->counter); (void)hipFree(w->keys_a); (void)hipFree(w->keys_b);
    *w = SelectWork{};
}

// ---------------------------------------------------------------------------
// Streaming-read microbenchmark (roofline denominator measured on the node itself).
template <bool NT>
__global__ __launch_bounds__(256) void stream_read_kernel(const f32x4* __restrict__ src, uint64_t n16,
                                                          float* __restrict__ sink) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    uint64_t i = (uint64_t)blockIdx.x * 256 * 4 + threadIdx.x;
    for (; i + 3 * 256 < n16; i += stride) {
        const f32x4 a = ld16<NT>(src + i), b = ld16<NT>(src + i + 256), c = ld16<NT>(src + i + 512),
                    d = ld16<NT>(src + i + 768);
        acc += (a + b) + (c + d);
    }
    for (; i < n16; i += 256) acc += ld16<NT>(src + i);
    const float s = hsum(acc);
    if (s == 123.456f) sink[blockIdx.x] = s;  // practically never true; keeps the loads live
}

hipError_t launch_stream_read(const float* d_src, uint64_t bytes, int nt, int grid, float* d_sink, hipStream_t st) {
    const uint64_t n16 = bytes / 16;
    if (grid <= 0) grid = 2048;
    if (nt)
        hipLaunchKernelGGL((stream_read_kernel<true>), dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const f32x4*>(d_src), n16, d_sink);
    else
        hipLaunchKernelGGL((stream_read_kernel<false>), dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const f32x4*>(d_src), n16, d_sink);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
hipError_t device_shift_down(void* base, uint64_t dst_off, uint64_t src_off, uint64_t bytes, void* bounce,
                             uint64_t bounce_bytes, hipStream_t st) {
    // Regions overlap (dst < src): move ascending in bounce-sized pieces; piece c's destination only
    // overlaps source bytes of pieces <= c, which have already been consumed.
    char* b = static_cast<char*>(base);
    uint64_t done = 0;
    while (done < bytes) {
        const uint64_t len = (bytes - done) < bounce_bytes ? (bytes - done) : bounce_bytes;
        hipError_t e = hipMemcpyAsync(bounce, b + src_off + done, len, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(b + dst_off + done, bounce, len, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
        done += len;
    }
    return hipSuccess;
}

}  // namespace wax
