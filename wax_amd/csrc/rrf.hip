// rrf.hip — weighted reciprocal-rank fusion of ranked id lists on the device, one workgroup per query.
//
// Reference: HybridSearch.rrfFusion(lists:k:) (HybridSearch.swift:25-52) == UnifiedSearch.rrfFusionResults
// (UnifiedSearch.swift:590-699) minus the diagnostics. Lists ("lanes": text, vector, timeline, …) in order; a lane with
// weight <= 0 is skipped; entry at 0-based position p adds weight / Float(max(0, k) + p + 1) to its frame's f32 score, in
// lane order then position order; bestRank = the smallest p + 1 over the lanes; result = every frame seen, sorted by
// (score desc, bestRank asc, frameId asc).
//
// A batched-query service keeps the vector lane's hits in HBM (wax_hip_search_batch_hits_device) and only the fused
// top-n comes back. Device shape: an open-addressing table in LDS (6 144 slots: at most 4 096 entries per query),
//   * lanes strictly one after another (a barrier between them) and, inside a lane, positions in chunks of 256 with a
//     per-slot claim (atomicMin of the position) so that two occurrences of one id in the same chunk are added in
//     position order: every frame's f32 additions happen in exactly the reference's order => bit-identical scores;
//   * a bitonic sort of slot indices under the reference's comparator.
#include "kernels.h"

namespace wax {

constexpr uint32_t RRF_SLOTS = 6144;
constexpr uint64_t RRF_EMPTY = 0xffffffffffffffffull;   // also ID_PAD: padding entries of a hit list are skipped

struct RrfLaneDev {
    const uint64_t* ids;      // entry p of query q: ids[(q * stride + p) * pitch]
    const uint32_t* counts;   // per query, or null = stride entries
    uint32_t stride, pitch;
    float weight;
};
struct RrfArgs {
    RrfLaneDev lane[WAX_HIP_RRF_MAX_LANES];
    uint32_t n_lanes, nq, out_stride;
    int32_t k;
    uint64_t* out_ids; float* out_scores; uint32_t* out_best_rank; uint32_t* out_sources; uint32_t* out_counts;
};

__device__ __forceinline__ uint32_t rrf_hash(uint64_t id) {
    uint64_t x = id;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return (uint32_t)(((x & 0xffffffffull) * (uint64_t)RRF_SLOTS) >> 32);
}

__global__ __launch_bounds__(256) void rrf_fuse_kernel(RrfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);                    // [SLOTS]
    float* score = reinterpret_cast<float*>(keys + RRF_SLOTS);                                // [SLOTS]
    unsigned int* meta = reinterpret_cast<unsigned int*>(score + RRF_SLOTS);                   // bestRank | sources << 16
    unsigned int* claim = meta + RRF_SLOTS;                                                    // [SLOTS]
    unsigned short* order = reinterpret_cast<unsigned short*>(claim + RRF_SLOTS);              // [4096] slot indices
    __shared__ unsigned int n_occ;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    for (uint32_t s = tid; s < RRF_SLOTS; s += 256u) { keys[s] = RRF_EMPTY; score[s] = 0.f; meta[s] = 0xffffu; claim[s] = 0xffffffffu; }
    if (tid == 0) n_occ = 0u;
    __syncthreads();
    const int32_t kc = a.k > 0 ? a.k : 0;
    for (uint32_t l = 0; l < a.n_lanes; ++l) {
        const RrfLaneDev L = a.lane[l];
        if (!(L.weight > 0.0f)) continue;                    // wave-uniform (kernel argument)
        const uint32_t cnt = L.counts ? (L.counts[q] < L.stride ? L.counts[q] : L.stride) : L.stride;
        const uint64_t* base = L.ids + (size_t)q * L.stride * L.pitch;
        for (uint32_t p0 = 0; p0 < cnt; p0 += 256u) {
            const uint32_t p = p0 + tid;
            uint64_t id = RRF_EMPTY;
            if (p < cnt) id = base[(size_t)p * L.pitch];
            bool pending = id != RRF_EMPTY;
            uint32_t slot = 0;
            if (pending) {                                   // find or insert
                slot = rrf_hash(id);
                for (;;) {
                    const unsigned long long prev = atomicCAS(&keys[slot], RRF_EMPTY, (unsigned long long)id);
                    if (prev == RRF_EMPTY || prev == id) break;
                    slot = slot + 1u == RRF_SLOTS ? 0u : slot + 1u;
                }
            }
            const float contribution = L.weight / (float)(kc + (int32_t)p + 1);
            // duplicates of one id inside this chunk take turns in position order
            for (;;) {
                const unsigned int ticket = p - p0;          // claim[] is 0xffffffff whenever no turn is being decided
                if (pending) atomicMin(&claim[slot], ticket);
                __syncthreads();
                if (pending && claim[slot] == ticket) {
                    score[slot] += contribution;
                    const unsigned int m = meta[slot];
                    const unsigned int br = (m & 0xffffu) < p + 1u ? (m & 0xffffu) : p + 1u;
                    meta[slot] = br | (m & 0xffff0000u) | (1u << (16u + l));
                    claim[slot] = 0xffffffffu;               // the next occurrence (if any) claims again
                    pending = false;
                }
                if (!__syncthreads_or(pending ? 1 : 0)) break;
            }
        }
    }
    __syncthreads();
    // occupied slots -> order[], padded to a power of two with an "after everything" marker
    for (uint32_t s = tid; s < RRF_SLOTS; s += 256u)
        if (keys[s] != RRF_EMPTY) order[atomicAdd(&n_occ, 1u)] = (unsigned short)s;
    __syncthreads();
    const uint32_t m = n_occ;
    uint32_t mp = 1;
    while (mp < m) mp <<= 1;
    for (uint32_t i = m + tid; i < mp; i += 256u) order[i] = 0xffffu;
    __syncthreads();
    auto before = [&](unsigned short x, unsigned short y) -> bool {   // the reference's comparator (:44-50, :661-665)
        if (y == 0xffffu) return x != 0xffffu;
        if (x == 0xffffu) return false;
        const float sx = score[x], sy = score[y];
        if (sx != sy) return sx > sy;
        const unsigned int rx = meta[x] & 0xffffu, ry = meta[y] & 0xffffu;
        if (rx != ry) return rx < ry;
        return keys[x] < keys[y];
    };
    for (uint32_t size = 2; size <= mp; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = tid; i < (mp >> 1); i += 256u) {
                const uint32_t lo = 2u * i - (i & (stride - 1u));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0u;
                const unsigned short x = order[lo], y = order[hi];
                if (up ? before(y, x) : before(x, y)) { order[lo] = y; order[hi] = x; }
            }
            __syncthreads();
        }
    }
    const uint32_t n_out = m < a.out_stride ? m : a.out_stride;
    for (uint32_t i = tid; i < a.out_stride; i += 256u) {
        const size_t o = (size_t)q * a.out_stride + i;
        if (i < n_out) {
            const unsigned short s = order[i];
            a.out_ids[o] = keys[s];
            a.out_scores[o] = score[s];
            if (a.out_best_rank) a.out_best_rank[o] = meta[s] & 0xffffu;
            if (a.out_sources) a.out_sources[o] = meta[s] >> 16;
        } else {
            a.out_ids[o] = RRF_EMPTY;
            a.out_scores[o] = 0.f;
            if (a.out_best_rank) a.out_best_rank[o] = 0u;
            if (a.out_sources) a.out_sources[o] = 0u;
        }
    }
    if (tid == 0 && a.out_counts) a.out_counts[q] = n_out;
}

hipError_t launch_rrf_fuse(const wax_hip_rrf_lane* lanes, uint32_t n_lanes, uint32_t nq, int32_t k, uint64_t* out_ids,
                           float* out_scores, uint32_t* out_best_rank, uint32_t* out_sources, uint32_t out_stride,
                           uint32_t* out_counts, hipStream_t st) {
    RrfArgs a{};
    for (uint32_t l = 0; l < n_lanes; ++l) {
        a.lane[l].ids = lanes[l].d_ids; a.lane[l].counts = lanes[l].d_counts; a.lane[l].stride = lanes[l].stride;
        a.lane[l].pitch = lanes[l].pitch ? lanes[l].pitch : 1u; a.lane[l].weight = lanes[l].weight;
    }
    a.n_lanes = n_lanes; a.nq = nq; a.out_stride = out_stride; a.k = k;
    a.out_ids = out_ids; a.out_scores = out_scores; a.out_best_rank = out_best_rank; a.out_sources = out_sources; a.out_counts = out_counts;
    constexpr size_t smem = (size_t)RRF_SLOTS * (8 + 4 + 4 + 4) + 4096 * 2;
    static std::atomic<uint64_t> attr_set{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&rrf_fuse_kernel), smem, attr_set);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rrf_fuse_kernel, dim3(nq), dim3(256), smem, st, a);
    return hipGetLastError();
}

}  // namespace wax
