// topk.h — wavefront-private streaming top-k and workgroup rank-merge (device only).
//
// Replaces the reference's multi-pass GPU selection (TopKReduction.metal:103-167:
// per-256-chunk lane-0 heap or bitonic sort, iterated with one launch and one
// fresh buffer per pass, MetalVectorEngine.swift:511-575) with a design that
// fits CDNA4: every 64-lane wave keeps a private candidate list in LDS behind a
// running threshold `tau` (the wave's current k-th smallest key), so that after
// warm-up almost every row is rejected with one 64-bit compare and the
// selection costs ~0 next to the HBM stream. Keys are unique 64-bit integers
// (ordered distance : row), so order is total and deterministic.
#pragma once
#include "common.h"

namespace wax {

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

// Compiler-level ordering point for LDS traffic inside ONE wave. A wave's DS
// instructions execute in order, so no hardware wait is needed between a
// ds_write and a later ds_read of another lane's data; this only stops hipcc
// from reordering across the point.
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef __attribute__((address_space(3))) int64_t lds_i64;

struct PruneOut {
    int cnt;
    int64_t tau;
};

// Rank-sort the `cnt` live candidates of one wave's list, keep the k smallest in
// buf[0..min(cnt,k)) ascending, return the new count and threshold. O(cnt*CAP/64) compares per
// lane, no inter-lane shuffles. Deliberately NOT inlined: it is cold code (a wave prunes a handful
// of times per launch) and every inlined copy is ~3 KB of instructions that a single-workgroup
// kernel pays for in instruction-cache misses (merge_keys_kernel: 5 500 lines of ISA, 20 us).
// UNIQUE: the caller guarantees that no key occurs twice (keys carry the global row in their low half, and every
// internal producer offers a row once), so a rank is just the number of smaller keys: one 64-bit compare per pair
// instead of three compares. Lists of foreign keys (gathered shard hits) use UNIQUE = false, which orders duplicates
// by slot index. Slices of 64 slots beyond the live count are skipped (wave-uniform).
template <int CAP, bool UNIQUE>
__device__ __attribute__((noinline)) PruneOut wave_prune(lds_i64* buf, int cnt, int k) {
    constexpr int E = CAP / 64;
    wave_lds_fence();
    const int lane = lane_id();
    int64_t mine[E];
    int rank[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane + 64 * e;
        mine[e] = (idx < cnt) ? buf[idx] : KEY_PAD;
        rank[e] = 0;
    }
    const int live = __builtin_amdgcn_readfirstlane(cnt);
    // 4 broadcast keys per trip (two ds_read_b128): the loop is LDS-latency-bound otherwise.
    int j = 0;
    for (; j + 4 <= live; j += 4) {
        int64_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = buf[j + u];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (64 * e < live) {
                const int idx = lane + 64 * e;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (UNIQUE) rank[e] += (o[u] < mine[e]) ? 1 : 0;
                    else rank[e] += (o[u] < mine[e] || (o[u] == mine[e] && (j + u) < idx)) ? 1 : 0;
                }
            }
        }
    }
    for (; j < live; ++j) {
        const int64_t o = buf[j];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = lane + 64 * e;
            if (UNIQUE) rank[e] += (o < mine[e]) ? 1 : 0;
            else rank[e] += (o < mine[e] || (o == mine[e] && j < idx)) ? 1 : 0;
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane + 64 * e;
        if (idx < cnt && rank[e] < k) buf[rank[e]] = mine[e];
    }
    wave_lds_fence();
    PruneOut r;
    r.cnt = cnt < k ? cnt : k;
    r.tau = (r.cnt >= k) ? buf[k - 1] : KEY_PAD;
    return r;
}

// Streaming k-smallest over int64 keys, private to one wave.
//   CAP  : LDS slots (power of two, >= k + 64 so that one push of up to 64
//          candidates always fits after a prune)
template <int CAP, bool UNIQUE = true>
struct WaveTopK {
    static_assert(CAP % 64 == 0, "CAP must be a multiple of the wave size");

    int64_t* buf;  // CAP slots in LDS, this wave only
    int cnt;       // wave-uniform number of live candidates in buf[0..cnt)
    int k;         // wave-uniform
    int64_t tau;   // wave-uniform: only keys < tau can still enter the top-k

    __device__ inline void init(int64_t* lds, int k_) {
        buf = lds;
        cnt = 0;
        k = k_;
        tau = KEY_PAD;
    }

    // Every lane of the wave must call push() together (valid=false for idle lanes). The caller
    // guarantees room: cnt + (number of valid lanes) <= CAP, by calling make_room() first.
    __device__ inline void push(int64_t key, bool valid) {
        const bool pass = valid && (key < tau);
        const unsigned long long mask = __ballot(pass);
        if (mask == 0ull) return;  // wave-uniform
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (pass) buf[cnt + before] = key;
        cnt += __popcll(mask);
    }

    // Ensure the next `incoming` (<= 64, or <= CAP - k) candidates fit; prunes at most once.
    __device__ inline void make_room(int incoming) {
        if (cnt > CAP - incoming) prune();
    }

    // push() for kernels that offer up to 64 candidates at once: prunes only when the candidates
    // that actually pass the threshold do not fit (a fixed make_room(64) would prune after every
    // insertion once k >= CAP - 64).
    __device__ inline void push_wide(int64_t key, bool valid) {
        bool pass = valid && (key < tau);
        unsigned long long mask = __ballot(pass);
        if (mask == 0ull) return;
        if (cnt + __popcll(mask) > CAP) {
            prune();  // cnt <= k <= CAP - 64 afterwards, tau tightened: re-test
            pass = valid && (key < tau);
            mask = __ballot(pass);
            if (mask == 0ull) return;
        }
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (pass) buf[cnt + before] = key;
        cnt += __popcll(mask);
    }

    __device__ inline void prune() {
        const PruneOut r = wave_prune<CAP, UNIQUE>((lds_i64*)buf, cnt, k);
        cnt = r.cnt;
        tau = r.tau;
    }

    __device__ inline void finalize() { prune(); }
};

// Merge the sorted per-wave lists of a workgroup by ranking: an entry's final
// position is its own index plus, for every other wave, the number of that
// wave's entries ordered before it (binary search). One pass, one barrier.
//   lists  : LDS, wave w's sorted list at lists + w*stride, length counts[w] (<= k)
//   out    : k slots (global or LDS); slots past the merged total get KEY_PAD
// Must be called by all threads of the block; ends with all writes issued (no trailing barrier).
typedef __attribute__((address_space(3))) int lds_i32;

// Not inlined and not unrolled for the same instruction-footprint reason as wave_prune.
__device__ __attribute__((noinline)) void block_rank_merge_impl(const lds_i64* lists, int nwaves, int stride,
                                                                const lds_i32* counts, int k, int64_t* out) {
    const int tid = (int)threadIdx.x;
    const int nthreads = (int)blockDim.x;
    int total = 0;
    for (int w = 0; w < nwaves; ++w) total += counts[w];
    const int total_k = total < k ? total : k;
    for (int t = tid; t < nwaves * k; t += nthreads) {
        const int w = t / k, i = t - w * k;
        if (i >= counts[w]) continue;
        const int64_t key = lists[w * stride + i];
        int rank = i;
        for (int o = 0; o < nwaves; ++o) {
            if (o == w) continue;
            // number of entries in list o that precede `key` (ties: lower wave index first)
            int lo = 0, hi = counts[o];
            const lds_i64* lst = lists + o * stride;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int64_t v = lst[mid];
                const bool before = (o < w) ? (v <= key) : (v < key);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) out[rank] = key;
    }
    for (int t = total_k + tid; t < k; t += nthreads) out[t] = KEY_PAD;
}

template <int NWAVES>
__device__ inline void block_rank_merge(const int64_t* lists, int stride, const int* counts, int k,
                                        int64_t* out) {
    block_rank_merge_impl((const lds_i64*)lists, NWAVES, stride, (const lds_i32*)counts, k, out);
}

// ---- DPP cross-lane adds (no LDS traffic) ---------------------------------
// v + (v moved by a DPP pattern); lanes in rows excluded by ROW_MASK add +0.0.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}

// Sum over each aligned group of GROUP lanes. The total is valid in the LAST
// lane of each group (for GROUP <= 16 in every lane of the group).
template <int GROUP>
__device__ inline float group_sum(float v) {
    static_assert(GROUP == 4 || GROUP == 8 || GROUP == 16 || GROUP == 32 || GROUP == 64, "GROUP");
    v = dpp_add<0xB1, 0xF>(v);                         // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);                         // quad_perm [2,3,0,1]
    if (GROUP >= 8) v = dpp_add<0x141, 0xF>(v);        // row_half_mirror
    if (GROUP >= 16) v = dpp_add<0x140, 0xF>(v);       // row_mirror
    if (GROUP >= 32) v = dpp_add<0x142, 0xA>(v);       // row_bcast15 into rows 1,3
    if (GROUP >= 64) v = dpp_add<0x143, 0xC>(v);       // row_bcast31 into rows 2,3
    return v;
}

}  // namespace wax
