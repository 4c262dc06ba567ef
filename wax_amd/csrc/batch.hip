// batch.hip — batched queries: Q x D^T as a bf16 MFMA GEMM on the matrix cores, per-query
// candidate selection, exact f32 re-score, and an exactness certificate.
//
// The reference has no batched entry point (one `search(vector:topK:)` per query,
// MetalVectorEngine.swift:446-627); BASELINE.json configs 3 and 5 ask for this path because
// with Q >= 32 queries the scan is a genuine dense GEMM (arithmetic intensity Q flop per
// corpus byte) and belongs on MFMA, not on the HBM-bound VALU kernel.
//
// Pipeline (all on one stream; the corpus is walked in slabs whose size grows geometrically):
//   mirror_kernel        f32 store -> bf16 mirror (RNE; cosine rows pre-normalised), ||v||^2, max ||v||
//                        (once per mutation; the same kernel converts the query block)
//   GEMM over one slab   bf16 x bf16 -> f32 MFMA (v_mfma_f32_32x32x16_bf16) with the selection FUSED into the
//                        epilogue: an approx distance survives only if it beats its query's running threshold
//                        tau_q (the k'-th best approx distance over the slabs seen so far). The Q x N score matrix
//                        is never written: after the first 2K rows ~k' * slab/rows_so_far survivors per query per
//                        slab. Three kernels:
//     batch_gemm_rega_kernel    D in {128, 256, 384, 512}, cosine / dot: queries resident in VGPRs as A fragments,
//                               survivors into per-workgroup segments (no global atomics)
//     batch_gemm_ksplit_kernel  D = 768: the same with K split over the two waves of a SIMD
//     batch_gemm_kernel         everything else (D % 64 == 0, L2, the dense first slab): LDS-tiled 128 x 128
//   tighten_kernel       per query: best list + survivors -> best k' (sorted), tau_q tightened (between slabs)
//   rescore_kernel       exact f32 distance of every candidate, SAME lane mapping / summation order
//                        as scan_kernel => bit-identical to the single-query path
//   finalize_batch       sort by exact key, emit top-k hits + certificate:
//                        a non-candidate's approx distance >= a_max (the k'-th approx), so its exact
//                        distance >= a_max - eps (eps = rigorous bf16 rounding bound); if that is
//                        > the exact k-th best, the answer is provably the exact top-k. Otherwise (or if a
//                        candidate list overflowed) the host re-runs that query on the exact single-query path.
#include <type_traits>

#include "kernels.h"
#include "topk.h"

namespace wax {

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N) — the index is a constant expression inside
// f (inline-asm immediates need one; an unrolled loop variable is not).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { BM_COS = WAX_HIP_METRIC_COSINE, BM_DOT = WAX_HIP_METRIC_DOT, BM_L2 = WAX_HIP_METRIC_L2 };

__device__ inline unsigned short f32_to_bf16_rne(float x) {
    unsigned int b = __float_as_uint(x);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((b >> 16) | 0x0040u);  // quiet NaN
    b += 0x7fffu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}

// ---------------------------------------------------------------------------
// f32 rows -> bf16 rows (RNE). normalize=1 (cosine): each row is scaled by 1/||v|| first (0 if
// ||v|| <= 1e-6, CosineDistance.metal:323), so the GEMM epilogue is just d = 1 - acc.
// Also emits ||v||^2 (L2 epilogue) and the global max ||v|| (certificate bound for dot / L2).
// One wave per row; used for the corpus mirror and for the query block (rows in
// [n_rows, n_rows_padded) are zero-filled).
__global__ __launch_bounds__(256) void mirror_kernel(const float* __restrict__ src, uint32_t n_rows,
                                                     uint32_t n_rows_padded, uint32_t dims, int normalize,
                                                     unsigned short* __restrict__ dst, float* __restrict__ norm2,
                                                     unsigned int* __restrict__ max_norm_bits) {
    // max_norm_bits[0] = max ||v||; max_norm_bits[1] = max over the rows of ||x - bf16(x)||, x = the (scaled) f32 row that was
    // rounded: the row-side term of the cosine certificate bound, MEASURED instead of the worst case 2^-9 ||x|| (batch_prep_kernel)
    __shared__ unsigned int block_max, block_max_err;
    const int lane = lane_id();
    const uint32_t gwave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * 4;
    const bool vec4 = (dims & 3u) == 0;
    if (threadIdx.x == 0) { block_max = 0u; block_max_err = 0u; }
    __syncthreads();
    float wave_max = 0.f, wave_max_err = 0.f;  // one global atomic per workgroup: a per-row atomicMax serialises at ~11 ns each
    for (uint32_t r = gwave; r < n_rows_padded; r += nwaves) {
        unsigned short* out = dst + (size_t)r * dims;
        if (r >= n_rows) {
            for (uint32_t c = lane; c < dims; c += WAVE) out[c] = 0;
            if (lane == 0) norm2[r] = 0.f;
            continue;
        }
        const float* row = src + (size_t)r * dims;
        float acc = 0.f;
        if (vec4) {
            const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
            for (uint32_t c = lane; c < (dims >> 2); c += WAVE) {
                const f32x4 v = row4[c];
                acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
            }
        } else {
            for (uint32_t c = lane; c < dims; c += WAVE) acc = fmaf(row[c], row[c], acc);
        }
        acc = group_sum<64>(acc);
        acc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));  // the total lives in lane 63
        const float n = sqrtf(acc);
        const float scale = normalize ? ((n > 1e-6f) ? 1.0f / n : 0.0f) : 1.0f;
        float e2 = 0.f;        // ||x - bf16(x)||^2 of this row (x - bf16(x) is exact in f32: both are floats of one binade or neighbours)
        auto rnd = [&](float x) -> unsigned short {
            const unsigned short b = f32_to_bf16_rne(x);
            const float d = x - __uint_as_float((unsigned int)b << 16);
            e2 = fmaf(d, d, e2);
            return b;
        };
        if (vec4) {
            const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
            u16x4* out4 = reinterpret_cast<u16x4*>(out);
            for (uint32_t c = lane; c < (dims >> 2); c += WAVE) {
                const f32x4 v = row4[c];
                u16x4 o;
                o.x = rnd(v.x * scale); o.y = rnd(v.y * scale);
                o.z = rnd(v.z * scale); o.w = rnd(v.w * scale);
                out4[c] = o;
            }
        } else {
            for (uint32_t c = lane; c < dims; c += WAVE) out[c] = rnd(row[c] * scale);
        }
        e2 = group_sum<64>(e2);
        e2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e2), 63));
        const float err = sqrtf(e2);
        if (lane == 0) norm2[r] = acc;
        if (n == n && n > wave_max) wave_max = n;
        if (err == err && err > wave_max_err) wave_max_err = err;   // a NaN / inf row never passes a threshold and sorts last exactly
    }
    if (max_norm_bits != nullptr) {
        if (lane == 0) { atomicMax(&block_max, __float_as_uint(wave_max)); atomicMax(&block_max_err, __float_as_uint(wave_max_err)); }
        __syncthreads();
        if (threadIdx.x == 0 && block_max != 0u) atomicMax(max_norm_bits, block_max);
        if (threadIdx.x == 0 && block_max_err != 0u) atomicMax(max_norm_bits + 1, block_max_err);
    }
}

hipError_t launch_mirror(const float* src, uint32_t n_rows, uint32_t n_rows_padded, uint32_t dims, int normalize,
                         unsigned short* dst, float* norm2, unsigned int* max_norm_bits, hipStream_t st) {
    if (n_rows_padded == 0) return hipSuccess;
    uint64_t blocks = ((uint64_t)n_rows_padded + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mirror_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, n_rows, n_rows_padded, dims, normalize,
                       dst, norm2, max_norm_bits);
    return hipGetLastError();
}

// Maximum over each aligned group of 32 lanes, valid in the group's LAST lane (31 / 63). DPP only, like group_sum.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_max(float v) {
    // lanes in rows excluded by ROW_MASK receive `old` = their own value: max(v, v) = v
    const int moved = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return __builtin_fmaxf(v, __int_as_float(moved));
}
__device__ inline float group_max32(float v) {
    v = dpp_max<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v = dpp_max<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    v = dpp_max<0x141, 0xF>(v);     // row_half_mirror
    v = dpp_max<0x140, 0xF>(v);     // row_mirror
    v = dpp_max<0x142, 0xA>(v);     // row_bcast15 into rows 1, 3
    return v;
}

// ---------------------------------------------------------------------------
// bf16 GEMM tile: 128 queries (M) x 128 corpus rows (N), K chunks of 64, 4 waves each 64x64
// (2x2 v_mfma_f32_32x32x16_bf16 blocks). Both operands are K-contiguous ("NT" GEMM), so every
// MFMA fragment is one 16-byte read. LDS rows are padded 128 -> 144 B: ds_read_b128 is
// bank-conflict-free for the MFMA lane groups (MI355X_MICROARCH.md §LDS). The next K chunk is
// prefetched into registers while the current one is multiplied.
constexpr int GM = 128, GN = 128, GK = 64;
constexpr int LDS_STRIDE = GK + 8;  // bf16 elements per LDS row (144 bytes)

// SAMPLE = true (one-pass pipeline on this kernel: L2, and cosine / dot at the dimensions the register-resident kernels do
// not serve): corpus tile index i of the launch is the i-th of a.sample_tiles tiles spread evenly over the slab, and instead
// of filtering the workgroup records, per query, the best "similarity" of the tile in a.tile_max[i][query] — the dot
// product for cosine / dot, MINUS the distance for L2 (pick_tau_kernel turns the j-th best into the admission threshold).
template <int METRIC, bool SAMPLE = false>
__global__ __launch_bounds__(256) void batch_gemm_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short As[GM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[GN * LDS_STRIDE];

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const uint32_t qt = blockIdx.x % a.nqt;          // query tiles of one corpus tile are adjacent in launch order
    const uint32_t ct = blockIdx.x / a.nqt;
    const uint32_t m0 = qt * GM;
    const uint32_t ntiles_all = (a.slab_rows + GN - 1) / GN;
    const uint32_t ptile = SAMPLE ? (uint32_t)(((unsigned long long)ct * ntiles_all) / a.sample_tiles) : ct;
    const uint32_t n0 = a.slab0 + ptile * GN;
    const uint32_t D = a.dims;

    // staging map: 1024 16-byte segments per operand tile, 4 per thread; 8 consecutive threads
    // cover one 128-byte row chunk.
    const int seg = tid & 7;
    const int srow = tid >> 3;  // 0..31
    const u32x4* gA[4];
    const u32x4* gB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t qrow = m0 + srow + 32 * i;                       // always < nq_pad
        uint32_t crow = n0 + srow + 32 * i;
        crow = crow < a.n_rows ? crow : a.n_rows - 1;                    // clamp: masked at the store
        gA[i] = reinterpret_cast<const u32x4*>(a.qb + (size_t)qrow * D) + seg;
        gB[i] = reinterpret_cast<const u32x4*>(a.cb + (size_t)crow * D) + seg;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = gA[i][0]; rb[i] = gB[i][0]; }

    const uint32_t nchunks = D / GK;
    for (uint32_t kc = 0; kc < nchunks; ++kc) {
        __syncthreads();  // previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(&As[(srow + 32 * i) * LDS_STRIDE + seg * 8]) = ra[i];
            *reinterpret_cast<u32x4*>(&Bs[(srow + 32 * i) * LDS_STRIDE + seg * 8]) = rb[i];
        }
        __syncthreads();
        if (kc + 1 < nchunks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[i] = gA[i][(kc + 1) * 8]; rb[i] = gB[i][(kc + 1) * 8]; }
        }
#pragma unroll
        for (int ks = 0; ks < GK / 16; ++ks) {
            bf16x8 fa[2], fb[2];
            const int kofs = ks * 16 + 8 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 ua = *reinterpret_cast<const u32x4*>(&As[(wm * 64 + i * 32 + (lane & 31)) * LDS_STRIDE + kofs]);
                const u32x4 ub = *reinterpret_cast<const u32x4*>(&Bs[(wn * 64 + i * 32 + (lane & 31)) * LDS_STRIDE + kofs]);
                fa[i] = __builtin_bit_cast(bf16x8, ua);
                fb[i] = __builtin_bit_cast(bf16x8, ub);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31 (corpus row),
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5) (query). Fused selection: each approx distance is tested
    // against its query's threshold (staged in LDS); survivors are appended to the query's global
    // candidate list with one atomic each. After the first slab a tile appends almost nothing.
    float* tau_s = reinterpret_cast<float*>(As);          // [GM] thresholds of this tile's queries
    float* qn2 = reinterpret_cast<float*>(As) + GM;       // [GM] ||q||^2 (L2 only)
    __syncthreads();                                      // every wave is done reading As/Bs fragments
    unsigned int* wcnt = reinterpret_cast<unsigned int*>(As) + 2 * GM;  // [4] staged-candidate counters, one per wave
    if (tid < GM) {
        tau_s[tid] = a.tau[m0 + tid];
        if (METRIC == BM_L2) qn2[tid] = a.q_n2[m0 + tid];
        if (tid < 4) wcnt[tid] = 0u;
    }
    __syncthreads();
    // Thresholds (and ||q||^2) of the 32 queries this lane's accumulators belong to, fetched once:
    // a per-element LDS read + compare + branch chain costs more than the MFMAs of the tile.
    float tq[2][16], qq[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qloc = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            tq[i][r] = tau_s[qloc];
            qq[i][r] = (METRIC == BM_L2) ? qn2[qloc] : 0.f;
        }
    const uint32_t slab_end = a.slab0 + a.slab_rows;
    uint32_t rowj[2];
    float vn2j[2];
    if (SAMPLE) {
        // per query: the best similarity over this workgroup's 128 rows — over j and the 32 lanes of a half-wave (DPP),
        // then over the two wave columns through LDS
        float* smax = reinterpret_cast<float*>(Bs);       // [2 wave columns][GM]; Bs is free after the K loop
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rowj[j] = n0 + wn * 64 + j * 32 + (lane & 31);
            vn2j[j] = (METRIC == BM_L2 && rowj[j] < slab_end) ? a.v_n2[rowj[j]] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float best = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float dot = acc[i][j][r];
                    const float sim = (METRIC == BM_L2) ? -((qq[i][r] + vn2j[j] - 2.0f * dot) + 0.0f) : dot;
                    best = __builtin_fmaxf(best, rowj[j] < slab_end ? sim : -__builtin_inff());   // NaN never wins (maxNum)
                }
                best = group_max32(best);
                if ((lane & 31) == 31) smax[wn * GM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = best;
            }
        __syncthreads();
        if (tid < GM) a.tile_max[(size_t)ct * (a.nqt * 128u) + m0 + (uint32_t)tid] = __builtin_fmaxf(smax[tid], smax[GM + tid]);
        return;
    }
    unsigned long long pass = 0ull;  // bit j*32 + i*16 + r
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rowj[j] = n0 + wn * 64 + j * 32 + (lane & 31);
        const bool row_ok = rowj[j] < slab_end;
        vn2j[j] = (METRIC == BM_L2 && row_ok) ? a.v_n2[rowj[j]] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dot = acc[i][j][r];
                float d;
                if (METRIC == BM_L2) d = qq[i][r] + vn2j[j] - 2.0f * dot;
                else d = 1.0f - dot;  // cosine: both operands were normalised by mirror_kernel; dot: USearch ip
                d += 0.0f;
                // NaN fails the test (such rows can never be candidates); padded queries have tau = -inf
                const bool p = row_ok && (d <= tq[i][r]);
                pass |= (unsigned long long)(p ? 1u : 0u) << (j * 32 + i * 16 + r);
            }
    }
    if (a.dense != nullptr) {
        // First slab: no threshold exists yet, every distance would be appended. Store the tile densely
        // (coalesced 128-byte runs per query row); tighten_kernel reads it back as the first candidate set.
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (rowj[j] >= slab_end) continue;
            float* __restrict__ dst = a.dense + (rowj[j] - a.slab0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qloc = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float dot = acc[i][j][r];
                    float d;
                    if (METRIC == BM_L2) d = qq[i][r] + vn2j[j] - 2.0f * dot;
                    else d = 1.0f - dot;
                    d = (d != d) ? __builtin_inff() : d;
                    dst[(size_t)(m0 + qloc) * a.dense_ld] = d + 0.0f;
                }
        }
        return;
    }
    if (!__any(pass != 0ull)) return;  // the common case once tau is tight
    // Survivors (~kp * slab / rows_seen per query per slab) are first compacted into a per-wave LDS
    // stage (no global traffic), then appended with all 64 lanes' atomics in flight at once: one
    // atomic round trip per 64 survivors instead of one per accumulator slot.
    constexpr unsigned STAGE_CAP = 256;                                   // 16-byte entries per wave
    u32x4* stage = reinterpret_cast<u32x4*>(Bs) + wave * STAGE_CAP;        // Bs is free after the K loop (18 KB)
    const unsigned mine = (unsigned)__popcll(pass);
    unsigned off = 0;
    if (mine) off = atomicAdd(&wcnt[wave], mine);                          // LDS atomic: exclusive offset of this lane
    wave_lds_fence();
    const unsigned total = (unsigned)__builtin_amdgcn_readfirstlane((int)wcnt[wave]);
    if (total <= STAGE_CAP) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if ((pass >> (j * 32 + i * 16 + r)) & 1ull) {
                        const int qloc = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const float dot = acc[i][j][r];
                        float d;
                        if (METRIC == BM_L2) d = qq[i][r] + vn2j[j] - 2.0f * dot;
                        else d = 1.0f - dot;
                        d += 0.0f;
                        const int64_t key = make_key(d, a.row_base + rowj[j]);
                        u32x4 e;
                        e.x = (unsigned)((unsigned long long)key & 0xffffffffull);
                        e.y = (unsigned)((unsigned long long)key >> 32);
                        e.z = m0 + (unsigned)qloc;
                        e.w = 0u;
                        stage[off++] = e;
                    }
                }
        wave_lds_fence();
        for (unsigned t = (unsigned)lane; t < total; t += WAVE) {
            const u32x4 e = stage[t];
            const uint32_t q = e.z;
            const uint32_t pos = atomicAdd(&a.cand_count[(size_t)q * CAND_COUNT_STRIDE], 1u);
            if (pos < a.cand_cap)
                a.cand[(size_t)q * a.cand_cap + pos] = (int64_t)(((unsigned long long)e.y << 32) | (unsigned long long)e.x);
        }
        return;
    }
    // Stage overflow (a loose threshold, e.g. adversarially ordered rows): direct per-slot appends.
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((pass >> (j * 32 + i * 16 + r)) & 1ull) {
                    const int qloc = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float dot = acc[i][j][r];
                    float d;
                    if (METRIC == BM_L2) d = qq[i][r] + vn2j[j] - 2.0f * dot;
                    else d = 1.0f - dot;
                    d += 0.0f;
                    const uint32_t q = m0 + qloc;
                    const uint32_t pos = atomicAdd(&a.cand_count[(size_t)q * CAND_COUNT_STRIDE], 1u);
                    if (pos < a.cand_cap) a.cand[(size_t)q * a.cand_cap + pos] = make_key(d, a.row_base + rowj[j]);
                }
            }
}

typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) unsigned int lds_u32;

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// ---------------------------------------------------------------------------
// Register-resident-queries GEMM (cosine / dot, D in {128, 256, 384, 512}; every slab after the first).
//
// The query block is tiny and reused against every corpus row, so it never goes through LDS: each of
// the 8 waves of a workgroup keeps its 32 queries x D as MFMA A-fragments in VGPRs for the whole launch
// (96 VGPRs at D = 384). Only the corpus streams: persistent workgroups walk 64-row tiles (contiguous
// 48 KB in the bf16 mirror), double-buffered in LDS with one barrier per tile; all 8 waves read the
// same B fragments (ds_read_b128, rows padded by 16 B => conflict-free), so per tile a SIMD issues
// 2 waves x 48 MFMAs against 384 KB of LDS reads (50 % of the LDS pipe). At Q = 256 one pass over the
// corpus is HBM-bound (48 KB per 3072 MFMA cycles per CU = 9.6 TB/s at MFMA peak).
// Epilogue per tile: 32 compares against the lane's 16 thresholds. A survivor takes a slot in THIS workgroup's
// segment of its query's candidate row (slot index from an LDS counter, a plain 8-byte global store) — no
// global atomics: with ~50 K survivors per slab funnelled through 256 counters in 8 cache lines, device-scope
// atomics cost ~0.9 us per thousand survivors (a 16 K-row slab took 120 us, profiles/r01/y_growth_trace_tail.csv).
//
// Staging, GLDS = false: global -> VGPRs (issued at the top of an iteration) -> ds_write at its end, two LDS tiles.
// Staging, GLDS = true: LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass). The DMA writes
// 64 lanes x 16 B contiguously, so the padded tile image (row stride D*2 + 16) is cut into 1-KB pieces and each lane
// FETCHES whatever belongs at its slot (the pad slot of a row re-fetches the row's last segment). With three
// tiles in LDS (D <= 384) a tile is requested two iterations before it is read: the wait at the end of an
// iteration is a counted vmcnt that leaves the newest tile in flight across the (raw) barrier.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

template <int N>
__device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Claim the next tile of a group: a returning global atomic whose result is NOT waited for here. (HIP's atomicAdd is
// rewritten by the compiler's atomic optimizer into bcnt + atomic + `s_waitcnt vmcnt(0)` + readfirstlane on the spot,
// which also drains the tile loads issued just before it — the whole prefetch.) The caller executes
// `s_waitcnt vmcnt(0)` (claim_wait) before reading the value; the compiler's own counted waits only ever over-wait
// because of the extra request in flight.
__device__ __forceinline__ unsigned int claim_tile_async(uint32_t* ctr) {
    unsigned int old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(old) : "v"(ctr), "v"(1u) : "memory");
    return old;
}
__device__ __forceinline__ void claim_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int D>
constexpr int rega_lds_tiles(bool glds) { return (glds && 3 * 64 * (D * 2 + 16) + 3088 <= 160 * 1024) ? 3 : 2; }

// SAMPLE = true (one-pass pipeline, threshold estimation): instead of filtering against a threshold, the workgroup
// visits `a.sample_tiles` tiles spread evenly over the slab (logical index i -> tile i * ntiles / sample_tiles) and
// records, per query, the best similarity of each visited tile in a.tile_max[i][query] (pick_tau_kernel turns the
// j-th best tile maximum into the query's admission threshold).
// FREE = true ("batch_rega" = 4; register staging, THREE LDS tiles; where they fit: D <= 384): NO workgroup barrier in the tile loop.
// The phase clock (debug bit 10, profiles/r03/i_gemm_phase_clock.txt) showed where the ~40 % of a tile period that is not
// matrix-pipe time goes: the one barrier per tile joins eight waves whose per-tile durations differ (arbitration of the shared
// matrix pipe and the LDS port), and every wave waits for the slowest — 13 % (late half) to 27 % (early half) of its cycles.
// What the barrier protected is two hazards on the staged tiles, and each is covered by one LDS counter per tile buffer:
//   RAW  stored[b]: a wave adds 1 behind its last store of a tile into buffer b (a wave's DS operations execute in order);
//        a reader spins until the counter shows 8 x (the buffer's fill number) before its first fragment read.
//   WAR  read[b]: a wave adds 1 behind its last fragment read of the tile in buffer b; a writer spins on it before its first
//        store of the buffer's next fill.
// The staging runs TWO tiles ahead (iteration t loads tile t+2 from HBM at its top and stores it from the second half of its
// K loop into the buffer tile t-1 was read from), so the RAW wait of iteration t is on stores made during iteration t-2 — a
// whole tile period of slack — and the WAR wait, half-way through K loop t, is on reads that ended with K loop t-1: half a
// period of slack. Waves drift apart by that much and re-converge without anybody having waited for it.
// Measured (profiles/r03/j_gemm_phase_clock_free_running.txt, r_bench_free_running_ab.txt): the wait falls from 13-27 % of a
// wave's cycles to 5 %, the kernel alone gains 2-5 % (frac 0.477 -> 0.491 at Q = 256) — and the pipelined batches LOSE 1.7 %,
// because a 150 KB workgroup keeps the neighbouring batch's finish / prep kernels off the CU. Hence a variant, not the default.
// What the remaining gap is made of is in DESIGN.md ("the matrix roof"): with embedding-like operands the K loop ALONE
// (tools/mfma_probe.hip) sustains 1.57 PFLOP/s on this part — the matrix clock is power-limited and data-dependent.
//
// SYNC = 2 ("batch_rega" = 5): the SPLIT barrier — the default's two LDS tiles and one-tile-ahead staging, but the per-tile
// workgroup barrier becomes "arrive" (a wave adds 1 to an LDS counter behind its K loop) and "wait" (spin on the counter in front
// of the next K loop): a wave's selection sits between the two, so it no longer waits for the slowest wave before selecting.
// Everybody still finishes K loop t-1 before anybody starts K loop t (that is what covers both hazards with two tiles), so there
// is less slack than in the free-running variant — and no extra LDS.
template <int D, bool GLDS, int AHEAD, bool SAMPLE = false, bool PROF = false, int SYNC = 0>
__global__ __launch_bounds__(512, 2) void batch_gemm_rega_kernel(GemmArgs a, uint32_t blocks_per_group) {
    constexpr bool FREE = SYNC == 1, SPLIT = SYNC == 2;
    static_assert(!SPLIT || (!GLDS && !SAMPLE), "the split barrier is a variant of the register-staged filtering kernel");
    constexpr int KS = D / 16;                       // MFMA k-steps
    constexpr int ROW_B = D * 2 + 16;                // LDS row stride (bytes)
    constexpr int TROWS = 64;                        // corpus rows per tile
    constexpr int SEG_PER_ROW = D * 2 / 16;
    constexpr int SEGS = TROWS * SEG_PER_ROW;        // 16-byte segments per tile
    constexpr int LOADS = SEG_PER_ROW / 8;           // per thread (8 threads per row, 64 rows)
    constexpr int BUF_B = TROWS * ROW_B;
    static_assert(!FREE || (!GLDS && !SAMPLE), "the free-running variant is the register-staged filtering kernel");
    constexpr int NBUF = FREE ? 3 : rega_lds_tiles<D>(GLDS);
    constexpr int PRE = NBUF - 1;                    // GLDS: tiles requested ahead of the one being read
    constexpr int SLOTS_PER_ROW = ROW_B / 16;        // 16-byte slots per padded row; == 1-KB pieces per tile (64 rows)
    constexpr int PIECES = SLOTS_PER_ROW;
    constexpr int PPW = (PIECES + 7) / 8;            // pieces per wave (waves with index >= PIECES % 8 carry one less)
    static_assert(SEGS == 512 * LOADS && D % 64 == 0, "tile must split evenly over 512 threads");
    static_assert(BUF_B == PIECES * 1024, "a padded tile is a whole number of 1-KB DMA pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* buf0 = smem;
    float* tau_s = reinterpret_cast<float*>(smem + NBUF * BUF_B);              // [8][32] exact thresholds
    unsigned int* cnt_s = reinterpret_cast<unsigned int*>(tau_s + 8 * 32);     // [8][32] survivors per query (this workgroup)
    float* sim_s = reinterpret_cast<float*>(cnt_s + 8 * 32);                   // [8][32] conservative similarity bounds
    unsigned int* next_s = reinterpret_cast<unsigned int*>(sim_s + 8 * 32);    // [2] claimed tile indices (dynamic tile order)
    unsigned int* stored_s = next_s + 4;                                       // [3] FREE: wave signals per LDS tile buffer: "my stores of its current fill are in"
    unsigned int* read_s = next_s + 8;                                         // [3] FREE: "my reads of its current fill are done"
    unsigned int* turn_s = next_s + 12;                                        // [4] FREE, debug bit 11: whose K loop runs next on each SIMD (pipe token)

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t group = blockIdx.x / blocks_per_group;   // 256 queries per group
    const uint32_t bidx = blockIdx.x % blocks_per_group;
    const uint32_t q0 = group * 256 + wave * 32;             // this wave's 32 queries

    // A fragments: lane l holds query (l & 31), k = 16*ks + 8*(l >> 5) .. +7
    bf16x8 fa[KS];
    {
        const u32x4* qp = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + (lane & 31)) * D) + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[ks] = __builtin_bit_cast(bf16x8, qp[ks * 2]);
    }
    if (!SAMPLE && lane < 32) {
        const float tq = a.tau[q0 + lane];
        tau_s[wave * 32 + lane] = tq;
        // fl(1 - acc) <= tq implies acc >= (1 - tq) - 2^-23 (|1 - tq| + |tq|); 4e-7 (1 + |tq|) covers it with slack.
        // tq = +inf (no threshold yet) gives -inf: everything is flagged; tq = -inf (padding query) gives NaN: nothing is.
        sim_s[wave * 32 + lane] = (1.0f - tq) - 4e-7f * (1.0f + __builtin_fabsf(tq));
    }
    if (tid < 256) cnt_s[tid] = 0u;
    if (SYNC != 0 && tid == 0) next_s[3] = 0u;                                 // "a wave gave up waiting" (see wait_count)
    if (FREE && tid < 3) {
        stored_s[tid] = tid < 2 ? 8u : 0u;                                     // tiles 0 and 1 are staged by the prologue, behind a barrier
        read_s[tid] = 0u;
    }
    if (SPLIT && tid == 0) stored_s[0] = 0u;                                   // SPLIT: K loops finished, all waves (arrivals of the split barrier)
    if (FREE && tid >= 64 && tid < 68) turn_s[tid - 64] = 0u;
    // this workgroup's segment of every query's candidate row
    const uint32_t seg_slots = a.seg_area / blocks_per_group;
    // element offset (32-bit: rows * cand_cap < 2^23) of this lane's first query row, at this workgroup's segment;
    // kept as ONE VGPR + per-query scalar multiples — 16 hoisted 64-bit row pointers would spill
    const uint32_t seg_lane0 = (q0 + 4u * ((uint32_t)lane >> 5)) * a.cand_cap + a.seg_base + bidx * seg_slots;

    const uint32_t ntiles_all = (a.slab_rows + TROWS - 1) / TROWS;
    const uint32_t ntiles = SAMPLE ? a.sample_tiles : ntiles_all;   // loop range (logical tiles)
    const uint32_t slab_end = a.slab0 + a.slab_rows;
    const unsigned char* cbase = reinterpret_cast<const unsigned char*>(a.cb);
    // logical -> physical tile (identity unless sampling)
    auto phys = [&](uint32_t tile) -> uint32_t {
        return SAMPLE ? (uint32_t)(((unsigned long long)tile * ntiles_all) / a.sample_tiles) : tile;
    };

    // register staging map: 8 threads per tile row; a thread moves the 16-byte segments (tid & 7) + 8*p of its
    // row, so every global / LDS address is one per-tile base plus a compile-time offset (no address arrays).
    const uint32_t srow = (uint32_t)tid >> 3;
    const uint32_t sseg = ((uint32_t)tid & 7u) * 16u;
    u32x4 regs[GLDS ? 1 : LOADS];
    auto issue_loads = [&](uint32_t tile) {
        uint32_t grow = a.slab0 + phys(tile) * TROWS + srow;
        grow = grow < a.n_rows ? grow : a.n_rows - 1;        // clamp: masked in the epilogue
        const unsigned char* src = cbase + (size_t)grow * (D * 2) + sseg;
#pragma unroll
        for (int p = 0; p < (GLDS ? 0 : LOADS); ++p) regs[p] = *reinterpret_cast<const u32x4*>(src + p * 128);
    };
    auto store_tile = [&](unsigned char* buf) {
        unsigned char* dst = buf + srow * ROW_B + sseg;
#pragma unroll
        for (int p = 0; p < (GLDS ? 0 : LOADS); ++p) *reinterpret_cast<u32x4*>(dst + p * 128) = regs[p];
    };
    // LDS-DMA map: wave w moves pieces w, w + 8, ...; lane l of piece P fills slot P*64 + l of the padded image
    uint32_t prow[PPW], pcol[PPW];
    int my_pieces = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const uint32_t P = (uint32_t)wave + 8u * i;
        const uint32_t slot = P * 64u + (uint32_t)lane;
        const uint32_t r = slot / SLOTS_PER_ROW;
        uint32_t c = slot - r * SLOTS_PER_ROW;
        c = c < (uint32_t)SEG_PER_ROW ? c : (uint32_t)SEG_PER_ROW - 1u;   // pad slot: any valid bytes
        prow[i] = r;
        pcol[i] = c * 16u;
        if (P < (uint32_t)PIECES) ++my_pieces;
    }
    const bool full_wave = my_pieces == PPW;                 // wave-uniform
    auto dma_tile = [&](uint32_t tile, uint32_t buf_off) {
        const uint32_t row0 = a.slab0 + phys(tile) * TROWS;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const uint32_t P = (uint32_t)wave + 8u * i;
            if (i < PPW - 1 || full_wave) {
                uint32_t grow = row0 + prow[i];
                grow = grow < a.n_rows ? grow : a.n_rows - 1;
                const unsigned char* src = cbase + (size_t)grow * (D * 2) + pcol[i];
                __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(smem + buf_off + P * 1024u), 16, 0, 0);
            }
        }
    };
    // this wave's DMA requests still allowed in flight: `keep_tiles` whole tiles (0 or 1)
    auto dma_wait = [&](bool keep_one_tile) {
        if (!keep_one_tile) wait_vmcnt<0>();
        else if (full_wave) wait_vmcnt<PPW>();
        else wait_vmcnt<PPW - 1>();
    };

    // Tile order. Static: bidx, bidx + W, bidx + 2W, ... (W = workgroups per group). Dynamic (a.tile_ctr, register
    // staging only): the first two tiles are the static ones, every further one is claimed from the group's counter one
    // iteration before its loads are issued — thread 0 starts the atomic at the top of an iteration and parks the
    // result in LDS at its end, next to the tile barrier, so its latency never sits on the critical path.
    const bool dyn = !SAMPLE && !GLDS && !FREE && !SPLIT && a.tile_ctr != nullptr;
    uint32_t t = bidx;
    if (dyn && tid == 0) {
        const unsigned int c0 = claim_tile_async(a.tile_ctr + group * 32u);
        claim_wait();
        next_s[0] = 2u * blocks_per_group + c0;
    }
    if (GLDS) {
        bool second = false;
        if (t < ntiles) dma_tile(t, 0u);
        if (PRE == 2 && t + blocks_per_group < ntiles) { dma_tile(t + blocks_per_group, (uint32_t)BUF_B); second = true; }
        dma_wait(second);
        __builtin_amdgcn_s_barrier();                         // also publishes tau_s / cnt_s
        asm volatile("" ::: "memory");
    } else {
        if (t < ntiles) {
            issue_loads(t);
            store_tile(buf0);
        }
        if (FREE && t + blocks_per_group < ntiles) {
            issue_loads(t + blocks_per_group);
            store_tile(buf0 + BUF_B);
        }
        __syncthreads();
    }
    // The two waves that share a SIMD (w and w + 4: a workgroup's waves are dealt to the SIMDs cyclically) do
    // the same work per tile in OPPOSITE order. Waves 0-3 run MFMAs(t) then select(t); waves 4-7 run
    // select(t - 1) — on accumulators carried over from the previous iteration — then MFMAs(t). While one wave
    // of a SIMD is in its VALU/LDS-only selection the other has the matrix pipe to itself, so the selection
    // (~25 % of a tile's issue slots) hides under MFMAs instead of idling the pipe for both waves at once.
    const bool late = wave >= 4 && !(a.debug & 16u);   // debug bit4: every wave in the same order
    const bool dbg_noload = !FREE && (a.debug & 1u) != 0, dbg_nomfma = !FREE && (a.debug & 2u) != 0;  // timing experiments only
    const bool prio = (a.debug & 32u) == 0;            // s_setprio 1 around the MFMA stream (+2-3 % at Q = 1024); debug bit5 turns it off
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // K loop, software-pipelined in the source: the B fragments of k-step ks + AHEAD are read (2 ds_read_b128)
    // before the two MFMAs of k-step ks issue, and a sched_barrier pins every step, so the wait in front of an
    // MFMA is a counted lgkmcnt(2 * AHEAD) on reads issued AHEAD steps earlier — never on the reads just issued.
    // (Left alone, hipcc hoists all 2*KS reads above the MFMAs: 192 VGPRs at D = 384, spilling the A fragments;
    // with sched_group_barrier "2 DS, 2 MFMA" groups it emitted `ds_read x2; s_waitcnt lgkmcnt(0); mfma`, i.e. one
    // exposed LDS round trip per k-step and a matrix pipe ~50 % idle.)
    // `st_dst` != nullptr: the next tile's staged segments (regs[], loaded at the top of the iteration) are written to LDS
    // from INSIDE the K loop — piece p in front of k-step KS/2 + 2p — instead of in one burst after it: the burst was a
    // phase of ~600 LDS cycles per tile in which no wave of the workgroup had MFMAs left to issue.
    // FREE / SPLIT: spin until counter `ctr` (LDS) shows `target`. Wave-uniform. Bounded (~0.3 s), so that a protocol error cannot
    // hang the GPU — and a wave that gives up says so: the workgroup then reports every one of its queries as overflowed
    // (count 2^30, below), which sends them to the exact path. Never a silent wrong answer.
    bool gave_up = SYNC != 0 && (a.debug & 16384u) != 0 && blockIdx.x == 1 && wave == 3;   // debug bit 14: pretend one wave timed out (tests)
    auto wait_count = [&](const unsigned int* ctr, unsigned int target) {
        bool ok = false;
        for (unsigned int spins = 0; spins < (1u << 22); ++spins) {
            const unsigned int v = __hip_atomic_load((const lds_u32*)ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)v) >= target) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) gave_up = true;
        asm volatile("" ::: "memory");
    };
    // FREE: one signal per wave, behind everything the wave has issued to the LDS so far (a wave's DS operations execute in
    // order; the wait makes "issued" "done" for the reads, whose data the MFMAs have consumed anyway)
    auto signal_count = [&](unsigned int* ctr) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_u32*)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // war_ctr / war_target (FREE): the buffer st_dst points into must have been read by all eight waves before the first store
    // debug bit 11 (FREE): the two waves of a SIMD (w, w + 4) take strict turns at the matrix pipe — wave w runs K loop t as turn
    // 2t, wave w + 4 as turn 2t + 1, the other one is in its selection meanwhile; the turn is handed over TOKEN_REL k-steps
    // before the end of the loop to cover the hand-over latency
    const bool token = FREE && (a.debug & 2048u) != 0;
    const bool dbg_nostage = FREE && (a.debug & 4096u) != 0;   // debug bit 12 (FREE, timing only): no HBM loads, no staging stores, no counters
    constexpr int TOKEN_REL = 2;
    auto mfma_tile = [&](const unsigned char* cur, unsigned char* st_dst, const unsigned int* war_ctr = nullptr, unsigned int war_target = 0u, unsigned int my_turn = 0u) {
        if (dbg_nomfma) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            if (st_dst) store_tile(st_dst - (srow * ROW_B + sseg));
            return;
        }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // first k-step: C = 0 (inline constant), no 32 v_mov per tile
        const unsigned char* b0 = cur + (lane & 31) * ROW_B + (lane >> 5) * 16;
        const unsigned char* b1 = b0 + 32 * ROW_B;
        constexpr int RING = AHEAD + 1;
        u32x4 fb0[RING], fb1[RING];
#pragma unroll
        for (int i = 0; i < AHEAD && i < KS; ++i) {
            fb0[i] = *reinterpret_cast<const u32x4*>(b0 + i * 32);
            fb1[i] = *reinterpret_cast<const u32x4*>(b1 + i * 32);
        }
        if (FREE && token) wait_count(turn_s + (wave & 3), my_turn);
        if (prio) __builtin_amdgcn_s_setprio(1);   // the SIMD's other wave is in its selection: MFMA issue first
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (FREE && ks == KS - TOKEN_REL && token && lane == 0)
                __hip_atomic_store((lds_u32*)(turn_s + (wave & 3)), my_turn + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ks + AHEAD < KS) {
                fb0[(ks + AHEAD) % RING] = *reinterpret_cast<const u32x4*>(b0 + (ks + AHEAD) * 32);
                fb1[(ks + AHEAD) % RING] = *reinterpret_cast<const u32x4*>(b1 + (ks + AHEAD) * 32);
            }
            if (!GLDS && ks >= KS / 2 && ((ks - KS / 2) & 1) == 0 && (ks - KS / 2) / 2 < LOADS) {
                if (FREE && ks == KS / 2 && st_dst && war_ctr) wait_count(war_ctr, war_target);
                if (st_dst) *reinterpret_cast<u32x4*>(st_dst + ((ks - KS / 2) / 2) * 128) = regs[GLDS ? 0 : (ks - KS / 2) / 2];
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fb0[ks % RING]), ks == 0 ? zero16 : acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fb1[ks % RING]), ks == 0 ? zero16 : acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
    };
    // Fused selection on the accumulators of `tile`: C[query][row], col = lane & 31 = corpus row,
    // reg r = query qo(r) = (r&3) + 8(r>>2) + 4(lane>>5).
    // Fast path, every tile: 32 compares of the accumulators (similarities) against the lane's 16 thresholds,
    // OR-ed into four wave-level flags (one per group of four queries) — no branches, one LDS round trip. (Reading each threshold from LDS next to
    // its compare cost 16 exposed LDS round trips per tile, ~25 % of the kernel: profiles/r01/ae_gemm_probe.txt.)
    // sim_s[q] = (1 - tau) - 4e-7 (1 + |tau|) is a conservative similarity bound: whatever passes
    // the exact test `1 - acc <= tau` below also passes `acc >= sim_lo`, so the mask is a superset.
    // Slow path (a wave-tile holds ~0.8 survivors at Q = 256, k' = 64): groups of four queries are re-examined
    // only if some lane flagged them; the group re-tests exactly (`1 - acc <= tau`, tau from LDS) and a survivor
    // takes a slot in this workgroup's segment of the query's candidate row.
    auto select_tile = [&](uint32_t tile) {
        if (a.debug & 8u) return;
        if (SAMPLE) {
            // best similarity of this tile per query: rows sit in lanes (lane & 31) of both accumulators; a clamped
            // row past the end of the store duplicates the last row and cannot raise a maximum. NaN never wins (maxNum).
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = group_max32(__builtin_fmaxf(acc0[r], acc1[r]));
                if ((lane & 31) == 31)
                    a.tile_max[(size_t)tile * (a.nqt * 128u) + q0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))] = m;
            }
            return;
        }
        // the lane's 16 bounds are four aligned float4 in LDS (queries 8j + 4(lane>>5) .. +3 for r = 4j .. 4j+3):
        // one batch of ds_read_b128 and ONE wait per tile; they are live only here, after the B-fragment ring died
        const lds_f32x4* sim_w = (const lds_f32x4*)(sim_s + wave * 32 + 4 * (lane >> 5));
        f32x4 lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = sim_w[2 * j];
        // wave-level flags per group of four queries: v_cmp into an SGPR pair + s_or_b64 (1 VALU + 1 SALU per element)
        unsigned long long hit[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int r = 0; r < 16; ++r)
            hit[r >> 2] |= __ballot(acc0[r] >= lo[r >> 2][r & 3]) | __ballot(acc1[r] >= lo[r >> 2][r & 3]);   // NaN fails
        if ((hit[0] | hit[1] | hit[2] | hit[3]) == 0ull || (a.debug & 64u)) return;   // debug bit6: hot test only (timing experiments)
        const uint32_t row0 = a.slab0 + tile * TROWS + (lane & 31);
        const uint32_t row1 = row0 + 32;
        const bool ok0 = row0 < slab_end, ok1 = row1 < slab_end;
        const lds_f32* tau_w = (const lds_f32*)(tau_s + wave * 32);
        lds_u32* cnt_w = (lds_u32*)(cnt_s + wave * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (hit[g] == 0ull) continue;
            const f32x4 tau4 = *(const lds_f32x4*)(tau_w + 8 * g + 4 * (lane >> 5));   // exact thresholds of the group
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                const int qo = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float tqr = tau4[r & 3];
                const float d0 = (1.0f - acc0[r]) + 0.0f, d1 = (1.0f - acc1[r]) + 0.0f;
                const bool p0 = ok0 && d0 <= tqr, p1 = ok1 && d1 <= tqr;
                if (p0 || p1) {
                    // slot in this workgroup's segment of the query's row: one LDS atomic, plain global stores
                    const unsigned n = (p0 ? 1u : 0u) + (p1 ? 1u : 0u);
                    unsigned off;
                    if (GLDS) {
                        // opaque to hipcc on purpose: before an LDS write it can see, the compiler drains every
                        // outstanding LDS-DMA request (s_waitcnt vmcnt(0)) — here once per tile, undoing the prefetch
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                     : "=v"(off)
                                     : "v"((unsigned)(size_t)(cnt_w + qo)), "v"(n)
                                     : "memory");
                    } else {
                        off = __hip_atomic_fetch_add(cnt_w + qo, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    const uint32_t e0 = seg_lane0 + (uint32_t)((r & 3) + 8 * (r >> 2)) * a.cand_cap;
                    if (a.debug & 8192u) continue;           // debug bit13: count, do not store (timing experiments: what do the stores cost the tile pipeline?)
                    if (p0 && off < seg_slots) a.cand[e0 + off++] = make_key(d0, a.row_base + row0);
                    if (p1 && off < seg_slots) a.cand[e0 + off] = make_key(d1, a.row_base + row1);
                }
            }
        }
    };

    constexpr bool prof = PROF;   // a separate instantiation (launch_rega, debug bit 10): the counters and the printf cost 8 VGPRs
    unsigned long long prof_mfma = 0, prof_sel = 0, prof_bar = 0;
    const unsigned long long prof_t0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t it = 0;
    uint32_t cur_idx = 0;                                     // GLDS / FREE: it % NBUF
    uint32_t pre_idx3 = 2;                                    // FREE: (it + 2) % 3, the buffer this iteration stages into
    uint32_t t_prev = 0;                                      // the tile of the previous iteration (late waves select it now)
    uint32_t t_next = t + blocks_per_group;                   // register staging: the tile whose loads this iteration issues
    for (; t < ntiles; ++it) {
        unsigned char* cur;
        unsigned char* nxt = buf0;
        uint32_t tn;
        uint32_t t_after = 0;                                 // register staging: the tile after t_next
        unsigned int claimed = 0;
        bool issued = false;
        if (GLDS) {
            cur = buf0 + cur_idx * BUF_B;
            tn = t + PRE * blocks_per_group;
            uint32_t pre_idx = cur_idx + PRE;
            pre_idx = pre_idx >= (uint32_t)NBUF ? pre_idx - NBUF : pre_idx;
            if (tn < ntiles) { dma_tile(tn, pre_idx * BUF_B); issued = true; }
        } else if (FREE) {
            cur = buf0 + cur_idx * BUF_B;
            nxt = buf0 + pre_idx3 * BUF_B;                        // where tile it + 2 goes: the buffer tile it - 1 was read from
            tn = dbg_nostage ? ntiles : t_next + blocks_per_group;   // the tile staged in this iteration: two ahead
            if (tn < ntiles) issue_loads(tn);
        } else {
            cur = buf0 + ((dbg_noload ? 0u : (it & 1u)) * BUF_B);  // debug bit0: always the prologue tile
            nxt = buf0 + ((it & 1) ^ 1) * BUF_B;
            tn = t_next;
            if (tn < ntiles && !dbg_noload) issue_loads(tn);
            if (dyn) {
                t_after = next_s[it & 1u];
                if (tid == 0 && t_after < ntiles) claimed = claim_tile_async(a.tile_ctr + group * 32u);   // read at the end of the iteration
            } else {
                t_after = tn + blocks_per_group;
            }
        }
        // debug bit7: the staged tile goes to LDS in one burst after the K loop (the round-1 schedule), for A/B timing
        // (ignored by the split / free-running barriers: their arrival is signalled right after the K loop, so a burst
        // stored after it would not be covered and other waves could read a half-written tile)
        const bool spread = !GLDS && (FREE || SPLIT || !(a.debug & 128u));
        unsigned char* st_dst = (!GLDS && spread && tn < ntiles && !dbg_noload) ? nxt + srow * ROW_B + sseg : nullptr;
        // FREE: tile `it` is fill it / 3 + 1 of its buffer; the buffer being refilled was read as tile it - 1
        const unsigned int raw_target = 8u * (it / 3u + 1u), war_target = it ? 8u * ((it - 1u) / 3u + 1u) : 0u;
        // debug bit10: per-wave phase clock (s_memtime around the MFMA phase, the selection and the tile barrier), printed by a
        // few workgroups at the end — where the 40 % of a tile period that is not matrix-pipe time goes. Perturbs the timing
        // (every reading drains the wave's LDS queue): a diagnosis run, never a benchmark.
        unsigned long long c0 = 0, c1 = 0, c2 = 0;
        if (prof) c0 = __builtin_amdgcn_s_memtime();
        if (late) {
            if (it > 0) select_tile(t_prev);
            if (prof) { c1 = __builtin_amdgcn_s_memtime(); prof_sel += c1 - c0; }
            if (FREE) {
                if (!dbg_nostage) wait_count(stored_s + cur_idx, raw_target);
                if (prof) { const unsigned long long cw = __builtin_amdgcn_s_memtime(); prof_bar += cw - c1; c1 = cw; }
                mfma_tile(cur, st_dst, it ? read_s + pre_idx3 : nullptr, war_target, 2u * it + (wave >= 4 ? 1u : 0u));
                signal_count(read_s + cur_idx);
                if (st_dst && lane == 0) __hip_atomic_fetch_add((lds_u32*)(stored_s + pre_idx3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (SPLIT && it > 0) wait_count(stored_s, 8u * it);    // every wave has finished K loop it - 1: tile `it` is stored, its other buffer is free
                mfma_tile(cur, st_dst);
                if (SPLIT) signal_count(stored_s);
            }
            if (prof) { c2 = __builtin_amdgcn_s_memtime(); prof_mfma += c2 - c1; }
        } else {
            if (FREE) {
                if (!dbg_nostage) wait_count(stored_s + cur_idx, raw_target);
                if (prof) { const unsigned long long cw = __builtin_amdgcn_s_memtime(); prof_bar += cw - c0; c0 = cw; }
                mfma_tile(cur, st_dst, it ? read_s + pre_idx3 : nullptr, war_target, 2u * it + (wave >= 4 ? 1u : 0u));
                signal_count(read_s + cur_idx);
                if (st_dst && lane == 0) __hip_atomic_fetch_add((lds_u32*)(stored_s + pre_idx3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (SPLIT && it > 0) wait_count(stored_s, 8u * it);
                mfma_tile(cur, st_dst);
                if (SPLIT) signal_count(stored_s);
            }
            if (prof) { c1 = __builtin_amdgcn_s_memtime(); prof_mfma += c1 - c0; }
            select_tile(t);
            if (prof) { c2 = __builtin_amdgcn_s_memtime(); prof_sel += c2 - c1; }
        }
        t_prev = t;
        if (FREE) {
            pre_idx3 = cur_idx;                                   // next iteration refills the buffer this one read
            cur_idx = cur_idx + 1 == 3u ? 0u : cur_idx + 1;
            t = t_next;
            t_next += blocks_per_group;
            if (prof) c2 = __builtin_amdgcn_s_memtime();
        } else if (GLDS) {
            // tile t + 1 must have landed (every wave waits for its own pieces, the barrier joins them); with
            // three tiles in LDS the one requested in this iteration stays in flight across the barrier
            dma_wait(PRE == 2 && issued);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
            t += blocks_per_group;
        } else {
            if (!spread && tn < ntiles && !dbg_noload) store_tile(nxt);
            if (dyn && tid == 0) {
                claim_wait();
                next_s[(it + 1u) & 1u] = t_after < ntiles ? 2u * blocks_per_group + claimed : t_after;
            }
            if (!SPLIT && !(a.debug & 4u)) __syncthreads();  // debug bit2 (only with bit0): no per-tile barrier
            t = tn;
            t_next = t_after;
        }
        if (prof) prof_bar += __builtin_amdgcn_s_memtime() - c2;
    }
    if (prof && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == 255))
        printf("WAXPROF rega D=%d blk %u wave %d tiles %u mfma %llu select %llu barrier %llu total %llu\n", D, (unsigned)blockIdx.x, wave, it,
               prof_mfma, prof_sel, prof_bar, (unsigned long long)(__builtin_amdgcn_s_memtime() - prof_t0));
    if (late && it > 0) select_tile(t_prev);
    if (SYNC != 0 && gave_up && lane == 0) __hip_atomic_fetch_or((lds_u32*)(next_s + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // unclamped counts: a count above seg_slots tells tighten_kernel that survivors were dropped (query -> exact path)
    __syncthreads();
    if (!SAMPLE && tid < 256) {
        const bool poisoned = SYNC != 0 && next_s[3] != 0u;   // a wave gave up waiting: nothing this workgroup selected can be trusted
        a.seg_count[(size_t)bidx * (a.nqt * 128u) + group * 256u + (uint32_t)tid] = poisoned ? 0x40000000u : cnt_s[tid];
    }
}

// ---------------------------------------------------------------------------
// Register-resident-queries GEMM for D = 768: 32 queries x D of A fragments do not fit beside the
// accumulators in 256 VGPRs, so the K dimension is split over the two waves that share a SIMD. Wave p (owner) and
// wave p + 4 (helper), p = 0..3, hold the same 32 queries; the owner keeps k in [0, D/2), the helper [D/2, D)
// (96 VGPRs each at D = 768). A workgroup therefore covers 128 queries; tiles are 32 corpus rows (48 KB at D = 768),
// double-buffered in LDS with one barrier per tile. Per tile every wave runs D/32 MFMAs on two independent
// accumulators; the helper then parks its partial sums in LDS (4 ds_write_b128, buffers alternate by tile parity),
// and at the top of the NEXT iteration — after the tile barrier — the owner adds them to its own and runs the
// selection, while the helper is already issuing the next tile's MFMAs (the pair shares a SIMD, so the owner's
// VALU/LDS work overlaps the helper's matrix work by construction).
// Selection, segments and thresholds are those of batch_gemm_rega_kernel (one row block per tile).
//
// SPLIT = true (the default for the filtering launch): the tile barrier is split as in batch_gemm_rega_kernel<..., SYNC = 2>.
// Every wave adds 1 to `done` behind its K loop and spins on it (8 x tile number) in front of the next one — that orders the
// tile buffers; a helper also adds 1 to its pair's `parked` counter behind its partial sums, and the owner waits for THAT before
// its selection — so an owner selects as soon as its own partner is through, while the slowest wave of the workgroup is still
// multiplying, instead of after everybody.
template <int D, int AHEAD, bool SAMPLE = false, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void batch_gemm_ksplit_kernel(GemmArgs a, uint32_t blocks_per_group) {
    constexpr int HALF = D / 2;
    constexpr int KS = HALF / 16;                    // MFMA k-steps per wave
    constexpr int ROW_B = D * 2 + 16;                // LDS row stride (bytes); (ROW_B / 4) % 64 == 4 => conflict-free b128 reads
    constexpr int TROWS = 32;                        // corpus rows per tile
    constexpr int THREADS_PER_ROW = 512 / TROWS;     // 16
    constexpr int LOADS = D * 2 / 16 / THREADS_PER_ROW;
    constexpr int BUF_B = TROWS * ROW_B;
    constexpr int PART_B = 4 * 64 * 16;              // one wave's partial sums: 16 floats per lane
    static_assert(D % 256 == 0 && LOADS * THREADS_PER_ROW * 16 == D * 2, "row must split evenly over 16 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* buf0 = smem;
    unsigned char* part0 = smem + 2 * BUF_B;                                    // [2 parities][4 pairs][4][64] float4
    float* tau_s = reinterpret_cast<float*>(part0 + 2 * 4 * PART_B);            // [4][32] exact thresholds
    unsigned int* cnt_s = reinterpret_cast<unsigned int*>(tau_s + 4 * 32);      // [4][32] survivors per query (this workgroup)
    float* sim_s = reinterpret_cast<float*>(cnt_s + 4 * 32);                    // [4][32] conservative similarity bounds
    unsigned int* next_s = reinterpret_cast<unsigned int*>(sim_s + 4 * 32);     // [2] claimed tile indices (dynamic tile order)
    unsigned int* done_s = next_s + 4;                                          // SPLIT: K loops finished (all waves)
    unsigned int* parked_s = next_s + 5;                                        // SPLIT: [4] partial sums parked, per pair

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int pair = wave & 3;
    const bool owner = wave < 4;
    const uint32_t group = blockIdx.x / blocks_per_group;   // 128 queries per group
    const uint32_t bidx = blockIdx.x % blocks_per_group;
    const uint32_t q0 = group * 128 + pair * 32;             // this pair's 32 queries

    // A fragments: lane l holds query (l & 31), k = khalf*HALF + 16*ks + 8*(l >> 5) .. +7
    bf16x8 fa[KS];
    {
        const u32x4* qp = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + (lane & 31)) * D + (owner ? 0 : HALF)) + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[ks] = __builtin_bit_cast(bf16x8, qp[ks * 2]);
    }
    if (SPLIT && tid < 5) done_s[tid] = 0u;   // done + parked[4]
    if (SPLIT && tid == 5) next_s[3] = 0u;    // "a wave gave up waiting"
    if (owner && lane < 32) {
        const float tq = SAMPLE ? 0.f : a.tau[q0 + lane];
        tau_s[pair * 32 + lane] = tq;
        sim_s[pair * 32 + lane] = (1.0f - tq) - 4e-7f * (1.0f + __builtin_fabsf(tq));   // see batch_gemm_rega_kernel
        cnt_s[pair * 32 + lane] = 0u;
    }
    const uint32_t seg_slots = a.seg_area / blocks_per_group;
    const uint32_t seg_lane0 = (q0 + 4u * ((uint32_t)lane >> 5)) * a.cand_cap + a.seg_base + bidx * seg_slots;

    const uint32_t ntiles_all = (a.slab_rows + TROWS - 1) / TROWS;
    const uint32_t ntiles = SAMPLE ? a.sample_tiles : ntiles_all;
    const uint32_t slab_end = a.slab0 + a.slab_rows;
    const unsigned char* cbase = reinterpret_cast<const unsigned char*>(a.cb);
    auto phys = [&](uint32_t tile) -> uint32_t {
        return SAMPLE ? (uint32_t)(((unsigned long long)tile * ntiles_all) / a.sample_tiles) : tile;
    };

    // staging: 16 threads per tile row, a thread moves the 16-byte segments (tid & 15) + 16*p of its row
    const uint32_t srow = (uint32_t)tid >> 4;
    const uint32_t sseg = ((uint32_t)tid & 15u) * 16u;
    u32x4 regs[LOADS];
    auto issue_loads = [&](uint32_t tile) {
        uint32_t grow = a.slab0 + phys(tile) * TROWS + srow;
        grow = grow < a.n_rows ? grow : a.n_rows - 1;        // clamp: masked in the selection
        const unsigned char* src = cbase + (size_t)grow * (D * 2) + sseg;
#pragma unroll
        for (int p = 0; p < LOADS; ++p) regs[p] = *reinterpret_cast<const u32x4*>(src + p * 256);
    };
    auto store_tile = [&](unsigned char* buf) {
        unsigned char* dst = buf + srow * ROW_B + sseg;
#pragma unroll
        for (int p = 0; p < LOADS; ++p) *reinterpret_cast<u32x4*>(dst + p * 256) = regs[p];
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // this wave's K half of one tile: two independent accumulator chains, B fragments read AHEAD k-steps early
    auto mfma_tile = [&](const unsigned char* cur, unsigned char* st_dst) {   // st_dst: see batch_gemm_rega_kernel
        f32x16 a0, a1;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // first use of a chain: C = 0
        const unsigned char* b0 = cur + (lane & 31) * ROW_B + (owner ? 0 : HALF * 2) + (lane >> 5) * 16;
        constexpr int RING = AHEAD + 1;
        u32x4 fb[RING];
#pragma unroll
        for (int i = 0; i < AHEAD && i < KS; ++i) fb[i] = *reinterpret_cast<const u32x4*>(b0 + i * 32);
        if (!(a.debug & 32u)) __builtin_amdgcn_s_setprio(1);   // the pair's other wave is selecting: MFMA issue first (+3 %)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + AHEAD < KS) fb[(ks + AHEAD) % RING] = *reinterpret_cast<const u32x4*>(b0 + (ks + AHEAD) * 32);
            if (ks >= KS / 2 && ((ks - KS / 2) & 1) == 0 && (ks - KS / 2) / 2 < LOADS) {
                if (st_dst) *reinterpret_cast<u32x4*>(st_dst + ((ks - KS / 2) / 2) * 256) = regs[(ks - KS / 2) / 2];
            }
            if (ks & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fb[ks % RING]), ks == 1 ? zero16 : a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fb[ks % RING]), ks == 0 ? zero16 : a0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(a.debug & 32u)) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = a0[r] + a1[r];
    };
    auto park_partial = [&](uint32_t parity) {      // helper: 16 floats per lane, [j][lane] float4
        f32x4* dst = reinterpret_cast<f32x4*>(part0 + (parity * 4 + pair) * PART_B) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j * 64] = f32x4{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
    };
    // owner: add the helper's half, then the fused selection of batch_gemm_rega_kernel on one 32-row block
    auto select_tile = [&](uint32_t tile, uint32_t parity) {
        const f32x4* src = reinterpret_cast<const f32x4*>(part0 + (parity * 4 + pair) * PART_B) + lane;
        const lds_f32x4* sim_w = (const lds_f32x4*)(sim_s + pair * 32 + 4 * (lane >> 5));
        f32x4 lo[4], hp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { hp[j] = src[j * 64]; lo[j] = sim_w[2 * j]; }
        if (a.debug & 8u) return;
        unsigned long long hit[4] = {0ull, 0ull, 0ull, 0ull};
        float full[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            full[r] = acc[r] + hp[r >> 2][r & 3];
            hit[r >> 2] |= __ballot(full[r] >= lo[r >> 2][r & 3]);   // NaN fails
        }
        if (SAMPLE) {   // per-query best similarity of this 32-row tile (see batch_gemm_rega_kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = group_max32(full[r]);
                if ((lane & 31) == 31)
                    a.tile_max[(size_t)tile * (a.nqt * 128u) + q0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))] = m;
            }
            return;
        }
        if ((hit[0] | hit[1] | hit[2] | hit[3]) == 0ull || (a.debug & 64u)) return;   // debug bit6: hot test only (timing experiments)
        const uint32_t row0 = a.slab0 + tile * TROWS + (lane & 31);
        const bool ok0 = row0 < slab_end;
        const lds_f32* tau_w = (const lds_f32*)(tau_s + pair * 32);
        lds_u32* cnt_w = (lds_u32*)(cnt_s + pair * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (hit[g] == 0ull) continue;
            const f32x4 tau4 = *(const lds_f32x4*)(tau_w + 8 * g + 4 * (lane >> 5));
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                const int qo = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float d0 = (1.0f - full[r]) + 0.0f;
                if (ok0 && d0 <= tau4[r & 3]) {
                    const unsigned off = __hip_atomic_fetch_add(cnt_w + qo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t e0 = seg_lane0 + (uint32_t)((r & 3) + 8 * (r >> 2)) * a.cand_cap;
                    if (off < seg_slots) a.cand[e0 + off] = make_key(d0, a.row_base + row0);
                }
            }
        }
    };

    // SPLIT: see batch_gemm_rega_kernel (bounded spin; a wave that gives up poisons the workgroup's counts -> exact path)
    bool gave_up = SPLIT && (a.debug & 16384u) != 0 && blockIdx.x == 1 && wave == 3;   // debug bit 14: pretend one wave timed out (tests)
    auto wait_count = [&](const unsigned int* ctr, unsigned int target) {
        bool ok = false;
        for (unsigned int spins = 0; spins < (1u << 22); ++spins) {
            const unsigned int v = __hip_atomic_load((const lds_u32*)ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)v) >= target) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) gave_up = true;
        asm volatile("" ::: "memory");
    };
    auto signal_count = [&](unsigned int* ctr) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_u32*)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // tile order: static stride, or (a.tile_ctr) claimed from the group's counter — see batch_gemm_rega_kernel
    const bool dyn = !SAMPLE && !SPLIT && a.tile_ctr != nullptr;
    uint32_t t = bidx;
    if (dyn && tid == 0) {
        const unsigned int c0 = claim_tile_async(a.tile_ctr + group * 32u);
        claim_wait();
        next_s[0] = 2u * blocks_per_group + c0;
    }
    if (t < ntiles) {
        issue_loads(t);
        store_tile(buf0);
    }
    __syncthreads();
    uint32_t it = 0;
    uint32_t t_prev = 0;
    uint32_t t_next = t + blocks_per_group;
    for (; t < ntiles; ++it) {
        unsigned char* cur = buf0 + (it & 1u) * BUF_B;
        unsigned char* nxt = buf0 + ((it & 1u) ^ 1u) * BUF_B;
        const uint32_t tn = t_next;
        if (tn < ntiles) issue_loads(tn);
        uint32_t t_after;
        unsigned int claimed = 0;
        if (dyn) {
            t_after = next_s[it & 1u];
            if (tid == 0 && t_after < ntiles) claimed = claim_tile_async(a.tile_ctr + group * 32u);   // read at the end of the iteration
        } else {
            t_after = tn + blocks_per_group;
        }
        const bool spread = !(a.debug & 128u);                 // debug bit7: one burst after the K loop, for A/B timing
        unsigned char* st_dst = (spread && tn < ntiles) ? nxt + srow * ROW_B + sseg : nullptr;
        if (SPLIT) {
            if (owner) {
                if (it > 0) {
                    wait_count(parked_s + pair, it);            // the partner's partial sums of tile it - 1 are in LDS
                    select_tile(t_prev, (it - 1u) & 1u);
                    wait_count(done_s, 8u * it);                // every wave is through K loop it - 1: tile `it` stored, its other buffer free
                }
                mfma_tile(cur, st_dst);
                signal_count(done_s);
            } else {
                if (it > 0) wait_count(done_s, 8u * it);        // (also: the owner has read the partial buffer this tile's sums go to)
                mfma_tile(cur, st_dst);
                signal_count(done_s);
                park_partial(it & 1u);
                signal_count(parked_s + pair);
            }
        } else if (owner) {
            if (it > 0) select_tile(t_prev, (it - 1u) & 1u);   // partial of the previous tile: parked before the last barrier
            mfma_tile(cur, st_dst);
        } else {
            mfma_tile(cur, st_dst);
            park_partial(it & 1u);
        }
        if (!spread && tn < ntiles) store_tile(nxt);
        if (dyn && tid == 0) {
            claim_wait();
            next_s[(it + 1u) & 1u] = t_after < ntiles ? 2u * blocks_per_group + claimed : t_after;
        }
        if (!SPLIT) __syncthreads();
        t_prev = t;
        t = tn;
        t_next = t_after;
    }
    if (SPLIT && owner && it > 0) wait_count(parked_s + pair, it);
    if (owner && it > 0) select_tile(t_prev, (it - 1u) & 1u);
    if (SPLIT && gave_up && lane == 0) __hip_atomic_fetch_or((lds_u32*)(next_s + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (SPLIT && next_s[3] != 0u && tid < 128) cnt_s[tid] = 0x40000000u;   // nothing this workgroup selected can be trusted -> exact path
    if (SPLIT) __syncthreads();
    if (!SAMPLE && tid < 128) a.seg_count[(size_t)bidx * (a.nqt * 128u) + group * 128u + (uint32_t)tid] = cnt_s[tid];
}

// ---------------------------------------------------------------------------
// Register-resident-queries GEMM for D = 768, WHOLE K per wave ("wide": the default for 768 since round 4).
//
// The K-split kernel above is bound by the LDS port, not by the matrix pipe (profiles/r04: per 32-row tile a CU's eight waves
// issue 192 ds_read_b128 = 768 LDS cycles, the staging stores ~620 and the partial-sum exchange ~270, against 1 536 matrix
// cycles per SIMD): a tile staged in LDS is multiplied against only 128 queries. Here a wave keeps its 32 queries x 768 as
// 192 VGPRs of A fragments — which fits beside ONE 16-register accumulator once nothing else needs registers:
//   * staging is LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass — the DMA does not go through the
//     VGPR -> LDS transfer path that limits ds_write_b128 to ~79 B/clk), NBUF tile buffers, requested NBUF - 1 tiles ahead;
//   * one accumulator chain (a 32x32x16 MFMA accumulates back-to-back into the same registers without a stall; the other
//     wave of the SIMD fills the pipe between them anyway);
//   * the selection's bounds are read after the B-fragment ring has died.
// A workgroup is 8 waves = 256 queries (as for D <= 512), tiles are 32 rows: per tile a wave issues 48 MFMAs against
// 48 ds_read_b128 — the LDS reads per MFMA of the 384-d kernel — and a staged byte is used by twice as many queries as in
// the K-split kernel; no partial sums cross LDS. Per tile and CU: 3 072 matrix cycles per SIMD against 1 536 LDS cycles of
// fragment reads + 48.5 KB of DMA writes.
// The padded tile image (row stride 2D + 16 B: conflict-free ds_read_b128) is 48.5 KB; it is cut into 49 1-KB DMA pieces
// (lane l of piece P fetches what belongs at slot 64 P + l; a row's pad slot re-fetches its last segment; the upper half of
// the last piece lands in the buffer's 512 B of slack). Selection, survivor segments, thresholds, seg_count layout and the
// "late" order of waves 4-7 are those of batch_gemm_rega_kernel, so the host side and batch_finish_kernel do not know
// which kernel ran.
// SPLIT = true: the tile barrier is split as in batch_gemm_rega_kernel<..., SYNC = 2> — "arrive" (a wave adds 1 to an LDS counter
// behind its K loop, once its own DMA pieces of the next tile have landed) and "wait" (spin on the counter in front of the next
// K loop, BEFORE requesting the tile that will overwrite the buffer the previous K loop read): an early wave's selection sits
// between the two instead of in front of a workgroup barrier. The counter is touched through inline assembly only: an LDS
// access the compiler can see makes it drain the LDS-DMA queue first (s_waitcnt vmcnt(0)).
// (Issuing the next tile's DMA pieces from inside the K loop instead of in one burst at the top of the iteration was built and
// measured: 1 915 - 1 933 us against 1 807 - 1 856 us — the piece arithmetic does not fit beside 192 VGPRs of A fragments without
// spilling, and the burst was not the idle time it looked like; profiles/HISTORY.md.)
template <int D, int NBUF, int AHEAD, bool SAMPLE = false, int CHAINS = 1, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void batch_gemm_wide_kernel(GemmArgs a, uint32_t blocks_per_group) {
    constexpr int KS = D / 16;                       // MFMA k-steps
    constexpr int ROW_B = D * 2 + 16;                // LDS row stride (bytes); (ROW_B / 4) % 64 == 4
    constexpr int TROWS = 32;                        // corpus rows per tile
    constexpr int SEG_PER_ROW = D * 2 / 16;          // 16-byte segments per row
    constexpr int SLOTS_PER_ROW = ROW_B / 16;        // 16-byte slots per padded row
    constexpr int IMG_B = TROWS * ROW_B;             // padded tile image
    constexpr int PIECES = (IMG_B + 1023) / 1024;    // 1-KB DMA pieces per tile
    constexpr int BUF_B = PIECES * 1024;             // buffer stride (the image + slack for the last piece)
    constexpr int PPW = (PIECES + 7) / 8;            // pieces per wave (waves with index >= PIECES % 8 carry one less)
    constexpr int PRE = NBUF - 1;                    // tiles requested ahead of the one being read
    static_assert(D % 64 == 0 && (ROW_B / 4) % 64 == 4, "row stride must keep ds_read_b128 conflict-free");
    static_assert(NBUF * BUF_B + 3 * 8 * 32 * 4 + 64 <= 160 * 1024, "LDS budget of one CU");
    static_assert(PIECES % 8 != 0, "dma_wait assumes a ragged split (wave 0 carries one piece more)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tau_s = reinterpret_cast<float*>(smem + NBUF * BUF_B);              // [8][32] exact thresholds
    unsigned int* cnt_s = reinterpret_cast<unsigned int*>(tau_s + 8 * 32);     // [8][32] survivors per query (this workgroup)
    float* sim_s = reinterpret_cast<float*>(cnt_s + 8 * 32);                   // [8][32] conservative similarity bounds
    unsigned int* sync_s = reinterpret_cast<unsigned int*>(sim_s + 8 * 32);    // SPLIT: [0] arrivals, [1] "a wave gave up waiting"

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    // readfirstlane: tells hipcc the wave index is wave-uniform, so everything derived from it — the early / late order, and with
    // it the survivor counters modified under that branch — stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t group = blockIdx.x / blocks_per_group;   // 256 queries per group
    const uint32_t bidx = blockIdx.x % blocks_per_group;
    const uint32_t q0 = group * 256 + wave * 32;             // this wave's 32 queries

    // A fragments: lane l holds query (l & 31), k = 16*ks + 8*(l >> 5) .. +7
    bf16x8 fa[KS];
    {
        const u32x4* qp = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + (lane & 31)) * D) + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[ks] = __builtin_bit_cast(bf16x8, qp[ks * 2]);
    }
    if (!SAMPLE && lane < 32) {
        const float tq = a.tau[q0 + lane];
        tau_s[wave * 32 + lane] = tq;
        sim_s[wave * 32 + lane] = (1.0f - tq) - 4e-7f * (1.0f + __builtin_fabsf(tq));   // see batch_gemm_rega_kernel
    }
    if (tid < 256) cnt_s[tid] = 0u;
    if (SPLIT && tid < 2) sync_s[tid] = 0u;
    const uint32_t seg_slots = a.seg_area / blocks_per_group;
    const uint32_t seg_lane0 = (q0 + 4u * ((uint32_t)lane >> 5)) * a.cand_cap + a.seg_base + bidx * seg_slots;

    const uint32_t ntiles_all = (a.slab_rows + TROWS - 1) / TROWS;
    const uint32_t ntiles = SAMPLE ? a.sample_tiles : ntiles_all;
    const uint32_t slab_end = a.slab0 + a.slab_rows;
    const unsigned char* cbase = reinterpret_cast<const unsigned char*>(a.cb);
    auto phys = [&](uint32_t tile) -> uint32_t {
        return SAMPLE ? (uint32_t)(((unsigned long long)tile * ntiles_all) / a.sample_tiles) : tile;
    };

    // LDS-DMA map, recomputed per piece (a handful of VALU against 48 MFMAs; tables would cost 2 * PPW VGPRs):
    // wave w moves pieces w, w + 8, ...; lane l of piece P fills slot P*64 + l of the padded image
    const bool full_wave = (uint32_t)wave < (uint32_t)(PIECES % 8);          // wave-uniform: carries PPW pieces, the others PPW - 1
    // piece i of this wave for the tile whose first row is row0, into the buffer at buf_off. The lane index is made opaque per
    // piece: left to itself hipcc hoists the loop-invariant slot -> (row, column) arithmetic of all PPW pieces out of the tile loop —
    // 3 VGPRs per piece the A fragments have no room for (they were spilled to scratch, and every reload drained the DMA queue
    // with an s_waitcnt vmcnt(0))
    auto dma_piece = [&](int i, uint32_t row0, uint32_t buf_off) {
        if (i < PPW - 1 || full_wave) {
            uint32_t lane_o = (uint32_t)lane;
            asm volatile("" : "+v"(lane_o));
            const uint32_t P = (uint32_t)wave + 8u * (uint32_t)i;
            const uint32_t slot = P * 64u + lane_o;
            uint32_t r = slot / (uint32_t)SLOTS_PER_ROW;
            uint32_t c = slot - r * (uint32_t)SLOTS_PER_ROW;
            c = c < (uint32_t)SEG_PER_ROW ? c : (uint32_t)SEG_PER_ROW - 1u;   // pad slot: any valid bytes
            r = r < (uint32_t)TROWS ? r : (uint32_t)TROWS - 1u;               // slack behind the image: any valid bytes
            uint32_t grow = row0 + r;
            grow = grow < a.n_rows ? grow : a.n_rows - 1;                     // clamp: masked in the selection
            const unsigned char* src = cbase + (size_t)grow * (D * 2) + c * 16u;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(smem + buf_off + P * 1024u), 16, 0, 0);
        }
    };
    auto dma_tile = [&](uint32_t tile, uint32_t buf_off) {
        const uint32_t row0 = a.slab0 + phys(tile) * TROWS;
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(i, row0, buf_off);
    };
    // this wave's DMA requests still allowed in flight: one whole tile, or none
    auto dma_wait = [&](bool keep_one_tile) {
        if (!keep_one_tile) wait_vmcnt<0>();
        else if (full_wave) wait_vmcnt<PPW>();
        else wait_vmcnt<PPW - 1>();
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const bool prio = (a.debug & 32u) == 0;
    // K loop: see batch_gemm_rega_kernel (B fragments read AHEAD k-steps early, every step pinned by a sched_barrier)
    auto mfma_tile = [&](const unsigned char* cur) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned char* b0 = cur + (lane & 31) * ROW_B + (lane >> 5) * 16;
        constexpr int RING = AHEAD + 1;
        u32x4 fb[RING];
        f32x16 a1;
#pragma unroll
        for (int i = 0; i < AHEAD && i < KS; ++i) fb[i] = *reinterpret_cast<const u32x4*>(b0 + i * 32);
        if (prio) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + AHEAD < KS) fb[(ks + AHEAD) % RING] = *reinterpret_cast<const u32x4*>(b0 + (ks + AHEAD) * 32);
            if (CHAINS == 2 && (ks & 1))
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fb[ks % RING]), ks == 1 ? zero16 : a1, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fb[ks % RING]), ks == 0 ? zero16 : acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (CHAINS == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += a1[r];
        }
    };
    // Fused selection on one 32-row block. Hot path as in batch_gemm_rega_kernel (16 compares against conservative bounds, wave-level
    // flags per group of four queries). The cold path differs: a wave's 32 queries are ITS OWN (no other wave of the workgroup
    // selects for them), so the per-(workgroup, query) survivor counters live in 32 SGPRs of the wave instead of in LDS, and a
    // survivor's slot is counter + (passing lanes below it in its half-wave) — ballot, mbcnt, s_bcnt1: no LDS atomic round trip, no
    // threshold read. Rows are admitted on the conservative bound itself (acc >= sim_lo, a superset of `1 - acc <= tau` by less
    // than 1e-6: rejected rows still have d > tau, which is all the certificate uses). With the eight waves joined by a barrier
    // per tile, an LDS round trip taken by ANY wave (some wave has a survivor in ~9 of 10 tiles) was paid by all of them:
    // 175 - 215 us of a 1 770 us launch (profiles/r04).
    // (two 16-bit counters per SGPR — 32 full ones do not fit beside the kernel's other scalars and hipcc then keeps them in
    // VGPRs it has to spill — saturating at 0xFFFF, which is reported as an overflow: the query takes the exact path)
    unsigned cq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cq[r] = 0u;
    auto select_tile = [&](uint32_t tile) {
        if (a.debug & 8u) return;
        if (SAMPLE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = group_max32(acc[r]);
                if ((lane & 31) == 31)
                    a.tile_max[(size_t)tile * (a.nqt * 128u) + q0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))] = m;
            }
            return;
        }
        const lds_f32x4* sim_w = (const lds_f32x4*)(sim_s + wave * 32 + 4 * (lane >> 5));
        f32x4 lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = sim_w[2 * j];
        unsigned long long hit[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int r = 0; r < 16; ++r) hit[r >> 2] |= __ballot(acc[r] >= lo[r >> 2][r & 3]);   // NaN fails
        if ((hit[0] | hit[1] | hit[2] | hit[3]) == 0ull || (a.debug & 64u)) return;
        const uint32_t row0 = a.slab0 + tile * TROWS + (lane & 31);
        const bool ok0 = row0 < slab_end;
        uint32_t seg_o = seg_lane0;                   // opaque: the 16 per-query row offsets are computed HERE (cold path), not
        asm volatile("" : "+v"(seg_o));               // hoisted out of the tile loop into 16 VGPRs that do not exist
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (hit[g] == 0ull) continue;
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                const bool p = ok0 && acc[r] >= lo[r >> 2][r & 3];
                const unsigned long long m = __ballot(p);
                if (m == 0ull) continue;
                const unsigned n_lo = (unsigned)__builtin_popcount((unsigned)m), n_hi = (unsigned)__builtin_popcount((unsigned)(m >> 32));
                const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                unsigned c0 = cq[r] & 0xFFFFu, c1 = cq[r] >> 16;
                const unsigned off = lane < 32 ? c0 + below : c1 + (below - n_lo);
                if (p && off < seg_slots && !(a.debug & 8192u)) {   // debug bit 13 (timing only): everything but the store
                    const uint32_t e0 = seg_o + (uint32_t)((r & 3) + 8 * (r >> 2)) * a.cand_cap;
                    a.cand[e0 + off] = make_key((1.0f - acc[r]) + 0.0f, a.row_base + row0);
                }
                c0 = c0 + n_lo < 0xFFFFu ? c0 + n_lo : 0xFFFFu;   // not clamped to seg_slots: a count above it tells the finish kernel that survivors were dropped
                c1 = c1 + n_hi < 0xFFFFu ? c1 + n_hi : 0xFFFFu;
                cq[r] = c0 | (c1 << 16);
            }
        }
    };

    // SPLIT: bounded spin on the arrival counter (see batch_gemm_rega_kernel: a wave that gives up poisons the workgroup's
    // survivor counts, which sends its queries to the exact path — never a silent wrong answer)
    bool gave_up = SPLIT && (a.debug & 16384u) != 0 && blockIdx.x == 1 && wave == 3;   // debug bit 14: pretend one wave timed out (tests)
    const unsigned sync_addr = (unsigned)(size_t)(lds_u32*)sync_s;
    auto wait_arrivals = [&](unsigned int target) {
        bool ok = false;
        for (unsigned int spins = 0; spins < (1u << 22); ++spins) {
            unsigned int v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(sync_addr) : "memory");
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)v) >= target) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) gave_up = true;
    };
    auto arrive = [&]() {
        if (lane == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_u32 %0, %1" ::"v"(sync_addr), "v"(1u) : "memory");
    };

    uint32_t t = bidx;
    {
        bool second = false;
        if (t < ntiles) dma_tile(t, 0u);
        if (PRE == 2 && t + blocks_per_group < ntiles) { dma_tile(t + blocks_per_group, (uint32_t)BUF_B); second = true; }
        dma_wait(second);
        __builtin_amdgcn_s_barrier();                         // also publishes tau_s / cnt_s / sim_s
        asm volatile("" ::: "memory");
    }
    const bool late = wave >= 4 && !(a.debug & 16u);          // see batch_gemm_rega_kernel: the two waves of a SIMD work in opposite order
    // Pace gate (advisory). With G > 1 query groups every corpus tile is wanted G times, by the G workgroups that share a `bidx` —
    // they sit on the same XCD (block -> XCD is blockIdx % 8 and G * blocks_per_group = 256), so the second to G-th reader hit in its
    // L2 as long as the groups stay within a few tiles of each other. Over a launch of milliseconds they do not (different survivor
    // loads): config 5 whole fetched 1.13 - 1.64 x the mirror (FETCH_SIZE, profiles/r04). Every GATE_EVERY tiles wave 0 adds its
    // progress to a word shared by the G workgroups of its bidx and, if it is more than GATE_WINDOW tiles ahead of their average,
    // sleeps until they catch up — the other seven waves wait for it at the tile barrier. Bounded and advisory: a timeout (a group
    // that started late behind another kernel) just proceeds; a workgroup that leaves the loop credits the word so that nobody
    // waits for it.
    constexpr uint32_t GATE_EVERY = 8u, GATE_WINDOW = 6u;
    const uint32_t ngroups = gridDim.x / blocks_per_group;
    const bool gate = !SAMPLE && a.progress != nullptr && ngroups > 1u && (blocks_per_group & 7u) == 0u && ngroups * blocks_per_group <= 256u && !(a.debug & 4096u);
    const uint32_t* gate_word = gate ? a.progress + (bidx & 7u) * 32u + (bidx >> 3) : a.progress;   // one word per bidx, one 128-byte line per XCD
    uint32_t it = 0, cur_idx = 0, t_prev = 0;
    for (; t < ntiles; ++it) {
        if (gate && wave == 0 && (it & (GATE_EVERY - 1u)) == 0u && it > 0u) {
            // (returning atomics: the value comes from wherever agent-scope atomics execute, never from a stale cache line)
            unsigned int add = GATE_EVERY;
            for (uint32_t spins = 0; spins < 1024u; ++spins) {
                unsigned int total = 0u;
                if (lane == 0) total = __hip_atomic_fetch_add(const_cast<uint32_t*>(gate_word), add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
                total = (unsigned int)__builtin_amdgcn_readfirstlane((int)total);
                add = 0u;
                // ahead of the average by more than the window?  G * it - total > G * WINDOW   (total counts tiles of all G groups)
                if ((int)(ngroups * it - total) <= (int)(ngroups * GATE_WINDOW)) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        const unsigned char* cur = smem + cur_idx * BUF_B;
        const uint32_t tn = t + PRE * blocks_per_group;
        uint32_t pre_idx = cur_idx + PRE;
        pre_idx = pre_idx >= (uint32_t)NBUF ? pre_idx - NBUF : pre_idx;
        bool issued = false;
        if (SPLIT) {
            if (late && it > 0) select_tile(t_prev);
            // every wave is through K loop it - 1 (the buffer tile tn goes to is free) and has its pieces of tile `it` in LDS
            if (it > 0) wait_arrivals(8u * it);
            if (tn < ntiles) { dma_tile(tn, pre_idx * BUF_B); issued = true; }
            mfma_tile(cur);
            dma_wait(PRE == 2 && issued);
            arrive();
            if (!late) select_tile(t);
            t_prev = t;
            cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
            t += blocks_per_group;
            continue;
        }
        if (tn < ntiles) { dma_tile(tn, pre_idx * BUF_B); issued = true; }
        // tile t + 1 must have landed before the barrier (every wave waits for its own pieces, the barrier joins them); with
        // three tiles in LDS the one requested in this iteration stays in flight across it. The wait sits in FRONT of an early
        // wave's selection: vmcnt counts the selection's survivor stores too, and a store issued just before the wait put a
        // global-memory round trip on the barrier's critical path in ~9 of 10 tiles (some wave of the eight has a survivor).
        if (late) {
            if (it > 0) select_tile(t_prev);
            mfma_tile(cur);
            dma_wait(PRE == 2 && issued);
        } else {
            mfma_tile(cur);
            dma_wait(PRE == 2 && issued);
            select_tile(t);
        }
        t_prev = t;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
        t += blocks_per_group;
    }
    if (late && it > 0) select_tile(t_prev);
    if (gate && tid == 0) __hip_atomic_fetch_add(const_cast<uint32_t*>(gate_word), 1u << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // done: nobody waits for this group
    if (SPLIT && gave_up && lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(sync_addr + 4u), "v"(1u) : "memory");
    if (!SAMPLE) {   // the wave's 32 survivor counters, SGPRs -> LDS: lane q takes query q's
        unsigned mine = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mine = lane == (r & 3) + 8 * (r >> 2) ? (cq[r] & 0xFFFFu) : mine;
            mine = lane == (r & 3) + 8 * (r >> 2) + 4 ? (cq[r] >> 16) : mine;
        }
        if (lane < 32) cnt_s[wave * 32 + lane] = mine == 0xFFFFu ? 0x40000000u : mine;   // saturated = unknown = overflowed
    }
    __syncthreads();
    if (!SAMPLE && tid < 256) {
        const bool poisoned = SPLIT && sync_s[1] != 0u;       // a wave gave up waiting: nothing this workgroup selected can be trusted
        a.seg_count[(size_t)bidx * (a.nqt * 128u) + group * 256u + (uint32_t)tid] = poisoned ? 0x40000000u : cnt_s[tid];
    }
}

// ---------------------------------------------------------------------------
// Register-resident-queries GEMM, PING-PONG schedule (round 5; D * TROWS = 24 576: 768-d x 32 rows, 384-d x 64 rows, ...).
//
// Same data path as batch_gemm_wide_kernel — 32 queries x D per wave as MFMA A fragments, corpus tiles by LDS-DMA into a ring of
// NBUF padded tile images, one ds_read_b128 per MFMA, SGPR survivor counters — with two differences that the round-4 counters
// asked for (matrix pipe 55 % busy, waves parked 48 %, no bank conflicts, traffic 1.00 x):
//
//  1. NOTHING in the tile loop is an LDS access the compiler can see. hipcc's waitcnt pass cannot tell which LDS bytes a pending
//     global_load_lds will write, so in front of the first B-fragment read of every tile it emitted `s_waitcnt vmcnt(0)` — the wide
//     kernel issued tile t + 2's DMA and then WAITED FOR IT TO LAND (an L2 / HBM round trip per tile, on every wave) before its
//     first MFMA: the three-buffer ring never had anything in flight across a tile (rocm 7.2 ISA of the round-4 build:
//     `s_waitcnt vmcnt(0) lgkmcnt(0)` ahead of MFMA 0 of the K loop). Here the B fragments and the selection's bounds are read by
//     inline-asm ds_read_b128 with hand-counted lgkmcnt waits (LDS operations return in order: "at most N younger ones outstanding"
//     implies that read f has landed, whatever else is queued), and the only vmcnt waits are the counted ones below.
//  2. The two waves of a SIMD are HALF A TILE PERIOD apart (PING): waves 0-3 multiply tile t (48 MFMAs back to back, s_setprio 1)
//     while waves 4-7 select tile t - 1 and request their pieces of tile t + PRE; an s_barrier; then waves 4-7 multiply tile t while
//     waves 0-3 request their pieces, wait for tile t + 1 and select tile t; an s_barrier. A SIMD's matrix pipe always has exactly
//     one wave feeding it, and everything that is not an MFMA (DMA address arithmetic, the selection with its cold path and global
//     stores, the waits) runs in the shadow of the partner's MFMAs instead of at the tile boundary where BOTH waves used to do it
//     with the pipe idle. Two barriers per tile, but no wave ever arrives at one with matrix work pending behind it.
//     Hazards: tile t + PRE goes to the buffer tile t - 1 was read from; its last reader (a late wave's K loop of period t - 1)
//     ended before the barrier that opens period t, and both halves request after that barrier. A wave waits for its OWN pieces of
//     tile t + 1 (counted vmcnt: the PRE - 1 younger tiles stay in flight; survivor stores are older than those and complete first)
//     before the barrier that ends period t; readers start behind that barrier.
// PING = false keeps one barrier per tile and the early / late order of the wide kernel (A/B of item 1 alone).
template <int D, int TROWS, int NBUF, int AHEAD, bool SAMPLE = false, int MODE = 2>
__global__ __launch_bounds__(512, 2) void batch_gemm_pp_kernel(GemmArgs a, uint32_t blocks_per_group) {
    constexpr bool PING = MODE >= 1 && MODE <= 3;    // two barriers per tile, the halves half a period apart
    constexpr bool LATE_DMA = MODE == 2 || MODE == 3;   // only waves 4-7 request tiles; the late half's fragment ring is primed ahead of the MID barrier
    constexpr bool EARLY_HEAD = MODE == 2;           // ... and the early half's ahead of the END barrier (its ring is then live around the loop)
    constexpr bool EARLY_DMA_AFTER = MODE == 4;      // one barrier per tile; the early half multiplies FIRST and requests tile t + PRE behind its K loop
    constexpr bool SPLIT = MODE == 5;                // one SPLIT barrier per tile (arrive behind the K loop, wait in front of the next): selection between the two
    constexpr int KS = D / 16;                       // MFMA k-steps
    constexpr int RB = TROWS / 32;                   // 32-row blocks per tile = accumulators per wave
    constexpr int NF = KS * RB;                      // B fragments (= MFMAs) per wave and tile; fragment f = (k-step f / RB, block f % RB)
    constexpr int ROW_B = D * 2 + 16;                // LDS row stride (bytes); (ROW_B / 4) % 64 == 4
    constexpr int SEG_PER_ROW = D * 2 / 16;          // 16-byte segments per row
    constexpr int SLOTS_PER_ROW = ROW_B / 16;        // 16-byte slots per padded row
    constexpr int IMG_B = TROWS * ROW_B;             // padded tile image
    constexpr int PIECES = (IMG_B + 1023) / 1024;    // 1-KB DMA pieces per tile
    constexpr int BUF_B = PIECES * 1024;             // buffer stride (the image + slack for the last piece)
    constexpr int PRE = NBUF - 1;                    // tiles requested ahead of the one being read
    constexpr int RING = AHEAD + 1;
    static_assert(TROWS % 32 == 0 && RB >= 1 && RB <= 4, "tile = 1..4 MFMA row blocks");
    static_assert(D % 64 == 0 && (ROW_B / 4) % 64 == 4, "row stride must keep ds_read_b128 conflict-free");
    static_assert(NBUF >= 3, "a tile is requested at least two periods before it is read");
    static_assert(NBUF * BUF_B + 2 * 8 * 32 * 4 + 64 <= 160 * 1024, "LDS budget of one CU");
    static_assert((RB - 1) * 32 * ROW_B + (KS - 1) * 32 < 65536, "ds_read offset field");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned int* cnt_s = reinterpret_cast<unsigned int*>(smem + NBUF * BUF_B);    // [8][32] survivors per query (this workgroup; written once, at the end)
    float* sim_s = reinterpret_cast<float*>(cnt_s + 8 * 32);                       // [8][32] conservative similarity bounds
    unsigned int* sync_s = reinterpret_cast<unsigned int*>(sim_s + 8 * 32);        // SPLIT: [0] arrivals, [1] "a wave gave up waiting"

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the early / late split and the survivor counters stay scalar
    const uint32_t group = blockIdx.x / blocks_per_group;        // 256 queries per group
    const uint32_t bidx = blockIdx.x % blocks_per_group;
    const uint32_t q0 = group * 256 + wave * 32;                 // this wave's 32 queries

    // A fragments: lane l holds query (l & 31), k = 16*ks + 8*(l >> 5) .. +7
    bf16x8 fa[KS];
    {
        const u32x4* qp = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + (lane & 31)) * D) + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[ks] = __builtin_bit_cast(bf16x8, qp[ks * 2]);
        // The fragments are "used" HERE, so hipcc waits for their loads here — before the first DMA request. Left alone it places
        // that wait in front of the first MFMA of the tile loop, where the only count it can prove is vmcnt(0): every wave then
        // drains its whole DMA queue (the tile it requested a moment ago included) at the top of every K loop.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa[ks]));
    }
    if (!SAMPLE && lane < 32) {
        const float tq = a.tau[q0 + lane];
        sim_s[wave * 32 + lane] = (1.0f - tq) - 4e-7f * (1.0f + __builtin_fabsf(tq));   // see batch_gemm_rega_kernel
    }
    if (SPLIT && tid < 2) sync_s[tid] = 0u;
    const uint32_t seg_slots = a.seg_area / blocks_per_group;
    const uint32_t seg_lane0 = (q0 + 4u * ((uint32_t)lane >> 5)) * a.cand_cap + a.seg_base + bidx * seg_slots;

    const uint32_t ntiles_all = (a.slab_rows + TROWS - 1) / TROWS;
    const uint32_t ntiles = SAMPLE ? a.sample_tiles : ntiles_all;
    const uint32_t slab_end = a.slab0 + a.slab_rows;
    const unsigned char* cbase = reinterpret_cast<const unsigned char*>(a.cb);
    auto phys = [&](uint32_t tile) -> uint32_t {
        return SAMPLE ? (uint32_t)(((unsigned long long)tile * ntiles_all) / a.sample_tiles) : tile;
    };

    // LDS-DMA map (as in batch_gemm_wide_kernel): wave w moves pieces w, w + 8, ...; lane l of piece P fills slot P*64 + l of
    // the padded image; the arithmetic is redone per piece behind an opaque lane index so that it is not hoisted into VGPRs
    // LATE_DMA: the four late waves carry all of it (pieces w - 4, w, w + 4, ...)
    constexpr int DW = LATE_DMA ? 4 : 8;                                       // waves that request
    constexpr int PPWD = (PIECES + DW - 1) / DW;                               // pieces per requesting wave
    constexpr int FULLD = PIECES % DW == 0 ? DW : PIECES % DW;                 // requesting waves below this index carry PPWD pieces
    const int dwave = LATE_DMA ? wave - 4 : wave;
    const bool full_wave = dwave >= 0 && dwave < FULLD;
    auto dma_piece = [&](int i, uint32_t row0, uint32_t buf_off) {
        if (i < PPWD - 1 || full_wave) {
            uint32_t lane_o = (uint32_t)lane;
            asm volatile("" : "+v"(lane_o));
            const uint32_t P = (uint32_t)dwave + (uint32_t)DW * (uint32_t)i;
            const uint32_t slot = P * 64u + lane_o;
            uint32_t r = slot / (uint32_t)SLOTS_PER_ROW;
            uint32_t c = slot - r * (uint32_t)SLOTS_PER_ROW;
            c = c < (uint32_t)SEG_PER_ROW ? c : (uint32_t)SEG_PER_ROW - 1u;   // pad slot: any valid bytes
            r = r < (uint32_t)TROWS ? r : (uint32_t)TROWS - 1u;               // slack behind the image: any valid bytes
            uint32_t grow = row0 + r;
            grow = grow < a.n_rows ? grow : a.n_rows - 1;                     // clamp: masked in the selection
            const unsigned char* src = cbase + (size_t)grow * (D * 2) + c * 16u;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(smem + buf_off + P * 1024u), 16, 0, 0);
        }
    };
    // Filtering launch: no clamps. A pad slot fetches the first 16 bytes of the next row, the slack behind the image the row after the
    // tile, a tile that runs past the store whatever the mirror holds there — all of it inside the mirror's allocation (capacity + 64
    // slack rows, ensure_mirror) and none of it ever used (pad bytes are never read as fragments, rows >= slab_end are masked by the
    // selection; a garbage row only pollutes its own output column). Then slot -> global offset is 16 * (slot - slot / SLOTS_PER_ROW)
    // from a uniform tile base: a multiply, a shift, a subtract and a shift per piece (the general form above costs ~14 VALU plus
    // the spilled-SGPR traffic of its clamps).
    constexpr uint32_t DIV_SHIFT = 18, DIV_MAGIC = ((1u << DIV_SHIFT) + SLOTS_PER_ROW - 1) / SLOTS_PER_ROW;
    static_assert((uint64_t)(PIECES * 64) * (DIV_MAGIC * (uint64_t)SLOTS_PER_ROW - (1u << DIV_SHIFT)) < (1u << DIV_SHIFT),
                  "magic division must be exact for every slot of a tile");
    // Where the A fragments leave room (D <= 512) the per-piece offsets are computed once and stay in VGPRs: a request is then ONE
    // instruction (saddr form: uniform tile base + the lane's 32-bit offset).
    constexpr bool CACHE_OFF = !SAMPLE && D <= 512;
    uint32_t doff[CACHE_OFF ? PPWD : 1];
    if (CACHE_OFF) {
#pragma unroll
        for (int i = 0; i < PPWD; ++i) {
            const uint32_t slot = ((uint32_t)(dwave < 0 ? 0 : dwave) + (uint32_t)DW * (uint32_t)i) * 64u + (uint32_t)lane;
            doff[i] = (slot - ((slot * DIV_MAGIC) >> DIV_SHIFT)) * 16u;
        }
    }
    auto dma_tile = [&](uint32_t tile, uint32_t buf_off) {
        if (a.debug & 1u) return;                    // timing only: no corpus stream (the K loop reads whatever is in LDS)
        if (SAMPLE) {
            const uint32_t row0 = a.slab0 + phys(tile) * TROWS;
#pragma unroll
            for (int i = 0; i < PPWD; ++i) dma_piece(i, row0, buf_off);
            return;
        }
        const unsigned char* tbase = cbase + (size_t)(a.slab0 + tile * TROWS) * (D * 2);   // wave-uniform
#pragma unroll
        for (int i = 0; i < PPWD; ++i) {
            if (i < PPWD - 1 || full_wave) {
                const uint32_t P = (uint32_t)dwave + (uint32_t)DW * (uint32_t)i;
                uint32_t off;
                if constexpr (CACHE_OFF) {
                    off = doff[i];
                } else {
                    uint32_t lane_o = (uint32_t)lane;
                    asm volatile("" : "+v"(lane_o));  // (recomputed per piece: no per-piece VGPRs live around the tile loop)
                    const uint32_t slot = P * 64u + lane_o;
                    off = (slot - ((slot * DIV_MAGIC) >> DIV_SHIFT)) * 16u;
                }
                __builtin_amdgcn_global_load_lds((global_cvoid*)(tbase + off), (lds_void*)(smem + buf_off + P * 1024u), 16, 0, 0);
            }
        }
    };
    // this wave's DMA requests still allowed in flight: PRE - 1 whole tiles, or none
    auto dma_wait = [&](bool keep) {
        if (!keep) wait_vmcnt<0>();
        else if (full_wave) wait_vmcnt<(PRE - 1) * PPWD>();
        else wait_vmcnt<(PRE - 1) * (PPWD - 1)>();
    };

    f32x16 acc[RB];
    u32x4 fb[RING];                                  // B-fragment ring (kernel scope: LATE_DMA fills its head ahead of a barrier)
    const bool prio = (a.debug & 32u) == 0;
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_void*)smem;
    const uint32_t lane_boff = (uint32_t)(lane & 31) * (uint32_t)ROW_B + (uint32_t)(lane >> 5) * 16u;
    // K loop. Fragment f + AHEAD is requested before MFMA f; the wait in front of MFMA f leaves at most min(AHEAD, NF - 1 - f)
    // younger reads outstanding. Every step is pinned (sched_barrier) — hipcc otherwise hoists an MFMA over the asm wait it depends on.
    auto mfma_head = [&](uint32_t baddr) {           // request fragments 0 .. AHEAD - 1
        static_for<0, (AHEAD < NF ? AHEAD : NF)>([&](auto F) {
            constexpr int f = decltype(F)::value;
            u32x4(&fbr)[RING] = fb;                  // (asm operands alone do not make a generic lambda capture)
            const uint32_t ba = baddr;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbr[f % RING]) : "v"(ba), "n"((f % RB) * 32 * ROW_B + (f / RB) * 32));
        });
    };
    auto mfma_body = [&](uint32_t baddr) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a.debug & 2u) {                          // timing only: no K loop (the ring's head is retired, the accumulators defined)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < RB; ++b) acc[b] = zero16;
            return;
        }
        if (prio) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NF>([&](auto F) {
            constexpr int f = decltype(F)::value;
            constexpr int g = f + AHEAD;
            u32x4(&fbr)[RING] = fb;
            const uint32_t ba = baddr;
            if constexpr (g < NF)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbr[g % RING]) : "v"(ba), "n"((g % RB) * 32 * ROW_B + (g / RB) * 32));
            constexpr int younger = (NF - 1 - f) < AHEAD ? (NF - 1 - f) : AHEAD;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr int ks = f / RB, rb = f % RB;
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], __builtin_bit_cast(bf16x8, fbr[f % RING]), ks == 0 ? zero16 : acc[rb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (prio) __builtin_amdgcn_s_setprio(0);
    };
    auto mfma_tile = [&](uint32_t baddr) {
        mfma_head(baddr);
        mfma_body(baddr);
    };
    // Fused selection, see batch_gemm_wide_kernel (hot test against conservative bounds, wave-level flags per four queries; cold
    // path with the wave's 32 per-query survivor counters in SGPRs, two saturating 16-bit counters each). The bounds are read by
    // inline asm (point 1 above).
    unsigned cq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cq[r] = 0u;
    const uint32_t sim_addr = (uint32_t)(size_t)(lds_void*)sim_s + (uint32_t)(wave * 32 + 4 * (lane >> 5)) * 4u;
    auto select_tile = [&](uint32_t tile) {
        if (a.debug & 8u) return;
        if (SAMPLE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[0][r];
#pragma unroll
                for (int b = 1; b < RB; ++b) v = __builtin_fmaxf(v, acc[b][r]);   // rows past the store are clamped copies of its last row
                const float m = group_max32(v);
                if ((lane & 31) == 31)
                    a.tile_max[(size_t)tile * (a.nqt * 128u) + q0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))] = m;
            }
            return;
        }
        f32x4 lo[4];
        static_for<0, 4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            f32x4(&lor)[4] = lo;
            const uint32_t sa = sim_addr;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lor[j]) : "v"(sa), "n"(j * 32));
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long hit[RB][4];
        unsigned long long any = 0ull;
#pragma unroll
        for (int b = 0; b < RB; ++b) {
#pragma unroll
            for (int g = 0; g < 4; ++g) hit[b][g] = 0ull;
#pragma unroll
            for (int r = 0; r < 16; ++r) hit[b][r >> 2] |= __ballot(acc[b][r] >= lo[r >> 2][r & 3]);   // NaN fails
#pragma unroll
            for (int g = 0; g < 4; ++g) any |= hit[b][g];
        }
        if (any == 0ull || (a.debug & 64u)) return;
        uint32_t seg_o = seg_lane0;                   // opaque: the per-query row offsets are computed HERE (cold path)
        asm volatile("" : "+v"(seg_o));
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const uint32_t row0 = a.slab0 + tile * TROWS + (uint32_t)(b * 32) + (uint32_t)(lane & 31);
            const bool ok0 = row0 < slab_end;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (hit[b][g] == 0ull) continue;
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) {
                    const bool p = ok0 && acc[b][r] >= lo[r >> 2][r & 3];
                    const unsigned long long m = __ballot(p);
                    if (m == 0ull) continue;
                    const unsigned n_lo = (unsigned)__builtin_popcount((unsigned)m), n_hi = (unsigned)__builtin_popcount((unsigned)(m >> 32));
                    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    unsigned c0 = cq[r] & 0xFFFFu, c1 = cq[r] >> 16;
                    const unsigned off = lane < 32 ? c0 + below : c1 + (below - n_lo);
                    if (p && off < seg_slots) {
                        const uint32_t e0 = seg_o + (uint32_t)((r & 3) + 8 * (r >> 2)) * a.cand_cap;
                        a.cand[e0 + off] = make_key((1.0f - acc[b][r]) + 0.0f, a.row_base + row0);
                    }
                    c0 = c0 + n_lo < 0xFFFFu ? c0 + n_lo : 0xFFFFu;   // not clamped to seg_slots: a count above it tells the finish kernel that survivors were dropped
                    c1 = c1 + n_hi < 0xFFFFu ? c1 + n_hi : 0xFFFFu;
                    cq[r] = c0 | (c1 << 16);
                }
            }
        }
    };

    uint32_t t = bidx;
    {   // prologue: the first PRE tiles of this workgroup
        bool all = true;
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const uint32_t ti = t + (uint32_t)i * blocks_per_group;
            if (ti < ntiles) { if (!LATE_DMA || wave >= 4) dma_tile(ti, (uint32_t)(i * BUF_B)); } else all = false;
        }
        if (!LATE_DMA || wave >= 4) dma_wait(all);
        __builtin_amdgcn_s_barrier();                         // also publishes sim_s
        asm volatile("" ::: "memory");
    }
    const bool late = wave >= 4 && (PING || !(a.debug & 16u));   // debug bit 4 (timing only, MODE 0): every wave in the early order
    // Pace gate (advisory; see batch_gemm_wide_kernel): with G > 1 query groups the G workgroups of a bidx are kept within a few
    // tiles of each other so that a tile fetched for one is still in the XCD's L2 when the others want it. Wave 0 runs it in its
    // side phase (PING), in the shadow of the late half's MFMAs.
    constexpr uint32_t GATE_EVERY = 8u, GATE_WINDOW = 6u;
    const uint32_t ngroups = gridDim.x / blocks_per_group;
    const bool gate = !SAMPLE && a.progress != nullptr && ngroups > 1u && (blocks_per_group & 7u) == 0u && ngroups * blocks_per_group <= 256u && !(a.debug & 4096u);
    const uint32_t* gate_word = gate ? a.progress + (bidx & 7u) * 32u + (bidx >> 3) : a.progress;   // one word per bidx, one 128-byte line per XCD
    auto pace = [&](uint32_t it) {
        if (gate && wave == 0 && (it & (GATE_EVERY - 1u)) == 0u && it > 0u) {
            unsigned int add = GATE_EVERY;
            for (uint32_t spins = 0; spins < 1024u; ++spins) {
                unsigned int total = 0u;
                if (lane == 0) total = __hip_atomic_fetch_add(const_cast<uint32_t*>(gate_word), add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
                total = (unsigned int)__builtin_amdgcn_readfirstlane((int)total);
                add = 0u;
                if ((int)(ngroups * it - total) <= (int)(ngroups * GATE_WINDOW)) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
    };
    // SPLIT: bounded spin on the arrival counter (a wave that gives up poisons the workgroup's survivor counts, which sends its
    // queries to the exact path — never a silent wrong answer). The counter is touched through inline assembly only.
    bool gave_up = SPLIT && (a.debug & 16384u) != 0 && blockIdx.x == 1 && wave == 3;   // debug bit 14: pretend one wave timed out (tests)
    const unsigned sync_addr = (unsigned)(size_t)(lds_u32*)sync_s;
    auto wait_arrivals = [&](unsigned int target) {
        bool ok = false;
        for (unsigned int spins = 0; spins < (1u << 22); ++spins) {
            unsigned int v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(sync_addr) : "memory");
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)v) >= target) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) gave_up = true;
    };
    auto arrive = [&]() {
        if (lane == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_u32 %0, %1" ::"v"(sync_addr), "v"(1u) : "memory");
    };
    uint32_t it = 0, cur_idx = 0, t_prev = 0;
    if (EARLY_HEAD && !late && t < ntiles) {                  // tile t is published by the prologue's barrier
        mfma_head(smem_lds + lane_boff);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (; t < ntiles; ++it) {
        const uint32_t baddr = smem_lds + cur_idx * (uint32_t)BUF_B + lane_boff;
        const uint32_t tn = t + PRE * blocks_per_group;
        uint32_t pre_idx = cur_idx + PRE;
        pre_idx = pre_idx >= (uint32_t)NBUF ? pre_idx - NBUF : pre_idx;
        const bool issued = tn < ntiles;
        if (LATE_DMA) {
            // period t. Phase A: the early half multiplies tile t; the late half selects tile t - 1, requests tile t + PRE, waits for
            // tile t + 1 (so that the MID barrier publishes it) and requests the head of its own K loop. Phase B: the late half
            // multiplies tile t; the early half selects tile t and requests the head of K loop t + 1 ahead of the END barrier.
            if (!late) {
                if (!EARLY_HEAD) mfma_head(baddr);
                mfma_body(baddr);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                pace(it);
                select_tile(t);
                if (EARLY_HEAD && t + blocks_per_group < ntiles) {
                    uint32_t nxt_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
                    mfma_head(smem_lds + nxt_idx * (uint32_t)BUF_B + lane_boff);
                    // landed before anything the compiler may do with these registers around the loop edge (the wave would sit at
                    // the barrier meanwhile anyway)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else {
                if (it > 0) select_tile(t_prev);
                if (issued) dma_tile(tn, pre_idx * BUF_B);
                dma_wait(issued);
                mfma_head(baddr);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                mfma_body(baddr);
            }
        } else if (PING) {
            if (!late) {
                mfma_tile(baddr);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                pace(it);                                     // (its returning atomic drains this wave's DMA queue: tile t + 1, long landed)
                if (issued) dma_tile(tn, pre_idx * BUF_B);
                dma_wait(issued);
                select_tile(t);
            } else {
                if (it > 0) select_tile(t_prev);
                if (issued) dma_tile(tn, pre_idx * BUF_B);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                mfma_tile(baddr);
                dma_wait(issued);
            }
        } else if (SPLIT) {
            if (late && it > 0) select_tile(t_prev);
            // every wave is through K loop it - 1 (the buffer tile tn goes to is free) and has its pieces of tile `it` in LDS
            if (it > 0) wait_arrivals(8u * it);
            pace(it);
            if (issued) dma_tile(tn, pre_idx * BUF_B);
            mfma_tile(baddr);
            dma_wait(issued);
            arrive();
            if (!late) select_tile(t);
            t_prev = t;
            cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
            t += blocks_per_group;
            continue;
        } else if (EARLY_DMA_AFTER) {
            // Behind the barrier the early half goes straight into its K loop (the matrix pipe is busy at once) while the late half
            // selects tile t - 1 and requests its pieces of tile t + PRE in the shadow of those MFMAs; the early half requests its
            // pieces, waits and selects behind its K loop, in the shadow of the late half's.
            if (late) {
                if (it > 0) select_tile(t_prev);
                if (issued) dma_tile(tn, pre_idx * BUF_B);
                mfma_tile(baddr);
                dma_wait(issued);
            } else {
                mfma_tile(baddr);
                pace(it);
                if (issued) dma_tile(tn, pre_idx * BUF_B);
                dma_wait(issued);
                select_tile(t);
            }
        } else {
            pace(it);
            if (issued) dma_tile(tn, pre_idx * BUF_B);
            if (late) {
                if (it > 0) select_tile(t_prev);
                mfma_tile(baddr);
                dma_wait(issued);
            } else {
                mfma_tile(baddr);
                dma_wait(issued);
                select_tile(t);
            }
        }
        t_prev = t;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
        t += blocks_per_group;
    }
    if (late && it > 0) select_tile(t_prev);
    if (gate && tid == 0) __hip_atomic_fetch_add(const_cast<uint32_t*>(gate_word), 1u << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // done: nobody waits for this group
    if (SPLIT && gave_up && lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(sync_addr + 4u), "v"(1u) : "memory");
    if (!SAMPLE) {   // the wave's 32 survivor counters, SGPRs -> LDS: lane q takes query q's
        unsigned mine = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mine = lane == (r & 3) + 8 * (r >> 2) ? (cq[r] & 0xFFFFu) : mine;
            mine = lane == (r & 3) + 8 * (r >> 2) + 4 ? (cq[r] >> 16) : mine;
        }
        if (lane < 32) cnt_s[wave * 32 + lane] = mine == 0xFFFFu ? 0x40000000u : mine;   // saturated = unknown = overflowed
    }
    __syncthreads();
    if (!SAMPLE && tid < 256) {
        const bool poisoned = SPLIT && sync_s[1] != 0u;       // a wave gave up waiting: nothing this workgroup selected can be trusted
        a.seg_count[(size_t)bidx * (a.nqt * 128u) + group * 256u + (uint32_t)tid] = poisoned ? 0x40000000u : cnt_s[tid];
    }
}

// ---------------------------------------------------------------------------
// Register-resident-queries GEMM, ONE wave per SIMD ("w4": batch_rega = 3; D in {128, 256, 384, 512}).
//
// batch_gemm_rega_kernel keeps 32 queries per wave and two waves per SIMD: every B fragment read from LDS feeds one
// MFMA, the two waves of a SIMD split the matrix pipe, only ONE 48 KB tile per CU is in flight from HBM (which caps
// the stream near 6 TB/s: bytes in flight / latency), and its counters say the pipe is busy 48 % of the time with
// the waves parked on LDS / barriers for most of the rest (profiles/r01/ay_*). Here a workgroup is 4 waves — one per
// SIMD, 512 registers each — and a wave keeps 64 queries (two A-fragment sets, 192 VGPRs at D = 384):
//   * every B fragment feeds TWO MFMAs (both query sets): half the LDS reads per flop;
//   * a k-step is 1 ds_read_b128 + 2 MFMAs on two independent accumulators, ~2 other instructions per MFMA issue
//     slot, far below the ~5 a wave can hide behind a 32-cycle MFMA;
//   * tiles are 32 rows (24.5 KB at D = 384) in a ring of NBUF LDS buffers filled by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write pass), requested PRE = NBUF - 2 tiles ahead
//     (D = 384: 6 buffers, 4 tiles = 98 KB per CU in flight) with counted vmcnt waits and a raw s_barrier per tile,
//     so the requests stay in flight across barriers;
//   * the threshold test costs no compare against a per-query value: the accumulators start at -(sim_lo_q) instead
//     of 0 (the first MFMA of a tile takes its C operand from 32 loop-invariant registers), so "passes its query's
//     threshold" is the SIGN of the accumulator. The selection of tile t-1 (second accumulator set) is spread over
//     the k-steps of tile t: one quad of accumulators at a time — v_max3, v_max, v_cmp, branch — in the shadow of
//     the MFMAs; only a quad with a hit (~1 per tile at ~500 survivors per query) leaves the stream for ~40
//     instructions (exact re-test d = 1 - (acc + sim_lo) <= tau, slot from an LDS counter, 8-byte store).
// The padded tile image (row stride 2D + 16 bytes) is cut into 1-KB DMA pieces; lane l of piece P fetches whatever
// belongs at slot 64 P + l (a row's pad slot re-fetches its last segment); the last piece may run past the 32 rows
// into the rows that follow — the mirror is allocated with slack rows for that, and rows past the slab are masked by
// the selection. Survivor segments, thresholds and the exact test are those of batch_gemm_rega_kernel, so the host
// side and batch_finish_kernel do not know which of the two ran. A query whose threshold is not finite (no usable
// sample) is marked as overflowed (count 2^30) and answered by the exact path.
template <int D, int AH = 3>
struct W4Geom {
    static constexpr int KS = D / 16;                                   // MFMA k-steps
    static constexpr int ROW_B = D * 2 + 16;                            // LDS row stride (bytes): conflict-free ds_read_b128
    static constexpr int TROWS = 32;
    static constexpr int PIECES = (TROWS * ROW_B + 1023) / 1024;        // 1-KB DMA pieces per tile
    static constexpr int BUF_B = PIECES * 1024;
    static constexpr int NBUF_RAW = (160 * 1024 - 3 * 256 * 4) / BUF_B;
    static constexpr int NBUF = NBUF_RAW > 8 ? 8 : NBUF_RAW;
    static constexpr int PRE = NBUF - 2;                                // tiles requested ahead of the one being read
    static constexpr size_t SMEM = (size_t)NBUF * BUF_B + 256 * 4;
    static constexpr int AHEAD = AH, RING = AHEAD + 1;                  // B-fragment read-ahead (k-steps)
    static constexpr int UNITS = 8;                                     // selection units per tile: 2 accumulators x 4 quads
    static_assert(NBUF >= 4, "need at least two tiles in flight");
};

// Per-wave state of batch_gemm_w4_kernel shared by its (compile-time unrolled) helper functions. Everything is
// loop-invariant; after inlining it lives in registers.
template <int D, int AH>
struct W4Ctx {
    bf16x8 fa0[W4Geom<D, AH>::KS], fa1[W4Geom<D, AH>::KS];     // A fragments of the wave's two query sets
    int64_t* cand;
    uint32_t cand_cap, row_base, slab0, slab_end, seg_slots, seg_w0;
    f32x16 nl0, nl1;             // accumulator start values: -(conservative similarity bound) of the register's query, per set
    uint32_t debug;              // timing experiments (GemmArgs::debug)
    int cnt_v;                   // lane l: survivors of the wave's query l so far (this workgroup's segment fill)
    int lane;
};

// One quad of accumulators of a finished tile: a clear sign bit => some (query, row) reached its threshold.
// Hot part: 3 integer ANDs, one compare, one branch. Cold part (one copy per unit; ~1 visit per wave and tile): with one
// wave per SIMD nothing hides a memory round trip, so it makes none: admission is the sign test itself (the conservative
// bound: a superset of "d <= tau"; every row it rejects has d > tau, which is what the certificate needs), the start
// value comes from the registers, and the slot in the query's segment comes from a per-wave counter kept in a VGPR
// (lane l = the wave's query l; only this wave ever appends to its 64 queries' segments of this workgroup), advanced
// with readlane + lane-compare adds in a scalar loop over the hit lanes. The only memory operation is the 8-byte store.
template <int D, int AH, int U>
__device__ __forceinline__ void w4_select_unit(W4Ctx<D, AH>& c, const f32x16 (&P)[2], uint32_t tile) {
    constexpr int set = U >> 2, qd = U & 3;
    const float v0 = P[set][4 * qd], v1 = P[set][4 * qd + 1], v2 = P[set][4 * qd + 2], v3 = P[set][4 * qd + 3];
    const int all_neg = __float_as_int(v0) & __float_as_int(v1) & __float_as_int(v2) & __float_as_int(v3);
    if (__ballot(all_neg >= 0) == 0ull) return;           // every sign bit set: nothing reached its threshold
    if (c.debug & 16u) return;                            // timing experiments: hot test only
    const uint32_t row = c.slab0 + tile * (uint32_t)W4Geom<D, AH>::TROWS + (uint32_t)(c.lane & 31);
    unsigned hits = (v0 >= 0.0f ? 1u : 0u) | (v1 >= 0.0f ? 2u : 0u) | (v2 >= 0.0f ? 4u : 0u) | (v3 >= 0.0f ? 8u : 0u);   // NaN fails
    hits = row < c.slab_end ? hits : 0u;                  // rows past the slab (clamped / slack rows of the last tile)
    const f32x16& nl = set ? c.nl1 : c.nl0;
    const float n0 = nl[4 * qd], n1 = nl[4 * qd + 1], n2 = nl[4 * qd + 2], n3 = nl[4 * qd + 3];
    // register 4 qd + j of set s belongs to the wave's query 32 s + j + 8 qd + 4 (lane >> 5)
    const uint32_t qq0 = 32u * (uint32_t)set + 8u * (uint32_t)qd + 4u * ((uint32_t)c.lane >> 5);
    for (;;) {
        const bool has = hits != 0u;
        unsigned long long todo = __ballot(has);
        if (todo == 0ull) break;
        const unsigned j = has ? (unsigned)__builtin_ctz(hits) : 0u;   // the lane's lowest pending register
        hits &= hits - 1u;
        const float v = j == 0u ? v0 : (j == 1u ? v1 : (j == 2u ? v2 : v3));
        const float ng = j == 0u ? n0 : (j == 1u ? n1 : (j == 2u ? n2 : n3));
        const int qq = (int)(qq0 + j);
        const float d = (1.0f - (v - ng)) + 0.0f;           // v = sim - sim_lo, ng = -sim_lo
        int slot = 0;
        do {                                                // scalar loop over the hit lanes (usually one)
            const int L = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int qL = __builtin_amdgcn_readlane(qq, L);
            const int cL = __builtin_amdgcn_readlane(c.cnt_v, qL);
            c.cnt_v += (c.lane == qL) ? 1 : 0;              // (no v_writelane builtin in this toolchain: compare + add)
            slot = (c.lane == L) ? cL : slot;
        } while (todo != 0ull);
        if (has && (uint32_t)slot < c.seg_slots && !(c.debug & 32u))   // debug bit5: no survivor store
            c.cand[c.seg_w0 + (uint32_t)qq * c.cand_cap + (uint32_t)slot] = make_key(d, c.row_base + row);
    }
}

template <int D, int AH, int U0, int U1>
__device__ __forceinline__ void w4_select_units(W4Ctx<D, AH>& c, const f32x16 (&P)[2], uint32_t tile) {
    if constexpr (U0 < U1) {
        w4_select_unit<D, AH, U0>(c, P, tile);
        w4_select_units<D, AH, U0 + 1, U1>(c, P, tile);
    }
}

// k-steps KSI .. KS-1 of one tile: read-ahead of B fragment KSI + AHEAD (inline asm), counted wait for fragment KSI,
// two MFMAs, then the selection units of the previous tile that are scheduled after this k-step.
// The B-fragment reads are inline asm with hand-counted waits: with branches inside the k-loop hipcc falls back to
// `s_waitcnt lgkmcnt(0)` in front of every MFMA pair (one exposed LDS round trip per k-step). LDS operations return
// in order, so "at most N younger operations outstanding" implies that read KSI has landed whatever else (the cold
// path's reads, scalar loads) is in the queue: extra operations only make a counted wait stricter.
template <int D, int AH, int KSI>
__device__ __forceinline__ void w4_ksteps(W4Ctx<D, AH>& c, f32x16 (&cur)[2], const f32x16 (&prev)[2], uint32_t prev_tile,
                                          uint32_t baddr, u32x4 (&fb)[W4Geom<D, AH>::RING]) {
    using G = W4Geom<D, AH>;
    if constexpr (KSI < G::KS) {
        if constexpr (KSI + G::AHEAD < G::KS)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[(KSI + G::AHEAD) % G::RING]) : "v"(baddr), "n"((KSI + G::AHEAD) * 32));
        constexpr int younger = (G::KS - 1 - KSI) < G::AHEAD ? (G::KS - 1 - KSI) : G::AHEAD;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 B = __builtin_bit_cast(bf16x8, fb[KSI % G::RING]);
        if constexpr (KSI == 0) {   // C operand = the per-query start values: no accumulator initialisation pass
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.fa0[0], B, c.nl0, 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.fa1[0], B, c.nl1, 0, 0, 0);
        } else {
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.fa0[KSI], B, cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.fa1[KSI], B, cur[1], 0, 0, 0);
        }
        // selection units spread evenly over the k-steps: unit u runs after k-step floor(u * KS / UNITS)
        constexpr int u0 = (KSI * G::UNITS + G::KS - 1) / G::KS;            // first u with floor(u KS / UNITS) >= KSI
        constexpr int u1 = ((KSI + 1) * G::UNITS + G::KS - 1) / G::KS;      // first u with floor(u KS / UNITS) >= KSI + 1
        if (!(c.debug & 8u)) w4_select_units<D, AH, u0, (u1 < G::UNITS ? u1 : G::UNITS)>(c, prev, prev_tile);   // debug bit3: no selection
        __builtin_amdgcn_sched_barrier(0);
        w4_ksteps<D, AH, KSI + 1>(c, cur, prev, prev_tile, baddr, fb);
    }
}

template <int D, int AH, int I>
__device__ __forceinline__ void w4_prefetch_b(uint32_t baddr, u32x4 (&fb)[W4Geom<D, AH>::RING]) {
    using G = W4Geom<D, AH>;
    if constexpr (I < G::AHEAD && I < G::KS) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[I]) : "v"(baddr), "n"(I * 32));
        w4_prefetch_b<D, AH, I + 1>(baddr, fb);
    }
}

// MFMAs of one tile into `cur`, with the selection of the PREVIOUS tile (`prev`) interleaved between the k-steps.
template <int D, int AH>
__device__ __forceinline__ void w4_tile_step(W4Ctx<D, AH>& c, f32x16 (&cur)[2], const f32x16 (&prev)[2], uint32_t prev_tile,
                                             uint32_t baddr) {
    u32x4 fb[W4Geom<D, AH>::RING];
    w4_prefetch_b<D, AH, 0>(baddr, fb);
    __builtin_amdgcn_sched_barrier(0);
    w4_ksteps<D, AH, 0>(c, cur, prev, prev_tile, baddr, fb);
}

template <int D, int AH>
__global__ __launch_bounds__(256, 1) void batch_gemm_w4_kernel(GemmArgs a, uint32_t blocks_per_group) {
    using G = W4Geom<D, AH>;
    constexpr int KS = G::KS;
    constexpr int ROW_B = G::ROW_B, TROWS = G::TROWS, PIECES = G::PIECES, BUF_B = G::BUF_B, NBUF = G::NBUF, PRE = G::PRE;
    constexpr int SPR = ROW_B / 16;                  // 16-byte slots per padded row
    constexpr int PPW = (PIECES + 3) / 4;            // pieces per wave (waves with index >= PIECES % 4 carry one less, if PIECES % 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* neg_s = reinterpret_cast<float*>(smem + NBUF * BUF_B);          // [256] -(conservative similarity bound) (set-up only)

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t group = blockIdx.x / blocks_per_group;    // 256 queries per group
    const uint32_t bidx = blockIdx.x % blocks_per_group;
    const uint32_t q0 = group * 256 + wave * 64;              // this wave's 64 queries

    W4Ctx<D, AH> c;
    // A fragments: set s, lane l holds query 32 s + (l & 31), k = 16 ks + 8 (l >> 5) .. +7
    {
        const u32x4* qp0 = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + (lane & 31)) * D) + (lane >> 5);
        const u32x4* qp1 = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + 32 + (lane & 31)) * D) + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            c.fa0[ks] = __builtin_bit_cast(bf16x8, qp0[ks * 2]);
            c.fa1[ks] = __builtin_bit_cast(bf16x8, qp1[ks * 2]);
        }
    }
    {
        const float tq = a.tau[q0 + lane];
        const bool usable = (tq == tq) && (tq < __builtin_inff());        // -inf (padding query) is usable: it admits nothing
        c.cnt_v = usable ? 0 : 0x40000000;                                 // lane l counts the wave's query l
        // fl(1 - sim) <= tq implies sim >= (1 - tq) - 2^-23 (|1 - tq| + |tq|); 4e-7 (1 + |tq|) covers it and the rounding of
        // (acc + sim_lo) with slack. tq = -inf gives NaN: nothing is flagged.
        const float sim_lo = (1.0f - tq) - 4e-7f * (1.0f + __builtin_fabsf(tq));
        neg_s[wave * 64 + lane] = usable ? -sim_lo : __builtin_nanf("");
    }
    __syncthreads();     // the only compiler-visible LDS writes of the kernel: all before the first DMA request
    c.cand = a.cand; c.cand_cap = a.cand_cap; c.row_base = a.row_base; c.slab0 = a.slab0;
    c.slab_end = a.slab0 + a.slab_rows;
    c.seg_slots = a.seg_area / blocks_per_group;
    c.seg_w0 = q0 * a.cand_cap + a.seg_base + bidx * c.seg_slots;   // element offset of the wave's first query row, this workgroup's segment
    c.lane = lane;
    c.debug = a.debug;
    // the lane's 32 accumulator start values: register r of set s belongs to query 32 s + (r&3) + 8 (r>>2) + 4 (lane>>5)
    {
        const float* nw = neg_s + wave * 64 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            c.nl0[r] = nw[(r & 3) + 8 * (r >> 2)];
            c.nl1[r] = nw[32 + (r & 3) + 8 * (r >> 2)];
        }
    }

    const uint32_t ntiles = (a.slab_rows + TROWS - 1) / TROWS;
    const unsigned char* cbase = reinterpret_cast<const unsigned char*>(a.cb) + (size_t)a.slab0 * (D * 2);

    // LDS-DMA map: wave w moves pieces w, w + 4, ...; lane l of piece P fills slot 64 P + l of the padded image.
    // Source offset of that slot inside the tile's contiguous rows (loop-invariant: one register per piece).
    uint32_t poff[PPW];
    int my_pieces = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const uint32_t P = (uint32_t)wave + 4u * i;
        const uint32_t slot = P * 64u + (uint32_t)lane;
        const uint32_t r = slot / SPR;
        uint32_t cc = slot - r * SPR;
        cc = cc < (uint32_t)(SPR - 1) ? cc : (uint32_t)(SPR - 2);         // pad slot: re-fetch the row's last segment
        poff[i] = r * (uint32_t)(D * 2) + cc * 16u;                       // r may reach a few rows past the tile (last piece): slack rows
        if (P < (uint32_t)PIECES) ++my_pieces;
    }
    const bool full_wave = my_pieces == PPW;                  // wave-uniform
    auto dma_tile = [&](uint32_t tile, uint32_t buf_idx) {
        const unsigned char* src0 = cbase + (size_t)tile * (TROWS * D * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const uint32_t P = (uint32_t)wave + 4u * i;
            if (i < PPW - 1 || full_wave)
                __builtin_amdgcn_global_load_lds((global_cvoid*)(src0 + poff[i]), (lds_void*)(smem + buf_idx * BUF_B + P * 1024u), 16, 0, 0);
        }
    };
    // wait until this wave's DMA requests of all but the newest PRE - 1 tiles have landed (steady state), or all of them
    auto dma_wait_keep = [&](bool steady) {
        if (!steady) wait_vmcnt<0>();
        else if (full_wave) wait_vmcnt<(PRE - 1) * PPW>();
        else wait_vmcnt<(PRE - 1) * (PPW - 1)>();
    };

    const uint32_t lane_boff = (uint32_t)(lane & 31) * (uint32_t)ROW_B + (uint32_t)(lane >> 5) * 16u;
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_void*)smem;
    // timing experiments (results are garbage): bit0 no DMA after the prologue, bit1 no MFMA tile work, bit2 no per-tile
    // barrier, bit3 no selection
    const bool dbg_nodma = (a.debug & 1u) != 0, dbg_nomfma = (a.debug & 2u) != 0, dbg_nobar = (a.debug & 4u) != 0;

    f32x16 accA[2], accB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[i][r] = -1.0f; accB[i][r] = -1.0f; }   // "no hit": nothing to select before the first tile

    // The A fragments are ordinary loads: make them land BEFORE the first DMA request (beside a DMA in flight hipcc waits
    // vmcnt(0) for an ordinary load's first use, which would drain the prologue's prefetch). Loads return in order.
    asm volatile("" ::"v"(c.fa0[KS - 1]), "v"(c.fa1[KS - 1]));
    // prologue: request the first PRE tiles, wait for the first
    uint32_t t = bidx;
    {
        uint32_t tt = t;
        int issued = 0;
#pragma unroll
        for (int i = 0; i < PRE; ++i, tt += blocks_per_group)
            if (tt < ntiles) { dma_tile(tt, (uint32_t)i); ++issued; }
        dma_wait_keep(issued == PRE);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    uint32_t cur_idx = 0;                                      // ring position of tile t
    while (t < ntiles) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const uint32_t tp = t + (uint32_t)PRE * blocks_per_group;
            uint32_t pre_idx = cur_idx + (uint32_t)PRE;
            pre_idx = pre_idx >= (uint32_t)NBUF ? pre_idx - (uint32_t)NBUF : pre_idx;
            const bool issued = tp < ntiles && !dbg_nodma;
            if (issued) dma_tile(tp, pre_idx);
            const uint32_t baddr = smem_lds + cur_idx * (uint32_t)BUF_B + lane_boff;
            if (!dbg_nomfma) {
                if (par == 0) w4_tile_step<D, AH>(c, accA, accB, t - blocks_per_group, baddr);   // first iteration: accB is all -1, its tile index is never used
                else w4_tile_step<D, AH>(c, accB, accA, t - blocks_per_group, baddr);
            }
            // tile t + 1 must have landed (every wave waits for its own pieces, the barrier joins them); the younger
            // requests stay in flight across the barrier
            dma_wait_keep(issued);
            if (!dbg_nobar) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
            const uint32_t tn = t + blocks_per_group;
            if (tn >= ntiles) {
                if (par == 0) w4_select_units<D, AH, 0, G::UNITS>(c, accA, t);
                else w4_select_units<D, AH, 0, G::UNITS>(c, accB, t);
                t = tn;
                break;
            }
            t = tn;
        }
    }
    // unclamped counts: a count above seg_slots tells the consumer that survivors were dropped (query -> exact path)
    a.seg_count[(size_t)bidx * (a.nqt * 128u) + q0 + (uint32_t)lane] = (uint32_t)c.cnt_v;
}

static void rega_geometry(const GemmArgs& a, uint32_t* groups, uint32_t* per_group);
static bool w4_dims(uint32_t dims) { return dims == 128 || dims == 256 || dims == 384 || dims == 512; }
uint32_t batch_w4_slack_rows() { return 64; }   // rows the w4 kernel's last DMA piece may read past the end of the store

template <int D, int AH>
static hipError_t launch_w4_ah(const GemmArgs& a, hipStream_t st) {
    constexpr size_t smem = W4Geom<D, AH>::SMEM;
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_w4_kernel<D, AH>), smem, configured);
        if (e != hipSuccess) return e;
    }
    uint32_t groups, pg;
    rega_geometry(a, &groups, &pg);          // the same workgroups-per-group (= survivor segments) as batch_gemm_rega_kernel
    hipLaunchKernelGGL((batch_gemm_w4_kernel<D, AH>), dim3(groups * pg), dim3(256), smem, st, a, pg);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_w4(const GemmArgs& a, hipStream_t st) {
    // read-ahead 3 and 6 k-steps measure the same (profiles/r02/i_w4_ahead.txt): the kernel is issue-bound, not LDS-latency-bound
    return launch_w4_ah<D, 3>(a, st);
}

// D = 1024 would need 2 x 66 KB of tiles + 32 KB of partial sums (> 160 KB of LDS): it stays on the LDS-tiled kernel.
static bool ksplit_dims(uint32_t dims) { return dims == 768; }
// D = 768 has two register-resident kernels. "batch_rega" 1 / 6 / 7 select the K-split kernel (1 = workgroup barrier, 6 = split
// barrier at every size, 7 = its own size rule); every other value the wide kernel (whole K per wave, LDS-DMA staging).
static bool ksplit_mode(uint32_t use_rega) { return use_rega == 1u || use_rega == 6u || use_rega == 7u; }
// Queries per workgroup group of the register-resident filtering GEMM (the planner sizes survivor segments by it).
uint32_t batch_group_queries(uint32_t dims, uint32_t use_rega) { return (ksplit_dims(dims) && ksplit_mode(use_rega)) ? 128u : 256u; }

static void rega_geometry(const GemmArgs& a, uint32_t* groups, uint32_t* per_group) {
    const bool ksplit = batch_group_queries(a.dims, a.use_rega) == 128u;   // K-split: 128 queries per workgroup
    *groups = ksplit ? a.nqt : (a.nqt * 128 + 255) / 256;
    const uint32_t ntiles = ksplit_dims(a.dims) ? (a.slab_rows + 31) / 32 : (a.slab_rows + 63) / 64;   // 768: 32-row tiles (both kernels)
    uint32_t pg = 256 / *groups;                // one persistent workgroup per CU in total
    if (pg < 1) pg = 1;
    if (pg > ntiles) pg = ntiles;
    *per_group = pg;
}

static bool rega_eligible(const GemmArgs& a, int metric) {
    return a.dense == nullptr && a.use_rega && metric != BM_L2 &&
           (a.dims == 128 || a.dims == 256 || a.dims == 384 || a.dims == 512 || ksplit_dims(a.dims));
}

bool batch_gemm_segments(const GemmArgs& a, int metric, uint32_t* nseg, uint32_t* seg_slots) {
    if (!rega_eligible(a, metric)) return false;
    uint32_t groups, per_group;
    rega_geometry(a, &groups, &per_group);
    *nseg = per_group;
    *seg_slots = a.seg_area / per_group;
    return true;
}

template <int D, int AHEAD, bool SPLIT = true>
static hipError_t launch_ksplit(const GemmArgs& a, hipStream_t st) {
    constexpr size_t smem = 2 * 32 * (D * 2 + 16) + 2 * 4 * (4 * 64 * 16) + 3 * 4 * 32 * 4 + 16 + 32;  // tiles, partial sums, thresholds / counters / bounds, split-barrier counters
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_ksplit_kernel<D, AHEAD, false, SPLIT>), smem, configured);
        if (e != hipSuccess) return e;
    }
    uint32_t groups, per_group;
    rega_geometry(a, &groups, &per_group);
    hipLaunchKernelGGL((batch_gemm_ksplit_kernel<D, AHEAD, false, SPLIT>), dim3(groups * per_group), dim3(512), smem, st, a, per_group);
    return hipGetLastError();
}

template <int D, int NBUF, int AHEAD, int CHAINS = 1, bool SPLIT = false>
static hipError_t launch_wide(const GemmArgs& a, hipStream_t st) {
    constexpr size_t smem = (size_t)NBUF * (((32 * (D * 2 + 16)) + 1023) / 1024 * 1024) + 3 * 8 * 32 * 4 + 64;   // tile buffers, thresholds / counters / bounds
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_wide_kernel<D, NBUF, AHEAD, false, CHAINS, SPLIT>), smem, configured);
        if (e != hipSuccess) return e;
    }
    uint32_t groups, per_group;
    rega_geometry(a, &groups, &per_group);
    hipLaunchKernelGGL((batch_gemm_wide_kernel<D, NBUF, AHEAD, false, CHAINS, SPLIT>), dim3(groups * per_group), dim3(512), smem, st, a, per_group);
    return hipGetLastError();
}

template <int D, int TROWS, int NBUF, int AHEAD, int MODE>
static hipError_t launch_pp(const GemmArgs& a, hipStream_t st) {
    constexpr size_t smem = (size_t)NBUF * (((TROWS * (D * 2 + 16)) + 1023) / 1024 * 1024) + 2 * 8 * 32 * 4 + 64;   // tile buffers, counters, bounds
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_pp_kernel<D, TROWS, NBUF, AHEAD, false, MODE>), smem, configured);
        if (e != hipSuccess) return e;
    }
    uint32_t groups, per_group;
    rega_geometry(a, &groups, &per_group);
    hipLaunchKernelGGL((batch_gemm_pp_kernel<D, TROWS, NBUF, AHEAD, false, MODE>), dim3(groups * per_group), dim3(512), smem, st, a, per_group);
    return hipGetLastError();
}

template <int D, bool GLDS, int AHEAD, bool PROF = false, int SYNC = 0>
static hipError_t launch_rega_impl(const GemmArgs& a, hipStream_t st) {
    constexpr bool FREE = SYNC == 1;
    // tiles, thresholds, survivor counters, bounds, claimed tile indices (+ the two counter triples of the free-running variant)
    constexpr size_t smem = (size_t)(FREE ? 3 : rega_lds_tiles<D>(GLDS)) * 64 * (D * 2 + 16) + 3 * 8 * 32 * 4 + 16 + (SYNC != 0 ? 48 : 0);   // FREE / SPLIT: stored[3], read[3], turn[4] behind the claimed tile indices
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_rega_kernel<D, GLDS, AHEAD, false, PROF, SYNC>), smem, configured);
        if (e != hipSuccess) return e;
    }
    uint32_t groups, per_group;
    rega_geometry(a, &groups, &per_group);
    hipLaunchKernelGGL((batch_gemm_rega_kernel<D, GLDS, AHEAD, false, PROF, SYNC>), dim3(groups * per_group), dim3(512), smem, st, a, per_group);
    return hipGetLastError();
}

// whether the free-running variant (three LDS tiles, no tile barrier) exists for D
constexpr bool rega_free_dims(int d) { return 3 * 64 * (d * 2 + 16) + 3 * 8 * 32 * 4 + 64 <= 160 * 1024; }

template <int D>
static hipError_t launch_rega(const GemmArgs& a, hipStream_t st) {
    // B-fragment read-ahead: as deep as the 256-VGPR budget allows next to the D/4 VGPRs of A fragments
    constexpr int AHEAD = D >= 512 ? 1 : 3;
    if (a.use_rega == 2u) {
        if constexpr (D < 512) {
            switch ((a.debug >> 8) & 3u) {   // timing experiments: read-ahead depth
                case 1: return launch_rega_impl<D, true, 2>(a, st);
                case 2: return launch_rega_impl<D, true, 4>(a, st);
                default: break;
            }
        }
        return launch_rega_impl<D, true, AHEAD>(a, st);
    }
    if constexpr (rega_free_dims(D)) {
        if (a.use_rega == 4u) {
            if constexpr (D == 384) {
                if (a.debug & 1024u) return launch_rega_impl<D, false, AHEAD, true, 1>(a, st);   // phase clock, see below
            }
            return launch_rega_impl<D, false, AHEAD, false, 1>(a, st);
        }
    }
    if (a.use_rega == 5u || a.use_rega == 6u) {   // 6: like 5, and the K-split kernel keeps the split barrier at every size
        if constexpr (D == 384) {
            if (a.debug & 1024u) return launch_rega_impl<D, false, AHEAD, true, 2>(a, st);
        }
        return launch_rega_impl<D, false, AHEAD, false, 2>(a, st);
    }
    if constexpr (D == 384) {
        // diagnosis run: per-wave phase clock (see the kernel's main loop)
        if (a.debug & 1024u) return launch_rega_impl<D, false, AHEAD, true>(a, st);
        // timing experiment (debug bits 8-9 = 1): read-ahead 2 = 224 VGPRs, which leaves room for a 64-VGPR kernel of the
        // neighbouring batch (the finish kernel) beside the two GEMM waves of a SIMD; read-ahead 3 = 232 does not
        if (((a.debug >> 8) & 3u) == 1u) return launch_rega_impl<D, false, 2>(a, st);
    }
    return launch_rega_impl<D, false, AHEAD>(a, st);
}

hipError_t launch_batch_gemm(const GemmArgs& a, int metric, hipStream_t st) {
    // fast path: queries resident in registers (needs the query block padded to a multiple of 256 rows)
    if (rega_eligible(a, metric)) {
        if (a.use_rega == 3u && w4_dims(a.dims)) {   // one wave per SIMD, 64 queries per wave (same segments / geometry)
            switch (a.dims) {
                case 128: return launch_w4<128>(a, st);
                case 256: return launch_w4<256>(a, st);
                case 512: return launch_w4<512>(a, st);
                default: return launch_w4<384>(a, st);
            }
        }
        if (a.use_rega >= 8u && a.use_rega <= 12u) {   // round 5: 8 = ping-pong, late-half DMA, prefetched heads; 9 = asm-read fix alone; 10 = plain ping-pong
            const uint32_t v = (a.debug >> 8) & 3u;   // timing experiments: read-ahead depth
            if (a.dims == 768) {
                if (a.use_rega == 8u) return v == 1u ? launch_pp<768, 32, 3, 4, 3>(a, st) : v == 2u ? launch_pp<768, 32, 3, 2, 3>(a, st) : launch_pp<768, 32, 3, 3, 3>(a, st);
                if (a.use_rega == 10u) return launch_pp<768, 32, 3, 3, 1>(a, st);
                if (a.use_rega == 11u) return launch_pp<768, 32, 3, 3, 4>(a, st);
                if (a.use_rega == 12u) return launch_pp<768, 32, 3, 3, 5>(a, st);
                return v == 2u ? launch_pp<768, 32, 3, 2, 0>(a, st) : launch_pp<768, 32, 3, 3, 0>(a, st);
            }
            if (a.dims == 384) {
                if (a.use_rega == 8u) return v == 1u ? launch_pp<384, 64, 3, 6, 2>(a, st) : v == 2u ? launch_pp<384, 64, 3, 3, 3>(a, st) : launch_pp<384, 64, 3, 3, 2>(a, st);
                if (a.use_rega == 10u) return launch_pp<384, 64, 3, 3, 1>(a, st);
                if (a.use_rega == 11u) return launch_pp<384, 64, 3, 3, 4>(a, st);
                if (a.use_rega == 12u) return v == 1u ? launch_pp<384, 64, 3, 6, 5>(a, st) : launch_pp<384, 64, 3, 3, 5>(a, st);
                return launch_pp<384, 64, 3, 3, 0>(a, st);
            }
        }
        switch (a.dims) {
            case 128: return launch_rega<128>(a, st);
            case 256: return launch_rega<256>(a, st);
            case 384: return launch_rega<384>(a, st);
            case 512: return launch_rega<512>(a, st);
            case 768:
                if (!ksplit_mode(a.use_rega)) {
                    switch ((a.debug >> 8) & 3u) {   // timing experiments: LDS tile buffers / accumulator chains / read-ahead
                        case 1: return launch_wide<768, 2, 3>(a, st);             // two LDS tile buffers
                        case 2: return launch_wide<768, 3, 3, 1, true>(a, st);    // split tile barrier
                        case 3: return launch_wide<768, 3, 2>(a, st);             // read-ahead 2
                        default: return launch_wide<768, 3, 3>(a, st);            // three buffers, read-ahead 3, workgroup barrier
                    }
                }
                switch ((a.debug >> 8) & 3u) {   // timing experiments: B-fragment read-ahead depth
                    case 1: return launch_ksplit<768, 6>(a, st);
                    case 2: return launch_ksplit<768, 8>(a, st);
                    default: {
                        // The split barrier wins on short launches (1.25M rows: +1.2 % alone, +2.5 % pipelined) and LOSES 4 % on
                        // the 16 ms launch over 10M rows (profiles/r03/zi_ksplit_split_barrier_by_size.txt) — long enough for the
                        // power limit to set the clock, where the busier pipe buys nothing and the polling costs: workgroup barrier
                        // from 4 096 tiles per workgroup up. ("batch_rega" = 1 forces the workgroup barrier, 6 the split one.)
                        uint32_t groups = 1, per_group = 1;
                        rega_geometry(a, &groups, &per_group);
                        const uint32_t tiles_per_wg = ((a.slab_rows + 31u) / 32u + per_group - 1u) / per_group;
                        if (a.use_rega == 1u || (a.use_rega != 6u && tiles_per_wg > 4096u)) return launch_ksplit<768, 4, false>(a, st);
                        return launch_ksplit<768, 4>(a, st);
                    }
                }
            default: break;
        }
    }
    const uint32_t ctiles = (a.slab_rows + GN - 1) / GN;
    const dim3 grid(ctiles * a.nqt);
    switch (metric) {
        case BM_COS: hipLaunchKernelGGL((batch_gemm_kernel<BM_COS>), grid, dim3(256), 0, st, a); break;
        case BM_DOT: hipLaunchKernelGGL((batch_gemm_kernel<BM_DOT>), grid, dim3(256), 0, st, a); break;
        case BM_L2: hipLaunchKernelGGL((batch_gemm_kernel<BM_L2>), grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Between slabs, per query: keep the best kp of the appended candidates (sorted ascending at the head
// of the list), reset the count, tighten tau. A list that overflowed its capacity marks the query
// (it will be answered by the exact path).
template <int CAP>
__global__ __launch_bounds__(SCAN_THREADS) void tighten_kernel(TightenArgs a) {
    __shared__ int64_t lds[SCAN_WAVES * CAP + SCAN_WAVES + FUSED_MAX_K];
    int* counts = reinterpret_cast<int*>(lds + SCAN_WAVES * CAP);
    int64_t* fin = lds + SCAN_WAVES * CAP + SCAN_WAVES;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t q = blockIdx.x;
    const int kp = a.kp;
    // The kernel is a chain of memory round trips with little work in between (waves wait ~80 % of their cycles), so
    // every load that does not depend on another one is issued up front: the list length, this thread's segment
    // count and a speculative best-list entry go out together; the segment slots follow in one batch of 8.
    bool dropped = false;
    int64_t* __restrict__ mine = a.cand + (size_t)q * a.cand_cap;
    WaveTopK<CAP> tk;
    tk.init(lds + wave * CAP, kp);
    if (a.dense != nullptr) {
        // first slab: a dense tile of <= 2048 approximate distances per query, 8 independent loads per thread
        const float* __restrict__ drow = a.dense + (size_t)q * a.dense_ld;
        constexpr int DLOADS = 8;
        for (uint32_t base = 0; base < a.dense_rows; base += SCAN_THREADS * DLOADS) {
            float v[DLOADS];
#pragma unroll
            for (int i = 0; i < DLOADS; ++i) {
                const uint32_t idx = base + i * SCAN_THREADS + threadIdx.x;
                v[i] = idx < a.dense_rows ? drow[idx] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < DLOADS; ++i) {
                const uint32_t idx = base + i * SCAN_THREADS + threadIdx.x;
                tk.push_wide(make_key(v[i], a.dense_row0 + idx), idx < a.dense_rows);
            }
        }
    } else if (a.nseg != 0u) {
        // register-resident GEMMs: [0, n_best) is the previous best list, then one segment per GEMM workgroup
        const uint32_t n_cnt = a.cand_count[(size_t)q * CAND_COUNT_STRIDE];
        const uint32_t seg0 = threadIdx.x;
        uint32_t c = (seg0 < a.nseg) ? a.seg_count[(size_t)seg0 * a.nq_pad + q] : 0u;
        const int64_t best = (threadIdx.x < a.seg_base) ? mine[threadIdx.x] : KEY_PAD;     // speculative: valid below n_best
        const uint32_t n_best = n_cnt < a.seg_base ? n_cnt : a.seg_base;
        for (uint32_t sb = 0; sb < a.nseg; sb += SCAN_THREADS) {
            const uint32_t seg = sb + threadIdx.x;
            if (sb != 0u) c = (seg < a.nseg) ? a.seg_count[(size_t)seg * a.nq_pad + q] : 0u;
            if (c > a.seg_slots) {
                dropped = true;
                c = a.seg_slots;
            }
            const int64_t* __restrict__ sp = mine + a.seg_base + (size_t)seg * a.seg_slots;
            constexpr uint32_t SLOTS = 8;
            int64_t key[SLOTS];
#pragma unroll
            for (uint32_t u = 0; u < SLOTS; ++u) key[u] = (u < c) ? sp[u] : KEY_PAD;
            if (sb == 0u) tk.push_wide(best, threadIdx.x < n_best);
#pragma unroll
            for (uint32_t u = 0; u < SLOTS; ++u) tk.push_wide(key[u], u < c);
            for (uint32_t j0 = SLOTS; __any(j0 < c); j0 += SLOTS) {       // rare: more than 8 survivors in one segment
#pragma unroll
                for (uint32_t u = 0; u < SLOTS; ++u) key[u] = (j0 + u < c) ? sp[j0 + u] : KEY_PAD;
#pragma unroll
                for (uint32_t u = 0; u < SLOTS; ++u) tk.push_wide(key[u], j0 + u < c);
            }
        }
    } else {
        // LDS-tiled GEMM: one counted list per query (best list + appended survivors)
        uint32_t n_in = a.cand_count[(size_t)q * CAND_COUNT_STRIDE];
        if (n_in > a.cand_cap) {
            dropped = true;
            n_in = a.cand_cap;
        }
        constexpr int LOADS = 4;
        for (uint32_t base = 0; base < n_in; base += SCAN_THREADS * LOADS) {
            int64_t keys[LOADS];
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const uint32_t idx = base + i * SCAN_THREADS + threadIdx.x;
                keys[i] = idx < n_in ? mine[idx] : KEY_PAD;
            }
#pragma unroll
            for (int i = 0; i < LOADS; ++i) tk.push_wide(keys[i], keys[i] != KEY_PAD);
        }
    }
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    if (__any(dropped) && lane == 0) a.overflow[q] = 1u;
    __syncthreads();
    block_rank_merge<SCAN_WAVES>(lds, CAP, counts, kp, fin);
    __syncthreads();   // all reads of the old list happened before the first barrier
    for (int t = (int)threadIdx.x; t < kp; t += SCAN_THREADS) mine[t] = fin[t];
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < SCAN_WAVES; ++w) total += counts[w];
        a.cand_count[(size_t)q * CAND_COUNT_STRIDE] = (uint32_t)(total < kp ? total : kp);
        const int64_t last = fin[kp - 1];
        a.tau[q] = (last == KEY_PAD) ? __builtin_inff() : key_distance(last);
    }
}

hipError_t launch_tighten(const TightenArgs& a, hipStream_t st) {
    if (a.kp <= 32)
        hipLaunchKernelGGL((tighten_kernel<128>), dim3(a.nq), dim3(SCAN_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((tighten_kernel<256>), dim3(a.nq), dim3(SCAN_THREADS), 0, st, a);
    return hipGetLastError();
}

// Fill tau with +inf and zero the counters / overflow flags for a new batch.
__global__ void batch_reset_kernel(float* tau, uint32_t* cand_count, uint32_t* overflow, uint32_t nq, uint32_t nq_pad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq_pad) {
        tau[i] = (i < nq) ? __builtin_inff() : -__builtin_inff();  // padding queries admit nothing
        cand_count[(size_t)i * CAND_COUNT_STRIDE] = 0u;
        overflow[i] = 0u;
    }
}

hipError_t launch_batch_reset(float* tau, uint32_t* cand_count, uint32_t* overflow, uint32_t nq, uint32_t nq_pad,
                              hipStream_t st) {
    hipLaunchKernelGGL(batch_reset_kernel, dim3((nq_pad + 255) / 256), dim3(256), 0, st, tau, cand_count, overflow, nq, nq_pad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Exact f32 re-score of the candidates with scan_kernel's lane mapping and summation order.
template <int METRIC>
__device__ inline float finish_distance_b(float acc, float nrm, float q_norm) {
    float d;
    if (METRIC == BM_COS) {
        const float vn = sqrtf(nrm);
        const float sim = (vn > 1e-6f && q_norm > 1e-6f) ? acc / (vn * q_norm) : 0.0f;
        d = 1.0f - sim;
    } else if (METRIC == BM_DOT) {
        d = 1.0f - acc;
    } else {
        d = acc;
    }
    d = (d != d) ? __builtin_inff() : d;
    return d + 0.0f;
}

template <int METRIC>
__device__ inline void accumulate_b(const f32x4& q, const f32x4& v, f32x4& acc, f32x4& nrm) {
    if (METRIC == BM_L2) {
        const f32x4 e = q - v;
        acc = __builtin_elementwise_fma(e, e, acc);
    } else {
        acc = __builtin_elementwise_fma(q, v, acc);
        if (METRIC == BM_COS) nrm = __builtin_elementwise_fma(v, v, nrm);
    }
}

__device__ inline float hsum_b(const f32x4& a) { return (a.x + a.y) + (a.z + a.w); }

template <int D4, int GROUP, int METRIC>
__global__ __launch_bounds__(256) void rescore_kernel(RescoreArgs a) {
    constexpr int LOADS = D4 / GROUP;
    constexpr int RPW = WAVE / GROUP;
    const int lane = lane_id();
    const int sub = lane / GROUP, gl = lane % GROUP;
    const uint32_t pair = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const uint32_t total = a.nq * (uint32_t)a.kp;
    const bool in_range = pair < total;
    const uint32_t p = in_range ? pair : total - 1;
    const uint32_t qslot = p / (uint32_t)a.kp;
    const uint32_t q = a.qlist ? a.qlist[qslot] : qslot;
    bool live;
    uint32_t grow, lrow;
    if (a.rows != nullptr) {  // listed rows (filtered search)
        live = in_range;
        lrow = a.rows[p];
        grow = a.row_base + lrow;
    } else {
        const int64_t ck = a.cand[(size_t)(a.by_slot ? qslot : q) * a.cand_cap + (p - qslot * (uint32_t)a.kp)];
        live = in_range && ck != KEY_PAD;
        grow = key_row(ck);
        lrow = grow - a.row_base;
    }
    if (a.by_slot && __ballot(live) == 0ull) {   // padded tails of a retry round's dense lists: nothing to fetch for this wave
        if (in_range && gl == GROUP - 1) a.exact[(size_t)qslot * (uint32_t)a.kp + (p - qslot * (uint32_t)a.kp)] = KEY_PAD;
        return;
    }
    lrow = (live && lrow < a.n_rows) ? lrow : 0;
    const f32x4* __restrict__ v4 = reinterpret_cast<const f32x4*>(a.store) + (size_t)lrow * D4 + gl;
    const f32x4* __restrict__ q4 = reinterpret_cast<const f32x4*>(a.queries) + (size_t)q * D4 + gl;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LOADS; ++j) accumulate_b<METRIC>(q4[j * GROUP], v4[j * GROUP], acc, nrm);
    const float s = group_sum<GROUP>(hsum_b(acc));
    float m = 0.f;
    if (METRIC == BM_COS) m = group_sum<GROUP>(hsum_b(nrm));
    const float d = finish_distance_b<METRIC>(s, m, a.q_norm[q]);
    if (in_range && gl == GROUP - 1) {
        if (a.dist_out != nullptr) a.dist_out[p] = d;
        else a.exact[(size_t)(a.by_slot ? qslot : q) * (uint32_t)a.kp + (p - qslot * (uint32_t)a.kp)] = live ? make_key(d, grow) : KEY_PAD;
    }
}

template <int METRIC>
__global__ __launch_bounds__(256) void rescore_generic_kernel(RescoreArgs a) {
    const int lane = lane_id();
    const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t total = a.nq * (uint32_t)a.kp;
    if (pair >= total) return;  // whole wave exits together
    const uint32_t q = pair / (uint32_t)a.kp;
    bool live;
    uint32_t grow, lrow;
    if (a.rows != nullptr) {
        live = true;
        lrow = a.rows[pair];
        grow = a.row_base + lrow;
    } else {
        const int64_t ck = a.cand[(size_t)q * a.cand_cap + (pair - q * (uint32_t)a.kp)];
        live = ck != KEY_PAD;
        grow = key_row(ck);
        lrow = grow - a.row_base;
    }
    // (survivor-area mode and query lists are served by the specialised kernel only: launch_rescore refuses them here)
    lrow = (live && lrow < a.n_rows) ? lrow : 0;
    const uint32_t D = a.dims;
    const float* row = a.store + (size_t)lrow * D;
    const float* qv = a.queries + (size_t)q * D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
    if ((D & 3u) == 0) {
        const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
        const f32x4* q4 = reinterpret_cast<const f32x4*>(qv);
        for (uint32_t c = lane; c < (D >> 2); c += WAVE) accumulate_b<METRIC>(q4[c], row4[c], acc, nrm);
    } else {
        for (uint32_t c = lane; c < D; c += WAVE) {
            const f32x4 qq = {qv[c], 0.f, 0.f, 0.f};
            const f32x4 vv = {row[c], 0.f, 0.f, 0.f};
            accumulate_b<METRIC>(qq, vv, acc, nrm);
        }
    }
    const float s = group_sum<64>(hsum_b(acc));
    float m = 0.f;
    if (METRIC == BM_COS) m = group_sum<64>(hsum_b(nrm));
    const float d = finish_distance_b<METRIC>(s, m, a.q_norm[q]);
    if (lane == WAVE - 1) {
        if (a.dist_out != nullptr) a.dist_out[pair] = d;
        else a.exact[pair] = live ? make_key(d, grow) : KEY_PAD;
    }
}

template <int D4, int GROUP>
static hipError_t launch_rescore_t(const RescoreArgs& a, int metric, hipStream_t st) {
    constexpr int RPW = WAVE / GROUP;
    const uint32_t total = a.nq * (uint32_t)a.kp;
    const dim3 grid((total + 4 * RPW - 1) / (4 * RPW));
    switch (metric) {
        case BM_COS: hipLaunchKernelGGL((rescore_kernel<D4, GROUP, BM_COS>), grid, dim3(256), 0, st, a); break;
        case BM_DOT: hipLaunchKernelGGL((rescore_kernel<D4, GROUP, BM_DOT>), grid, dim3(256), 0, st, a); break;
        case BM_L2: hipLaunchKernelGGL((rescore_kernel<D4, GROUP, BM_L2>), grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_rescore(const RescoreArgs& a, int metric, hipStream_t st) {
    switch (a.dims) {  // must mirror launch_scan's (D4, GROUP) table so distances are bit-identical
        case 64: return launch_rescore_t<16, 16>(a, metric, st);
        case 128: return launch_rescore_t<32, 32>(a, metric, st);
        case 256: return launch_rescore_t<64, 64>(a, metric, st);
        case 384: return launch_rescore_t<96, 32>(a, metric, st);
        case 512: return launch_rescore_t<128, 64>(a, metric, st);
        case 768: return launch_rescore_t<192, 64>(a, metric, st);
        case 1024: return launch_rescore_t<256, 64>(a, metric, st);
        case 1536: return launch_rescore_t<384, 64>(a, metric, st);
        default: break;
    }
    if (a.qlist != nullptr || a.by_slot) return hipErrorInvalidValue;   // query lists: specialised dims only
    const uint32_t total = a.nq * (uint32_t)a.kp;
    const dim3 grid((total + 3) / 4);
    switch (metric) {
        case BM_COS: hipLaunchKernelGGL((rescore_generic_kernel<BM_COS>), grid, dim3(256), 0, st, a); break;
        case BM_DOT: hipLaunchKernelGGL((rescore_generic_kernel<BM_DOT>), grid, dim3(256), 0, st, a); break;
        case BM_L2: hipLaunchKernelGGL((rescore_generic_kernel<BM_L2>), grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Per query: order the kp exact keys, emit the best k as hits, and certify.
__global__ __launch_bounds__(256) void finalize_batch_kernel(const int64_t* __restrict__ cand, uint32_t cand_cap,
                                                             const uint32_t* __restrict__ overflow,
                                                             const int64_t* __restrict__ exact, int kp, int k,
                                                             const float* __restrict__ eps,
                                                             const uint64_t* __restrict__ ids, uint32_t row_base,
                                                             uint32_t n_rows, wax_hip_hit* __restrict__ out,
                                                             uint32_t out_stride, uint32_t* __restrict__ certified) {
    __shared__ int64_t keys[FUSED_MAX_K];
    __shared__ int64_t sorted[FUSED_MAX_K];
    const uint32_t q = blockIdx.x;
    const int t = (int)threadIdx.x;
    if (t < kp) {
        keys[t] = exact[(size_t)q * kp + t];
        sorted[t] = KEY_PAD;
    }
    __syncthreads();
    if (t < kp) {
        const int64_t mine = keys[t];
        if (mine != KEY_PAD) {
            int rank = 0;
            for (int j = 0; j < kp; ++j) rank += (keys[j] < mine || (keys[j] == mine && j < t)) ? 1 : 0;
            sorted[rank] = mine;
        }
    }
    __syncthreads();
    for (uint32_t o = threadIdx.x; o < out_stride; o += 256) {   // the row is padded to out_stride
        wax_hip_hit h;
        h.key = ((int)o < k) ? sorted[o] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - row_base;
            h.frame_id = (ids != nullptr && local < n_rows) ? ids[local] : (uint64_t)key_row(h.key);
        }
        out[(size_t)q * out_stride + o] = h;
    }
    if (t == 0) {
        const int64_t last_cand = cand[(size_t)q * cand_cap + (kp - 1)];
        uint32_t ok;
        if (overflow[q] != 0u) {
            ok = 0;  // a candidate list overflowed: some candidates were dropped
        } else if (last_cand == KEY_PAD) {
            ok = 1;  // fewer than kp rows exist: every row was re-scored exactly
        } else {
            const float a_max = key_distance(last_cand);      // k'-th smallest approx distance
            const int64_t kth = sorted[k - 1];
            const float tau = (kth == KEY_PAD) ? __builtin_inff() : key_distance(kth);
            ok = (a_max - eps[q] > tau) ? 1u : 0u;             // strict: ties stay uncertified
        }
        certified[q] = ok;
    }
}

hipError_t launch_finalize_batch(const int64_t* cand, uint32_t cand_cap, const uint32_t* overflow, const int64_t* exact,
                                 int kp, int k, const float* eps,
                                 const uint64_t* ids, uint32_t row_base, uint32_t n_rows, uint32_t nq,
                                 wax_hip_hit* out, uint32_t out_stride, uint32_t* certified, hipStream_t st) {
    if (kp > FUSED_MAX_K || k > kp || k < 1 || out_stride < (uint32_t)k) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finalize_batch_kernel, dim3(nq), dim3(256), 0, st, cand, cand_cap, overflow, exact, kp, k, eps, ids,
                       row_base, n_rows, out, out_stride, certified);
    return hipGetLastError();
}


// ===========================================================================
// One-pass batched pipeline (large stores). The slab pipeline above tightens every query's threshold between
// geometrically growing slabs: 4 GEMM launches + 4 latency-bound tighten launches per batch at 1M rows. Here the
// threshold comes from a SAMPLE instead, so the store is filtered in ONE uninterrupted GEMM launch:
//   batch_prep_kernel      queries (already in HBM) -> bf16 block, exact norms, certificate bounds, per-batch state
//   sampling GEMM          ~1/64 of the tiles, spread evenly over the store (a sorted / clustered corpus is sampled
//                          across its whole range): per (tile, query) the best similarity of the tile
//   pick_tau_kernel        tau_q = 1 - (rank-th best tile maximum): at least `rank` sampled rows pass it, and about
//                          rank * tiles / sampled_tiles rows of the whole store (~8 k', see batch_onepass_plan)
//   filtering GEMM         batch_gemm_rega_kernel / batch_gemm_ksplit_kernel over ALL tiles with that fixed tau;
//                          survivors into per-workgroup segments
//   batch_finish_kernel    per query: survivors -> best k' (approximate keys) -> exact f32 re-score (scan_kernel's
//                          arithmetic) -> top-k + certificate, one launch
// Exactness is unchanged: the candidate set is "the k' smallest approximate distances of the whole store" (or every
// row below tau when fewer than k' pass), every non-candidate's approximate distance is >= a_max, and the certificate
// a_max - eps > exact k-th decides whether the answer is provably exact; anything else is re-run on the exact path.

// The register-resident filtering GEMMs (survivors into per-workgroup segments): cosine / dot at these dimensions.
bool batch_onepass_fast(uint32_t dims, int metric) {
    return metric != BM_L2 && (dims == 128 || dims == 256 || dims == 384 || dims == 512 || dims == 768);
}
// Everything else the MFMA path serves (L2; any other multiple of 64, e.g. 1024 / 1536) runs the same one-pass pipeline on
// the LDS-tiled 128 x 128 kernel: survivors are appended to one counted list per query.
bool batch_onepass_dims(uint32_t dims, int metric) {
    return (dims % 64u) == 0 && dims >= 64 && metric >= BM_COS && metric <= BM_L2;
}
uint32_t batch_tile_rows(uint32_t dims, int metric) {
    if (!batch_onepass_fast(dims, metric)) return (uint32_t)GN;
    return dims == 768 ? 32u : 64u;
}

__global__ __launch_bounds__(256) void batch_prep_kernel(PrepArgs a) {
    constexpr uint32_t CHUNK = 1024;                     // floats staged per wave at a time
    __shared__ float stage_s[4][CHUNK];
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t q = blockIdx.x * 4 + (uint32_t)wave;
    if (q >= a.nq_pad) return;
    if (q == 0 && a.tile_ctr) for (uint32_t i = lane; i < BATCH_TILE_CTRS * 32u; i += WAVE) a.tile_ctr[i] = 0u;   // one counter per 128-byte line
    const uint32_t D = a.dims;
    unsigned short* out = a.qb + (size_t)q * D;
    if (q >= a.nq) {                                     // padding query: admits nothing, matches nothing
        for (uint32_t c = lane; c < D; c += WAVE) out[c] = 0;
        if (lane == 0) {
            a.q_n2[q] = 0.f; a.q_norm[q] = 0.f; a.eps[q] = 0.f; a.tau[q] = -__builtin_inff(); a.overflow[q] = 0u;
            if (a.cand_count) a.cand_count[(size_t)q * CAND_COUNT_STRIDE] = 0u;
        }
        return;
    }
    const float* row = a.queries + (size_t)q * D;
    float* st = stage_s[wave];
    // (1) f32 norm for the bf16 block (approximate path only; same arithmetic as mirror_kernel) and
    // (2) the exact ||q|| exactly as the host computes it for the single-query path (engine.hip query_norm): four f64
    // partial sums over j = c, c + 4, ... in ascending order (every product of two floats is exact in f64, so a fused
    // multiply-add rounds like multiply-then-add), a tail into the first, (s0 + s1) + (s2 + s3), sqrt, one rounding
    // to f32. Lanes 0..3 each run one chain, reading the row from LDS (staged with coalesced loads: a chain of
    // dependent global loads cost 12 us per launch).
    float acc = 0.f;
    double part = 0.0;
    const uint32_t d4 = D & ~3u;
    for (uint32_t c0 = 0; c0 < D; c0 += CHUNK) {
        const uint32_t len = (D - c0 < CHUNK) ? D - c0 : CHUNK;
        wave_lds_fence();
        for (uint32_t c = lane; c < len; c += WAVE) st[c] = row[c0 + c];
        wave_lds_fence();
        if (lane < 4) {
            const uint32_t lim = (c0 + len <= d4) ? len : (d4 > c0 ? d4 - c0 : 0u);   // the 4-aligned part of this chunk
#pragma unroll 8
            for (uint32_t j = (uint32_t)lane; j < lim; j += 4) part += (double)st[j] * (double)st[j];
            if (lane == 0)
                for (uint32_t j = lim; j < len; ++j) part += (double)st[j] * (double)st[j];   // tail (last chunk only)
        }
    }
    for (uint32_t c = lane; c < D; c += WAVE) acc = fmaf(row[c], row[c], acc);
    acc = group_sum<64>(acc);
    acc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));
    const float nf = sqrtf(acc);
    const float scale = (a.metric == BM_COS) ? ((nf > 1e-6f) ? 1.0f / nf : 0.0f) : 1.0f;
    double qe2 = 0.0;                                    // ||x - bf16(x)||^2 of the scaled query block row x (exact differences)
    for (uint32_t c = lane; c < D; c += WAVE) {
        const float x = row[c] * scale;
        const unsigned short b = f32_to_bf16_rne(x);
        const double d = (double)x - (double)__uint_as_float((unsigned int)b << 16);
        qe2 += d * d;
        out[c] = b;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) qe2 += __shfl_xor(qe2, o);
    const double s0 = __shfl(part, 0), s1 = __shfl(part, 1), s2 = __shfl(part, 2), s3 = __shfl(part, 3);
    if (lane == 0) {
        const double total = (s0 + s1) + (s2 + s3);
        float c = (float)sqrt(total);
        // make the f32 result independent of the last bit of the device's f64 sqrt: c must be the float nearest to
        // sqrt(total); the midpoints to its neighbours are 25-bit numbers whose squares are exact in f64
        const float cp = nextafterf(c, 0.0f), cn = nextafterf(c, __builtin_inff());
        const double ml = 0.5 * ((double)c + (double)cp), mu = 0.5 * ((double)c + (double)cn);
        if (ml * ml > total) c = cp;
        else if (mu * mu < total) c = cn;
        a.q_norm[q] = c;
        if (a.q_norm_host) a.q_norm_host[q] = c;
        a.q_n2[q] = acc;
        // Certificate bound: |approximate distance - exact distance|. With x_q, x_v the f32 vectors that were rounded (normalised for
        // cosine) and q~, v~ their bf16 roundings: |q~.v~ - x_q.x_v| <= ||q~ - x_q|| ||v~|| + ||x_q|| ||v~ - x_v|| (Cauchy-Schwarz on the
        // ERROR vectors). Round 4 uses the MEASURED error norms: ||q~ - x_q|| is this query's (f64 over exact differences, above),
        // ||v~ - x_v|| is bounded by its maximum over the rows, measured when the mirror was built (a.max_row_err, + 0.1 % for its f32
        // accumulation); ||v~|| <= max||v|| (1 + 2^-8). On top: the MFMA accumulates D products in f32 and the f32 normalisation of
        // either side is off by at most D 2^-25 + 2^-23 relative (3 D 2^-24 of the product of the norms covers both), and the exact
        // distance it is compared with carries ~1e-6 of its own. Typically 0.0035 for unit vectors at D = 384.
        // The fallback when no measurement is at hand ("batch_eps_measured" = 0) is the worst case: bf16 keeps 8 significant bits,
        // so rounding moves an element by at most 2^-8 relative and a product of two rounded elements by 2^-7 (1 + 2^-9) — 0.0078.
        // (Rounds 1-3 used 2^-8 (1 + 2^-10) here, i.e. a unit roundoff of 2^-9 per operand: half of the true worst case. Random
        // rounding errors are two orders of magnitude below either, which is why no test ever saw it; errors that line up with the
        // query could have defeated it. The measured bound is rigorous AND about what the old constant was.)
        const double qn_d = a.metric == BM_COS ? 1.0 + 1e-6 : (double)c;
        const double vn_d = a.metric == BM_COS ? 1.0 + 1e-6 : (double)a.max_norm;
        const double u = 0.0078125 * (1.0 + 1.0 / 512.0) + (double)D * 5.97e-8 + 1e-6;          // worst case, relative to ||q|| max||v||
        double dot_err = u * qn_d * vn_d * 1.001;
        if (a.max_row_err > 0.f) {
            const double measured = sqrt(qe2) * vn_d * (1.0 + 1.0 / 256.0) + qn_d * (double)a.max_row_err * 1.001 +
                                    3.0 * (double)D * 5.97e-8 * qn_d * vn_d;
            if (measured < dot_err) dot_err = measured;
        }
        float eps;
        if (a.metric == BM_COS) eps = (float)(dot_err + 3e-6);
        else if (a.metric == BM_DOT) eps = (float)(dot_err + 1e-6 * (1.0 + qn_d * vn_d));
        else {   // L2: ||q||^2 + ||v||^2 - 2 q.v carries the factor 2 and the norms' own rounding
            const double ss = (double)c * (double)c + (double)a.max_norm * (double)a.max_norm;
            eps = (float)(2.0 * dot_err + 4e-6 * (1.0 + ss));
        }
        eps = nextafterf(eps, __builtin_inff());             // the double -> float conversion may have rounded down
        a.eps[q] = eps;
        a.tau[q] = __builtin_inff();
        a.overflow[q] = 0u;
        if (a.cand_count) a.cand_count[(size_t)q * CAND_COUNT_STRIDE] = 0u;
    }
}

hipError_t launch_batch_prep(const PrepArgs& a, hipStream_t st) {
    if (a.nq_pad == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_prep_kernel, dim3((a.nq_pad + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_rega_sample(const GemmArgs& a, hipStream_t st) {
    constexpr int AHEAD = D >= 512 ? 1 : 3;
    constexpr size_t smem = (size_t)rega_lds_tiles<D>(false) * 64 * (D * 2 + 16) + 3 * 8 * 32 * 4 + 16;
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_rega_kernel<D, false, AHEAD, true>), smem, configured);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.nqt * 128 + 255) / 256;
    uint32_t pg = 256 / groups;
    if (pg < 1) pg = 1;
    if (pg > a.sample_tiles) pg = a.sample_tiles;
    hipLaunchKernelGGL((batch_gemm_rega_kernel<D, false, AHEAD, true>), dim3(groups * pg), dim3(512), smem, st, a, pg);
    return hipGetLastError();
}

hipError_t launch_batch_gemm_sample(const GemmArgs& a, int metric, hipStream_t st) {
    if (!batch_onepass_dims(a.dims, metric) || a.tile_max == nullptr || a.sample_tiles == 0) return hipErrorInvalidValue;
    if (!batch_onepass_fast(a.dims, metric)) {   // LDS-tiled kernel: one workgroup per (sampled tile, 128 queries)
        const dim3 grid(a.sample_tiles * a.nqt);
        switch (metric) {
            case BM_COS: hipLaunchKernelGGL((batch_gemm_kernel<BM_COS, true>), grid, dim3(256), 0, st, a); break;
            case BM_DOT: hipLaunchKernelGGL((batch_gemm_kernel<BM_DOT, true>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((batch_gemm_kernel<BM_L2, true>), grid, dim3(256), 0, st, a); break;
        }
        return hipGetLastError();
    }
    switch (a.dims) {
        case 128: return launch_rega_sample<128>(a, st);
        case 256: return launch_rega_sample<256>(a, st);
        case 384: return launch_rega_sample<384>(a, st);
        case 512: return launch_rega_sample<512>(a, st);
        case 768: {
            constexpr int D = 768;
            constexpr size_t smem = 2 * 32 * (D * 2 + 16) + 2 * 4 * (4 * 64 * 16) + 3 * 4 * 32 * 4 + 16;
            static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
            {
                hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_ksplit_kernel<D, 4, true>), smem, configured);
                if (e != hipSuccess) return e;
            }
            const uint32_t groups = a.nqt;
            uint32_t pg = 256 / groups;
            if (pg < 1) pg = 1;
            if (pg > a.sample_tiles) pg = a.sample_tiles;
            hipLaunchKernelGGL((batch_gemm_ksplit_kernel<D, 4, true>), dim3(groups * pg), dim3(512), smem, st, a, pg);
            return hipGetLastError();
        }
        default: break;
    }
    return hipErrorInvalidValue;
}

// One-pass pipeline, step 3: per query, the admission threshold of the filtering GEMM from the sampled tile maxima:
// tau_sim = the `rank`-th largest of the `sample_tiles` per-tile best similarities (rank <= PICK_J). A sampled row above
// the threshold puts its tile's maximum above it, and the top few rows of a sample sit in distinct tiles, so the
// number of SAMPLED rows above tau_sim is ~rank and the number in the whole store is Gamma(rank) / f (f = sampled
// fraction): relative spread 1/sqrt(rank) (29 % at rank 12), against ~50 % with a heavy lower tail for the minimum of
// a few group maxima used before — what lets the planner aim at ~3 k' survivors instead of ~10 k' for the same risk of
// a threshold that admits fewer than k rows. (Two top rows sharing a tile only loosen the threshold.)
// Workgroup = 32 queries x 32 slices. Phase 1: thread (query, slice) keeps the PICK_J largest of its tiles i = slice,
// slice + 32, ... in a sorted register list (coalesced: 32 consecutive queries per tile row). Phase 2, through LDS:
// a wave takes two queries, lane = slice; `rank` rounds of {maximum of the 32 list heads by DPP, the first lane holding
// it pops}; the last maximum is the answer. ~4 us; a full sort per query — the first version — was an 18 us chain.
constexpr int PICK_J = 12;
__global__ __launch_bounds__(1024) void pick_tau_kernel(const float* __restrict__ tile_max, uint32_t sample_tiles,
                                                        uint32_t nq, uint32_t nq_pad, uint32_t rank,
                                                        float* __restrict__ tau, int negated) {
    __shared__ float lists[PICK_J][32][33];                   // [position][slice][query]
    const uint32_t qi = threadIdx.x & 31u, slice = threadIdx.x >> 5;
    const uint32_t q = blockIdx.x * 32u + qi;                 // < nq_pad (tile_max rows are nq_pad wide)
    float top[PICK_J];
#pragma unroll
    for (int p = 0; p < PICK_J; ++p) top[p] = -__builtin_inff();
    constexpr uint32_t U = 16;                                // 512 sampled tiles: every load of a thread in flight at once
    for (uint32_t i0 = slice; i0 < sample_tiles; i0 += 32u * U) {
        float v[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t i = i0 + 32u * u;
            v[u] = (i < sample_tiles) ? tile_max[(size_t)i * nq_pad + q] : -__builtin_inff();
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            float x = v[u];
            if (!(x > top[PICK_J - 1])) continue;             // also drops NaN
#pragma unroll
            for (int p = 0; p < PICK_J; ++p) {                // insertion: top stays sorted, descending
                const float hi = __builtin_fmaxf(top[p], x);
                x = __builtin_fminf(top[p], x);
                top[p] = hi;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PICK_J; ++p) lists[p][slice][qi] = top[p];
    __syncthreads();
    // wave w: queries 2w, 2w + 1; lane l: query 2w + (l >> 5), slice l & 31
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t q2 = 2u * wave + (lane >> 5);
    float mine[PICK_J];
#pragma unroll
    for (int p = 0; p < PICK_J; ++p) mine[p] = lists[p][lane & 31u][q2];
    float kth = -__builtin_inff();
    for (uint32_t r = 0; r < rank; ++r) {
        const float gm = group_max32(mine[0]);                // lanes 31 / 63 hold their group's maximum
        const float m_lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), 31));
        const float m_hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), 63));
        kth = (lane & 32u) ? m_hi : m_lo;
        const unsigned long long holders = __ballot(mine[0] == kth) >> (lane & 32u) & 0xffffffffull;
        const bool pop = holders != 0ull && (uint32_t)__builtin_ctzll(holders) == (lane & 31u) && kth > -__builtin_inff();
        if (pop) {
#pragma unroll
            for (int p = 0; p + 1 < PICK_J; ++p) mine[p] = mine[p + 1];
            mine[PICK_J - 1] = -__builtin_inff();
        }
    }
    const uint32_t qg = blockIdx.x * 32u + q2;
    if ((lane & 31u) == 0u && qg < nq) {
        // fewer than `rank` finite tile maxima (NaN query / tiny sample): no threshold => +inf, the GEMM marks the query
        // for the exact path instead of admitting the whole store
        // similarities are dot products (cosine / dot: distance = 1 - sim) or, `negated` (L2), minus the distance itself
        tau[qg] = (kth > -__builtin_inff()) ? (negated ? -kth : 1.0f - kth) : __builtin_inff();
    }
}

hipError_t launch_pick_tau(const float* tile_max, uint32_t sample_tiles, uint32_t nq, uint32_t nq_pad, uint32_t rank,
                           float* tau, int metric, hipStream_t st) {
    if (rank < 1 || rank > (uint32_t)PICK_J || sample_tiles == 0 || (nq_pad % 32u) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pick_tau_kernel, dim3((nq + 31) / 32), dim3(1024), 0, st, tile_max, sample_tiles, nq, nq_pad, rank, tau,
                       metric == BM_L2 ? 1 : 0);
    return hipGetLastError();
}

// Gather one query's survivors from the GEMM workgroups' segments into the wave-private lists.
template <int CAP>
__device__ inline bool gather_segments(WaveTopK<CAP>& tk, const int64_t* __restrict__ mine, const uint32_t* __restrict__ seg_count,
                                       uint32_t nseg, uint32_t seg_slots, uint32_t nq_pad, uint32_t q, uint32_t count_stride) {
    bool dropped = false;
    if (count_stride != 0u) {
        // counted list (LDS-tiled filtering GEMM): ONE list of seg_slots keys per query, its (unclamped) length at
        // seg_count[q * count_stride]; all threads share it
        uint32_t c = seg_count[(size_t)q * count_stride];
        if (c > seg_slots) {
            dropped = true;
            c = seg_slots;
        }
        constexpr uint32_t LOADS = 4;
        for (uint32_t base = 0; base < c; base += SCAN_THREADS * LOADS) {
            int64_t key[LOADS];
#pragma unroll
            for (uint32_t u = 0; u < LOADS; ++u) {
                const uint32_t idx = base + u * SCAN_THREADS + threadIdx.x;
                key[u] = idx < c ? mine[idx] : KEY_PAD;
            }
#pragma unroll
            for (uint32_t u = 0; u < LOADS; ++u) tk.push_wide(key[u], key[u] != KEY_PAD);
        }
        return dropped;
    }
    for (uint32_t sb = 0; sb < nseg; sb += SCAN_THREADS) {
        const uint32_t seg = sb + threadIdx.x;
        uint32_t c = (seg < nseg) ? seg_count[(size_t)seg * nq_pad + q] : 0u;
        if (c > seg_slots) {
            dropped = true;
            c = seg_slots;
        }
        const int64_t* __restrict__ sp = mine + (size_t)seg * seg_slots;
        constexpr uint32_t SLOTS = 8;
        for (uint32_t j0 = 0; __any(j0 < c); j0 += SLOTS) {
            int64_t key[SLOTS];
#pragma unroll
            for (uint32_t u = 0; u < SLOTS; ++u) key[u] = (j0 + u < c) ? sp[j0 + u] : KEY_PAD;
#pragma unroll
            for (uint32_t u = 0; u < SLOTS; ++u) tk.push_wide(key[u], j0 + u < c);
        }
    }
    return dropped;
}

template <int D4, int GROUP, int METRIC>
// At most 80 VGPRs (6 waves per SIMD): the filtering GEMM of the NEXT batch in flight leaves exactly that much of
// every SIMD's register file free (2 waves x 216), so this kernel's workgroups can run beside it instead of behind it.
__global__ __launch_bounds__(SCAN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void batch_finish_kernel(FinishArgs a) {
    constexpr int CAP = 256;
    constexpr int LOADS = D4 / GROUP;
    constexpr int RPW = WAVE / GROUP;
    __shared__ int64_t lds[SCAN_WAVES * CAP + SCAN_WAVES + 2 * FUSED_MAX_K];
    int* counts = reinterpret_cast<int*>(lds + SCAN_WAVES * CAP);
    int64_t* fin = lds + SCAN_WAVES * CAP + SCAN_WAVES;     // [kp] best approximate keys, ascending
    int64_t* ex = fin + FUSED_MAX_K;                        // [kp] their exact keys
    int64_t* sorted = lds;                                  // [kp] exact keys ascending (the wave lists are dead by then)
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t q = blockIdx.x;
    const int kp = a.kp;
    WaveTopK<CAP> tk;
    tk.init(lds + wave * CAP, kp);
    const bool dropped = gather_segments<CAP>(tk, a.cand + (size_t)q * a.cand_cap, a.seg_count, a.nseg, a.seg_slots, a.nq_pad, q, a.count_stride);
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    const int any_dropped = __syncthreads_or(dropped ? 1 : 0);
    block_rank_merge<SCAN_WAVES>(lds, CAP, counts, kp, fin);
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_WAVES; ++w) total += counts[w];
    const int m = total < kp ? total : kp;                  // candidates to re-score
    __syncthreads();                                        // everyone has read counts / the lists before `sorted` reuses them
    // exact f32 distance of every candidate with scan_kernel's (D4, GROUP) lane mapping and summation order
    {
        const int sub = lane / GROUP, gl = lane % GROUP;
        const float qn = a.q_norm[q];
        const f32x4* __restrict__ q4 = reinterpret_cast<const f32x4*>(a.queries) + (size_t)q * D4 + gl;
        f32x4 qv[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) qv[j] = q4[j * GROUP];
        // U row fetches in flight per lane group (one dependent HBM round trip per U candidates instead of per candidate)
        constexpr int U = 2;
        for (int c0 = wave * RPW; c0 < m; c0 += SCAN_WAVES * RPW * U) {
            int64_t ck[U];
            f32x4 v[U][LOADS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * SCAN_WAVES * RPW + sub;
                ck[u] = fin[c < m ? c : m - 1];
                uint32_t lrow = key_row(ck[u]) - a.row_base;
                lrow = lrow < a.n_rows ? lrow : 0;
                const f32x4* __restrict__ v4 = reinterpret_cast<const f32x4*>(a.store) + (size_t)lrow * D4 + gl;
#pragma unroll
                for (int j = 0; j < LOADS; ++j) v[u][j] = v4[j * GROUP];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * SCAN_WAVES * RPW + sub;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < LOADS; ++j) accumulate_b<METRIC>(qv[j], v[u][j], acc, nrm);
                const float s = group_sum<GROUP>(hsum_b(acc));
                float mm = 0.f;
                if (METRIC == BM_COS) mm = group_sum<GROUP>(hsum_b(nrm));
                const float d = finish_distance_b<METRIC>(s, mm, qn);
                if (c < m && gl == GROUP - 1) ex[c] = make_key(d, key_row(ck[u]));
            }
        }
    }
    __syncthreads();
    const int t = (int)threadIdx.x;
    if (t < m) {                                            // keys are unique (distinct rows): rank = number of smaller keys
        const int64_t mine = ex[t];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (ex[j] < mine) ? 1 : 0;
        sorted[rank] = mine;
    }
    __syncthreads();
    const int k = a.k;
    for (uint32_t o = threadIdx.x; o < a.out_stride; o += SCAN_THREADS) {   // the row is padded to out_stride
        wax_hip_hit h;
        h.key = ((int)o < k && (int)o < m) ? sorted[o] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - a.row_base;
            h.frame_id = (a.ids != nullptr && local < a.n_rows) ? a.ids[local] : (uint64_t)key_row(h.key);
        }
        a.out[(size_t)q * a.out_stride + o] = h;
    }
    if (t == 0) {
        uint32_t ok = 0;
        if (!any_dropped && a.overflow[q] == 0u && m >= k) {
            // every row outside the candidate set has an approximate distance >= a_max: the kp-th best approximate
            // distance when the list is full, else the admission threshold itself (everything below it is a candidate)
            // (the one-wave-per-SIMD GEMM admits on the conservative bound, i.e. slightly past tau: rows it rejected have
            // d > tau, rows beyond the kp-th candidate have d >= that candidate's)
            const float a_max = (total >= kp) ? __builtin_fminf(key_distance(fin[kp - 1]), a.tau[q]) : a.tau[q];
            const float kth = key_distance(sorted[k - 1]);
            ok = (a_max - a.eps[q] > kth) ? 1u : 0u;        // strict: ties stay uncertified
        }
        a.certified[q] = ok;
        // device-side copy of the flag for batch_retry_kernel (launched behind this kernel when the engine expects failures);
        // 2 = "retry pointless": something was dropped, or every survivor was a candidate already
        if (a.cert_dev != nullptr) a.cert_dev[q] = ok != 0u ? 1u : ((any_dropped || a.overflow[q] != 0u || total <= kp || m < k) ? 2u : 0u);
    }
}

// Device-side full retry (round 4; second rung of the exactness ladder without a host round trip). One 1 024-thread workgroup per
// query, launched right behind batch_finish_kernel on the batch's stream; a workgroup whose query is certified (cert_dev != 0)
// exits at once. Otherwise the first finish failed although nothing was dropped — a dense neighbourhood: more rows inside the bf16
// error band of the k-th neighbour than the k' candidates cover. EVERY row the filtering GEMM admitted is still in the query's
// segments, so all of them are re-scored exactly (scan_kernel's lane mapping and summation order: bit-identical distances), each
// exact key goes straight into its wave's top-k list, the sixteen lists are merged by rank and the k best written; with every
// survivor re-scored the certificate only needs tau - eps > the exact k-th. Round 3 did this from the host at collect time
// (three launches and three synchronisations per batch, queued behind the NEXT batch's GEMM: 2.1 x the batch time at k = 100 on
// a clustered corpus). The engine launches this kernel only while recent batches had uncertified queries ("retry hint"), so a
// well-separated corpus never pays for the extra launch; the host-driven rung stays behind it for whatever is left.
constexpr int RETRY_WAVES = 16;
constexpr int RETRY_LIST = 8192;                            // survivors (rows) listed in LDS per pass
template <int D4, int GROUP, int METRIC>
__global__ __launch_bounds__(RETRY_WAVES * 64) void batch_retry_kernel(FinishArgs a) {
    constexpr int CAP = 256;
    constexpr int LOADS = D4 / GROUP;
    constexpr int RPW = WAVE / GROUP;
    constexpr int NT = RETRY_WAVES * 64;
    __shared__ int64_t lds[RETRY_WAVES * CAP + RETRY_WAVES + FUSED_MAX_K];
    __shared__ uint32_t rows_l[RETRY_LIST];                 // global rows of the survivors of the current pass
    __shared__ uint32_t n_list, seg_next;
    int* counts = reinterpret_cast<int*>(lds + RETRY_WAVES * CAP);
    int64_t* best = lds + RETRY_WAVES * CAP + RETRY_WAVES;
    const uint32_t q = blockIdx.x;
    if (a.cert_dev[q] != 0u) return;                        // certified by the first finish, or not retryable (workgroup-uniform)
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int k = a.k;
    WaveTopK<CAP> tk;
    tk.init(lds + wave * CAP, k);
    const int sub = lane / GROUP, gl = lane % GROUP;
    const float qn = a.q_norm[q];
    const f32x4* __restrict__ q4 = reinterpret_cast<const f32x4*>(a.queries) + (size_t)q * D4 + gl;
    f32x4 qv[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) qv[j] = q4[j * GROUP];
    const int64_t* __restrict__ mine = a.cand + (size_t)q * a.cand_cap;
    const uint32_t nseg = a.count_stride != 0u ? 1u : a.nseg;
    // Which survivors can still matter: the first finish left its exact k-th distance in the hit row (m >= k here: cert_dev would be 2
    // otherwise). A survivor whose APPROXIMATE distance is beyond it by more than eps has an exact distance beyond it too (the
    // certificate's own inequality, `approx - eps > kth`, and in the same float form), and the final k-th can only be smaller — so
    // only the others are listed and re-scored. At k = 100 on the clustered corpus that is a few hundred rows of 1.5 KB per query
    // instead of every survivor (~2 600).
    const int64_t kth_key = a.out[(size_t)q * a.out_stride + (size_t)(k - 1)].key;
    const bool prune = kth_key != KEY_PAD && a.retry_all == 0u;
    const float kth0 = key_distance(kth_key), eps_q = a.eps[q];
    constexpr int U = LOADS >= 4 ? 2 : 4;                   // row fetches in flight per lane group
    // Passes: (1) the threads list the rows of as many whole segments as fit RETRY_LIST — one thread per segment, so the
    // dependent chain "count -> keys" runs for all segments at once instead of segment after segment; (2) the sixteen waves
    // re-score the listed rows, U * RPW rows per wave and step with all their loads in flight. A query's survivors almost
    // always fit one pass (the planner aims at a few hundred to a few thousand).
    if (threadIdx.x == 0) seg_next = 0u;
    __syncthreads();
    for (;;) {
        const uint32_t seg0 = seg_next;                     // first segment of this pass (workgroup-uniform)
        if (seg0 >= nseg) break;
        __syncthreads();
        if (threadIdx.x == 0) n_list = 0u;
        __syncthreads();
        // segments seg0 .. seg0 + NT - 1, one per thread; a segment is taken only if it fits (order does not matter: keys carry rows)
        const uint32_t seg = seg0 + threadIdx.x;
        uint32_t c = 0u;
        if (seg < nseg) {
            c = a.count_stride != 0u ? a.seg_count[(size_t)q * a.count_stride] : a.seg_count[(size_t)seg * a.nq_pad + q];
            c = c < a.seg_slots ? c : a.seg_slots;          // (no segment overflowed: cert_dev would be 2)
        }
        // inclusive prefix over the workgroup's segments (wave scan + wave totals through LDS)
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) counts[wave] = (int)inc;
        __syncthreads();
        uint32_t base = 0u;
        for (int w = 0; w < wave; ++w) base += (uint32_t)counts[w];
        const uint32_t end = base + inc;                    // rows listed up to and including this thread's segment
        const bool fits = seg < nseg && end <= (uint32_t)RETRY_LIST;
        if (fits && c > 0u) {
            const int64_t* __restrict__ sp = mine + (size_t)seg * a.seg_slots;
            for (uint32_t j = 0; j < c; ++j) {
                const int64_t key = sp[j];
                if (prune && key_distance(key) - eps_q > kth0) continue;
                rows_l[atomicAdd(&n_list, 1u)] = key_row(key);   // (order does not matter: the exact keys are unique and selected by value)
            }
        }
        // the pass takes the longest prefix of segments that fits; a single segment larger than the list cannot be retried here
        const unsigned long long fm = __ballot(fits || seg >= nseg);
        const int wave_fit = fm == ~0ull ? 64 : __builtin_ctzll(~fm);   // leading threads of this wave that fit
        __syncthreads();                                    // counts[] read by everyone above
        if (lane == 0) counts[wave] = wave_fit;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t taken = 0u;
            for (int w = 0; w < RETRY_WAVES; ++w) { taken += (uint32_t)counts[w]; if (counts[w] < 64) break; }
            seg_next = seg0 + taken;
        }
        __syncthreads();
        const uint32_t taken_to = seg_next;
        if (taken_to == seg0) {                             // nothing fits (one huge segment): leave the query to the host rungs
            if (threadIdx.x == 0) a.certified[q] = 0u;
            return;
        }
        const uint32_t n = n_list;                          // rows of segments [seg0, taken_to): the largest prefix end that fits
        for (uint32_t c0 = (uint32_t)wave * RPW * U; c0 < n; c0 += RETRY_WAVES * RPW * U) {   // wave-uniform trip count
            uint32_t grow[U];
            f32x4 v[U][LOADS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ci = c0 + (uint32_t)(u * RPW + sub);
                grow[u] = rows_l[ci < n ? ci : n - 1];
                uint32_t lrow = grow[u] - a.row_base;
                lrow = lrow < a.n_rows ? lrow : 0;
                const f32x4* __restrict__ v4 = reinterpret_cast<const f32x4*>(a.store) + (size_t)lrow * D4 + gl;
#pragma unroll
                for (int j = 0; j < LOADS; ++j) v[u][j] = v4[j * GROUP];
            }
            tk.make_room(RPW * U);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ci = c0 + (uint32_t)(u * RPW + sub);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < LOADS; ++j) accumulate_b<METRIC>(qv[j], v[u][j], acc, nrm);
                const float s2 = group_sum<GROUP>(hsum_b(acc));
                float mm = 0.f;
                if (METRIC == BM_COS) mm = group_sum<GROUP>(hsum_b(nrm));
                const float d = finish_distance_b<METRIC>(s2, mm, qn);
                tk.push(make_key(d, grow[u]), ci < n && gl == GROUP - 1);
            }
        }
    }
    tk.finalize();
    __syncthreads();
    if (lane == 0) counts[wave] = tk.cnt;
    __syncthreads();
    block_rank_merge<RETRY_WAVES>(lds, CAP, counts, k, best);
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < RETRY_WAVES; ++w) total += counts[w];
    for (uint32_t o = threadIdx.x; o < a.out_stride; o += NT) {
        wax_hip_hit h;
        h.key = ((int)o < k) ? best[o] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - a.row_base;
            h.frame_id = (a.ids != nullptr && local < a.n_rows) ? a.ids[local] : (uint64_t)key_row(h.key);
        }
        a.out[(size_t)q * a.out_stride + o] = h;
    }
    if (threadIdx.x == 0) {
        const bool okr = total >= k && best[k - 1] != KEY_PAD && a.tau[q] - a.eps[q] > key_distance(best[k - 1]);   // strict: ties with a rejected row stay uncertified
        a.certified[q] = okr ? 2u : 0u;                     // 2 = certified by the device-side retry (the host counts them)
    }
}

template <int D4, int GROUP>
static hipError_t launch_retry_t(const FinishArgs& a, int metric, hipStream_t st) {
    switch (metric) {
        case BM_COS: hipLaunchKernelGGL((batch_retry_kernel<D4, GROUP, BM_COS>), dim3(a.nq), dim3(RETRY_WAVES * 64), 0, st, a); break;
        case BM_DOT: hipLaunchKernelGGL((batch_retry_kernel<D4, GROUP, BM_DOT>), dim3(a.nq), dim3(RETRY_WAVES * 64), 0, st, a); break;
        case BM_L2: hipLaunchKernelGGL((batch_retry_kernel<D4, GROUP, BM_L2>), dim3(a.nq), dim3(RETRY_WAVES * 64), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// The device-side full retry behind a fused finish (same arguments; a.cert_dev must be the array that finish wrote).
bool batch_retry_dims(uint32_t dims) { return dims == 128 || dims == 256 || dims == 384 || dims == 512 || dims == 768; }
hipError_t launch_batch_retry(const FinishArgs& a, int metric, hipStream_t st) {
    if (a.nq == 0) return hipSuccess;
    if (a.cert_dev == nullptr || a.k < 1 || a.k > FUSED_MAX_K || a.qlist != nullptr) return hipErrorInvalidValue;   // any k' (fused or three-launch finish), k <= 192
    switch (a.dims) {   // (D4, GROUP) as in launch_batch_finish / launch_scan
        case 128: return launch_retry_t<32, 32>(a, metric, st);
        case 256: return launch_retry_t<64, 64>(a, metric, st);
        case 384: return launch_retry_t<96, 32>(a, metric, st);
        case 512: return launch_retry_t<128, 64>(a, metric, st);
        case 768: return launch_retry_t<192, 64>(a, metric, st);
        default: return hipErrorInvalidValue;
    }
}

// Large k' (193 .. 960): the same steps as three launches.
template <int CAP>
__global__ __launch_bounds__(SCAN_THREADS) void select_segments_kernel(FinishArgs a, uint32_t* __restrict__ overflow_out) {
    extern __shared__ __attribute__((aligned(16))) int64_t lds_dyn[];   // [SCAN_WAVES * CAP + SCAN_WAVES]
    int* counts = reinterpret_cast<int*>(lds_dyn + SCAN_WAVES * CAP);
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t q = a.qlist ? a.qlist[blockIdx.x] : blockIdx.x;
    const int kp = a.kp;
    WaveTopK<CAP> tk;
    tk.init(lds_dyn + wave * CAP, kp);
    const bool dropped = gather_segments<CAP>(tk, a.cand + (size_t)q * a.cand_cap, a.seg_count, a.nseg, a.seg_slots, a.nq_pad, q, a.count_stride);
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    const int any_dropped = __syncthreads_or(dropped ? 1 : 0);
    block_rank_merge<SCAN_WAVES>(lds_dyn, CAP, counts, kp, a.sel + (size_t)q * kp);   // pads with KEY_PAD
    if (threadIdx.x == 0 && any_dropped) overflow_out[q] = 1u;
}

// exact[q][0..kp) -> sorted, top-k hits, certificate; sel[q][kp-1] != KEY_PAD <=> the candidate list is full.
__global__ __launch_bounds__(1024) void finalize_big_kernel(FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) int64_t lds_dyn[];   // [2 * kp]
    const int kp = a.kp, k = a.k;
    int64_t* keys = lds_dyn;
    int64_t* sorted = lds_dyn + kp;
    const uint32_t q = a.qlist ? a.qlist[blockIdx.x] : blockIdx.x;
    for (int t = (int)threadIdx.x; t < kp; t += 1024) {
        keys[t] = a.exact[(size_t)q * kp + t];
        sorted[t] = KEY_PAD;
    }
    __syncthreads();
    for (int t = (int)threadIdx.x; t < kp; t += 1024) {
        const int64_t mine = keys[t];
        if (mine == KEY_PAD) continue;
        int rank = 0;
        for (int j = 0; j < kp; ++j) rank += (keys[j] < mine) ? 1 : 0;   // unique keys (distinct rows); PAD is the maximum
        sorted[rank] = mine;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < a.out_stride; t += 1024) {   // the row is padded to out_stride
        wax_hip_hit h;
        h.key = ((int)t < k) ? sorted[t] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - a.row_base;
            h.frame_id = (a.ids != nullptr && local < a.n_rows) ? a.ids[local] : (uint64_t)key_row(h.key);
        }
        a.out[(size_t)q * a.out_stride + t] = h;
    }
    if (threadIdx.x == 0) {
        uint32_t ok = 0;
        const int64_t kth_key = sorted[k - 1];
        if (a.overflow[q] == 0u && kth_key != KEY_PAD) {
            const int64_t last = a.sel[(size_t)q * kp + (kp - 1)];
            const float a_max = (last != KEY_PAD) ? __builtin_fminf(key_distance(last), a.tau[q]) : a.tau[q];
            ok = (a_max - a.eps[q] > key_distance(kth_key)) ? 1u : 0u;
        }
        a.certified[q] = ok;
        // flag for batch_retry_kernel (see batch_finish_kernel): 2 = a retry is pointless — survivors were dropped, or fewer than k' passed
        // the filter (every survivor was a candidate already)
        if (a.cert_dev != nullptr)
            a.cert_dev[q] = ok != 0u ? 1u : ((a.overflow[q] != 0u || kth_key == KEY_PAD || a.sel[(size_t)q * kp + (kp - 1)] == KEY_PAD) ? 2u : 0u);
    }
}

template <int D4, int GROUP>
static hipError_t launch_finish_t(const FinishArgs& a, int metric, hipStream_t st) {
    if (metric == BM_COS) hipLaunchKernelGGL((batch_finish_kernel<D4, GROUP, BM_COS>), dim3(a.nq), dim3(SCAN_THREADS), 0, st, a);
    else if (metric == BM_DOT) hipLaunchKernelGGL((batch_finish_kernel<D4, GROUP, BM_DOT>), dim3(a.nq), dim3(SCAN_THREADS), 0, st, a);
    else hipLaunchKernelGGL((batch_finish_kernel<D4, GROUP, BM_L2>), dim3(a.nq), dim3(SCAN_THREADS), 0, st, a);
    return hipGetLastError();
}

// Dimensions with a fused finish kernel (= a specialised scan kernel); any other multiple of 64 takes the three-launch form
// below, whose re-score has a generic-dims kernel.
bool batch_finish_fused_dims(uint32_t dims) {
    return dims == 64 || dims == 128 || dims == 256 || dims == 384 || dims == 512 || dims == 768 || dims == 1024 || dims == 1536;
}

hipError_t launch_batch_finish(const FinishArgs& a, int metric, hipStream_t st) {
    if (a.nq == 0) return hipSuccess;
    if (!batch_onepass_dims(a.dims, metric) || a.k < 1 || a.k > a.kp) return hipErrorInvalidValue;
    if (a.kp <= FUSED_MAX_K && a.qlist == nullptr && batch_finish_fused_dims(a.dims)) {
        switch (a.dims) {   // (D4, GROUP) must mirror launch_scan's table: distances bit-identical to the single-query path
            case 64: return launch_finish_t<16, 16>(a, metric, st);
            case 128: return launch_finish_t<32, 32>(a, metric, st);
            case 256: return launch_finish_t<64, 64>(a, metric, st);
            case 384: return launch_finish_t<96, 32>(a, metric, st);
            case 512: return launch_finish_t<128, 64>(a, metric, st);
            case 768: return launch_finish_t<192, 64>(a, metric, st);
            case 1024: return launch_finish_t<256, 64>(a, metric, st);
            case 1536: return launch_finish_t<384, 64>(a, metric, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (a.kp > 960 || a.sel == nullptr || a.exact == nullptr) return hipErrorInvalidValue;
    {
        constexpr int CAP = 1024;
        constexpr size_t smem = (size_t)(SCAN_WAVES * CAP + SCAN_WAVES) * sizeof(int64_t);
        hipLaunchKernelGGL((select_segments_kernel<CAP>), dim3(a.nq), dim3(SCAN_THREADS), smem, st, a,
                           const_cast<uint32_t*>(a.overflow));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    RescoreArgs r{};
    r.store = a.store; r.queries = a.queries; r.q_norm = a.q_norm; r.cand = a.sel; r.exact = a.exact;
    r.n_rows = a.n_rows; r.row_base = a.row_base; r.dims = a.dims; r.nq = a.nq; r.cand_cap = (uint32_t)a.kp; r.kp = a.kp;
    r.qlist = a.qlist;
    hipError_t e = launch_rescore(r, metric, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_big_kernel, dim3(a.nq), dim3(1024), (size_t)2 * a.kp * sizeof(int64_t), st, a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Full retry, step 1: pack the live survivors of a query (the filled front of each of its segments) into one dense list.
__global__ __launch_bounds__(256) void compact_survivors_kernel(CompactArgs a) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t run_base;
    const uint32_t slot = blockIdx.x;
    const uint32_t q = a.qlist[slot];
    const int64_t* __restrict__ src = a.cand + (size_t)q * a.cand_cap;
    int64_t* __restrict__ dst = a.dense + (size_t)slot * a.dense_stride;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) run_base = 0u;
    __syncthreads();
    uint32_t total = 0;
    if (a.count_stride != 0u) {   // one counted list: a straight copy
        uint32_t c = a.seg_count[(size_t)q * a.count_stride];
        c = c < a.seg_slots ? c : a.seg_slots;
        for (uint32_t i = threadIdx.x; i < c; i += 256u) dst[i] = src[i];
        total = c;
    } else {
        for (uint32_t s0 = 0; s0 < a.nseg; s0 += 256u) {
            const uint32_t seg = s0 + threadIdx.x;
            uint32_t c = seg < a.nseg ? a.seg_count[(size_t)seg * a.nq_pad + q] : 0u;
            c = c < a.seg_slots ? c : a.seg_slots;
            uint32_t inc = c;                                  // inclusive scan over the block's 256 segments
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            if (lane == 63) wave_tot[wave] = inc;
            __syncthreads();
            uint32_t off = run_base;
            for (int w = 0; w < wave; ++w) off += wave_tot[w];
            off += inc - c;
            const int64_t* sp = src + (size_t)seg * a.seg_slots;
            for (uint32_t j = 0; j < c; ++j) dst[off + j] = sp[j];
            __syncthreads();
            if (threadIdx.x == 0) run_base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            __syncthreads();
        }
        total = run_base;
    }
    for (uint32_t i = total + threadIdx.x; i < a.dense_stride; i += 256u) dst[i] = KEY_PAD;
    if (threadIdx.x == 0) a.live_out[slot] = total;
}

hipError_t launch_compact_survivors(const CompactArgs& a, hipStream_t st) {
    if (a.n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_survivors_kernel, dim3(a.n_slots), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Full retry, step 3 (second rung of the exactness ladder): per uncertified query, the exact keys of ALL its survivors -> top-k hits +
// certificate. One workgroup per query; the survivor area (up to 32 K keys, mostly dead) streams through the wave lists.
__global__ __launch_bounds__(SCAN_THREADS) void full_retry_select_kernel(FullRetryArgs a) {
    constexpr int CAP = 1024;
    extern __shared__ __attribute__((aligned(16))) int64_t lds_dyn[];   // [SCAN_WAVES * CAP + SCAN_WAVES + 512]
    int* counts = reinterpret_cast<int*>(lds_dyn + SCAN_WAVES * CAP);
    int64_t* fin = lds_dyn + SCAN_WAVES * CAP + SCAN_WAVES;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t slot = blockIdx.x;
    const uint32_t q = a.qlist[slot];
    const int k = a.k;
    // did any segment overflow? (then survivors were dropped and nothing can be certified)
    bool over = false;
    if (a.count_stride != 0u) {
        over = a.seg_count[(size_t)q * a.count_stride] > a.seg_slots;
    } else {
        for (uint32_t seg = threadIdx.x; seg < a.nseg; seg += SCAN_THREADS) over |= a.seg_count[(size_t)seg * a.nq_pad + q] > a.seg_slots;
    }
    WaveTopK<CAP> tk;
    tk.init(lds_dyn + wave * CAP, k);
    const int64_t* __restrict__ mine = a.exact + (size_t)slot * a.area;
    constexpr uint32_t LOADS = 8;
    for (uint32_t base = 0; base < a.area; base += SCAN_THREADS * LOADS) {
        int64_t key[LOADS];
#pragma unroll
        for (uint32_t u = 0; u < LOADS; ++u) {
            const uint32_t i = base + u * SCAN_THREADS + threadIdx.x;
            key[u] = i < a.area ? mine[i] : KEY_PAD;
        }
#pragma unroll
        for (uint32_t u = 0; u < LOADS; ++u) tk.push_wide(key[u], key[u] != KEY_PAD);
    }
    tk.finalize();
    if (lane == 0) counts[wave] = tk.cnt;
    const int any_over = __syncthreads_or(over ? 1 : 0);
    block_rank_merge<SCAN_WAVES>(lds_dyn, CAP, counts, k, fin);
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_WAVES; ++w) total += counts[w];
    for (uint32_t o = threadIdx.x; o < a.out_stride; o += SCAN_THREADS) {
        wax_hip_hit h;
        h.key = ((int)o < k) ? fin[o] : KEY_PAD;
        h.frame_id = ID_PAD;
        if (h.key != KEY_PAD) {
            const uint32_t local = key_row(h.key) - a.row_base;
            h.frame_id = (a.ids != nullptr && local < a.n_rows) ? a.ids[local] : (uint64_t)key_row(h.key);
        }
        a.out[(size_t)q * a.out_stride + o] = h;
    }
    if (threadIdx.x == 0) {
        uint32_t ok = 0;
        if (!any_over && a.overflow[q] == 0u && total >= k && fin[k - 1] != KEY_PAD)
            ok = (a.tau[q] - a.eps[q] > key_distance(fin[k - 1])) ? 1u : 0u;   // strict: ties with a rejected row stay uncertified
        a.certified[q] = ok;
    }
}

hipError_t launch_full_retry_select(const FullRetryArgs& a, hipStream_t st) {
    if (a.n_slots == 0) return hipSuccess;
    if (a.k < 1 || a.k > 960 || a.out_stride < (uint32_t)a.k || a.area == 0) return hipErrorInvalidValue;
    constexpr size_t smem = (size_t)(SCAN_WAVES * 1024 + SCAN_WAVES + 1024) * sizeof(int64_t);
    hipLaunchKernelGGL(full_retry_select_kernel, dim3(a.n_slots), dim3(SCAN_THREADS), smem, st, a);
    return hipGetLastError();
}

}  // namespace wax
