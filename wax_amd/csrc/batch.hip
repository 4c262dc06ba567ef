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
//                        slab. Two kernels:
//     batch_gemm_rq_kernel      D in {128, 256, 384, 512, 768}, cosine / dot: queries resident in VGPRs as A fragments,
//                               corpus tiles by LDS-DMA, survivors into per-workgroup segments (no global atomics)
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
// row_list != nullptr (round 6: rows overwritten by an upsert): work item i converts row row_list[i] of src into row row_list[i]
// of dst / norm2 instead of row i.
__global__ __launch_bounds__(256) void mirror_kernel(const float* __restrict__ src, uint32_t n_rows,
                                                     uint32_t n_rows_padded, uint32_t dims, int normalize,
                                                     unsigned short* __restrict__ dst, float* __restrict__ norm2,
                                                     unsigned int* __restrict__ max_norm_bits,
                                                     const uint32_t* __restrict__ row_list) {
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
    for (uint32_t item = gwave; item < n_rows_padded; item += nwaves) {
        const uint32_t r = row_list != nullptr ? row_list[item] : item;
        unsigned short* out = dst + (size_t)r * dims;
        if (item >= n_rows) {
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
                       dst, norm2, max_norm_bits, (const uint32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_mirror_rows(const float* src, const uint32_t* d_rows, uint32_t n_listed, uint32_t dims, int normalize,
                              unsigned short* dst, float* norm2, unsigned int* max_norm_bits, hipStream_t st) {
    if (n_listed == 0) return hipSuccess;
    uint64_t blocks = ((uint64_t)n_listed + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mirror_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, n_listed, n_listed, dims, normalize,
                       dst, norm2, max_norm_bits, d_rows);
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
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

template <int N>
__device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------
// Register-resident-queries GEMM ("rq"): cosine / dot at D in {128, 256, 384, 512, 768}; the filtering launch of the one-pass
// pipeline, its sampling launch (SAMPLE), and every slab after the first of the slab pipeline.
//
// The query block is tiny and reused against every corpus row, so it never goes through LDS: each of a workgroup's 8 waves keeps
// its 32 queries x D as MFMA A fragments in VGPRs for the whole launch (D / 4 registers: 192 at D = 768). Only the corpus
// streams: persistent workgroups (one per CU; 256 queries per group of workgroups) walk TROWS-row tiles of the bf16 mirror through
// a ring of NBUF padded tile images in LDS (row stride 2 D + 16 bytes: conflict-free ds_read_b128), filled by LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass). All 8 waves read the same B fragments: one ds_read_b128 per MFMA,
// RB = TROWS / 32 accumulators per wave, KS * RB MFMAs (v_mfma_f32_32x32x16_bf16) per wave and tile. The selection is fused:
// per tile 16 RB sign tests against conservative per-query bounds; a survivor takes a slot in THIS workgroup's segment of its
// query's candidate row (plain 8-byte store, no global atomics); the workgroup's 256 per-query survivor counters live in LDS.
//
// What round 5 changed (one kernel instead of four; D = 768 from 0.40 - 0.45 to 0.49 - 0.52 of the dense bf16 peak, same answers):
//  1. NOTHING in the tile loop is an LDS access the compiler can see, and the A fragments are "used" before the first DMA request.
//     hipcc's waitcnt pass cannot tell which LDS bytes a pending global_load_lds will write, nor count the requests of a loop it has
//     not unrolled: the round-4 768-d kernel carried `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of MFMA 0 of EVERY tile — it requested
//     tile t + 2 and then waited for it to land before its first MFMA; the three-buffer ring never had anything in flight across
//     a tile. Here B fragments and bounds are read by inline-asm ds_read_b128 with hand-counted lgkmcnt waits (LDS operations
//     return in order: "at most N younger ones outstanding" implies that read f has landed, whatever else is queued), and the only
//     vmcnt waits in the loop are the counted ones of dma_wait. (-6.5 % at D = 768.)
//  2. A DMA request is four VALU instructions, or one. The padded image is cut into 1-KB pieces; lane l of piece P fills slot
//     64 P + l. Slot -> global offset is 16 (slot - slot / SLOTS_PER_ROW) from a wave-uniform tile base (saddr form), with no clamps:
//     a pad slot fetches the first bytes of the next row, the slack behind the image the row after the tile, a tile that runs past
//     the store whatever the mirror holds there — all inside the mirror's allocation (capacity + BATCH_MIRROR_SLACK_ROWS) and never used
//     (rows >= slab_end are masked by the selection; a garbage row only pollutes its own output column). Where registers allow
//     (D <= 512) the per-piece offsets stay in VGPRs. The round-4 form cost ~14 VALU + spilled-SGPR traffic per piece, 49 pieces per
//     tile: -7 % at D = 768; with it the LDS-DMA kernel also overtook the register-staged one at D = 384.
//  3. SPLIT (filtering launches): the tile barrier is "arrive" (a wave adds 1 to an LDS counter behind its K loop, once its own DMA
//     pieces of the next tile have landed) and "wait" (spin on the counter in front of the next K loop, BEFORE requesting the tile
//     that overwrites the buffer the previous K loop read): an early wave's selection sits between the two. (-1.5 ... -2.5 %.)
//     The spin is bounded; a wave that gives up poisons its workgroup's survivor counts, which sends those queries to the exact
//     path — never a silent wrong answer.
// Measured and NOT kept (profiles/HISTORY.md, round 5): putting the two waves of a SIMD half a tile period apart ("ping-pong":
// two barriers per tile, one wave multiplies while its partner requests and selects; also with all requests on the late half and
// with fragment rings primed ahead of the barriers) — equal or slower at both dimensions: under any schedule the chip delivers the
// same matrix rate at this power, what counts is the instructions and bytes per MFMA; deeper / shallower read-ahead: no change.
// The two waves of a SIMD still work in opposite order (waves 0-3: K loop, then select; waves 4-7: select the previous tile, then K
// loop), which is worth 11 %.
//
// SAMPLE = true (one-pass pipeline, threshold estimation): instead of filtering, the workgroup visits `a.sample_tiles` tiles spread
// evenly over the slab (logical index i -> tile i * ntiles / sample_tiles) and records, per query, the best similarity of each
// visited tile in a.tile_max[i][query] (pick_tau_kernel turns the j-th best tile maximum into the query's admission threshold).
template <int D, int TROWS, int NBUF, int AHEAD, bool SAMPLE, bool SPLIT, bool NT = false, bool PROF = false>
__global__ __launch_bounds__(512, 2) void batch_gemm_rq_kernel(GemmArgs a, uint32_t blocks_per_group) {
    // PROF (diagnosis build of the filtering launch, "batch_prof_ptr"): every wave accumulates the shader cycles (s_memtime) it spends in
    // each phase of the tile loop in SGPRs and leaves them in a.prof[(workgroup * 8 + wave) * RQ_PROF_WORDS ...]; same answers, ~10 % slower.
    unsigned int ph[RQ_PROF_WORDS];
#pragma unroll
    for (int i = 0; i < RQ_PROF_WORDS; ++i) ph[i] = 0u;
    auto now = [&]() -> unsigned int {
        if constexpr (PROF) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned int v = (unsigned int)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            return v;
        } else {
            return 0u;
        }
    };
    unsigned long long rt0 = 0ull;
    if constexpr (PROF) rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned int k_entry = now();
    constexpr int KS = D / 16;                       // MFMA k-steps
    constexpr int RB = TROWS / 32;                   // 32-row blocks per tile = accumulators per wave
    constexpr int NF = KS * RB;                      // B fragments (= MFMAs) per wave and tile; fragment f = (k-step f / RB, block f % RB)
    constexpr int ROW_B = D * 2 + 16;                // LDS row stride (bytes); (ROW_B / 4) % 64 == 4
    constexpr int SEG_PER_ROW = D * 2 / 16;          // 16-byte segments per row
    constexpr int SLOTS_PER_ROW = ROW_B / 16;        // 16-byte slots per padded row
    constexpr int IMG_B = TROWS * ROW_B;             // padded tile image
    constexpr int PIECES = (IMG_B + 1023) / 1024;    // 1-KB DMA pieces per tile
    constexpr int BUF_B = PIECES * 1024;             // buffer stride (the image + slack for the last piece)
    constexpr int PPW = (PIECES + 7) / 8;            // pieces per wave
    constexpr int FULL_WAVES = PIECES % 8 == 0 ? 8 : PIECES % 8;   // waves below this index carry PPW pieces, the others PPW - 1
    constexpr int PRE = NBUF - 1;                    // tiles requested ahead of the one being read
    constexpr int RING = AHEAD + 1;
    static_assert(!(SAMPLE && SPLIT), "the sampling launch keeps the workgroup barrier");
    static_assert(TROWS % 32 == 0 && RB >= 1 && RB <= 4, "tile = 1..4 MFMA row blocks");
    static_assert(TROWS + 2 <= (int)BATCH_MIRROR_SLACK_ROWS, "the un-clamped requests of the last tile stay inside the mirror's slack rows");
    static_assert(D % 64 == 0 && (ROW_B / 4) % 64 == 4, "row stride must keep ds_read_b128 conflict-free");
    static_assert(NBUF >= 2, "one tile is read while the next lands");
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
        if (a.qf != nullptr) {   // fragment order: one contiguous 1-KB run per k-step
            const u32x4* qp = reinterpret_cast<const u32x4*>(a.qf) + (size_t)(q0 >> 5) * (KS * 64) + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa[ks] = __builtin_bit_cast(bf16x8, qp[ks * 64]);
        } else {
            const u32x4* qp = reinterpret_cast<const u32x4*>(a.qb + (size_t)(q0 + (lane & 31)) * D) + (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa[ks] = __builtin_bit_cast(bf16x8, qp[ks * 2]);
        }
        // The fragments are "used" HERE, so hipcc waits for their loads here — before the first DMA request. Left alone it places
        // that wait in front of the first MFMA of the tile loop, where the only count it can prove is vmcnt(0): every wave then
        // drains its whole DMA queue (the tile it requested a moment ago included) at the top of every K loop.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa[ks]));
    }
    if (!SAMPLE && lane < 32) {
        // conservative bound for the hot test: rows with acc >= sim_lo are admitted (a superset of `1 - acc <= tau` by less than
        // 1e-6: rejected rows still have d > tau, which is all the certificate uses)
        const float tq = a.tau[q0 + lane];
        sim_s[wave * 32 + lane] = (1.0f - tq) - 4e-7f * (1.0f + __builtin_fabsf(tq));
    }
    if (SPLIT && tid < 2) sync_s[tid] = 0u;
    if (!SAMPLE && tid < 256) cnt_s[tid] = 0u;                // the (workgroup, query) survivor counters: LDS, returning adds (select_tile's cold path)
    const uint32_t cnt_lane0 = (uint32_t)(size_t)(lds_void*)cnt_s + (uint32_t)(wave * 32 + 4 * (lane >> 5)) * 4u;
    const uint32_t seg_slots = a.seg_area / blocks_per_group;
    const uint32_t seg_lane0 = (q0 + 4u * ((uint32_t)lane >> 5)) * a.cand_cap + a.seg_base + bidx * seg_slots;

    const uint32_t ntiles_all = (a.slab_rows + TROWS - 1) / TROWS;
    const uint32_t ntiles = SAMPLE ? a.sample_tiles : ntiles_all;
    const uint32_t slab_end = a.slab0 + a.slab_rows;
    const unsigned char* cbase = reinterpret_cast<const unsigned char*>(a.cb);

    // LDS-DMA map: wave w moves pieces w, w + 8, ...; lane l of piece P fills slot P*64 + l of the padded image (point 2 above)
    const bool full_wave = wave < FULL_WAVES;
    constexpr uint32_t DIV_SHIFT = 18, DIV_MAGIC = ((1u << DIV_SHIFT) + SLOTS_PER_ROW - 1) / SLOTS_PER_ROW;
    static_assert((uint64_t)(PIECES * 64) * (DIV_MAGIC * (uint64_t)SLOTS_PER_ROW - (1u << DIV_SHIFT)) < (1u << DIV_SHIFT),
                  "magic division must be exact for every slot of a tile");
    constexpr bool CACHE_OFF = !SAMPLE && D <= 512;  // per-piece offsets in VGPRs: a request is ONE instruction
    uint32_t doff[CACHE_OFF ? PPW : 1];
    if (CACHE_OFF) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const uint32_t slot = ((uint32_t)wave + 8u * (uint32_t)i) * 64u + (uint32_t)lane;
            doff[i] = (slot - ((slot * DIV_MAGIC) >> DIV_SHIFT)) * 16u;
        }
    }
    auto dma_tile = [&](uint32_t tile, uint32_t buf_off) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (i < PPW - 1 || full_wave) {
                const uint32_t P = (uint32_t)wave + 8u * (uint32_t)i;
                if constexpr (SAMPLE) {
                    // scattered tiles, the last of which may run past the store: rows are clamped to its last row (a duplicate of a
                    // real row cannot raise a tile maximum), pad and slack slots to valid bytes
                    const uint32_t row0 = a.slab0 + (uint32_t)(((unsigned long long)tile * ntiles_all) / a.sample_tiles) * TROWS;
                    uint32_t lane_o = (uint32_t)lane;
                    asm volatile("" : "+v"(lane_o));      // (recomputed per piece: no per-piece VGPRs live around the tile loop)
                    const uint32_t slot = P * 64u + lane_o;
                    uint32_t r = slot / (uint32_t)SLOTS_PER_ROW;
                    uint32_t c = slot - r * (uint32_t)SLOTS_PER_ROW;
                    c = c < (uint32_t)SEG_PER_ROW ? c : (uint32_t)SEG_PER_ROW - 1u;
                    r = r < (uint32_t)TROWS ? r : (uint32_t)TROWS - 1u;
                    uint32_t grow = row0 + r;
                    grow = grow < a.n_rows ? grow : a.n_rows - 1;
                    const unsigned char* src = cbase + (size_t)grow * (D * 2) + c * 16u;
                    __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(smem + buf_off + P * 1024u), 16, 0, 0);
                } else {
                    const unsigned char* tbase = cbase + (size_t)(a.slab0 + tile * TROWS) * (D * 2);   // wave-uniform
                    uint32_t off;
                    if constexpr (CACHE_OFF) {
                        off = doff[i];
                    } else {
                        uint32_t lane_o = (uint32_t)lane;
                        asm volatile("" : "+v"(lane_o));
                        const uint32_t slot = P * 64u + lane_o;
                        off = (slot - ((slot * DIV_MAGIC) >> DIV_SHIFT)) * 16u;
                    }
                    // NT (one query group: every tile is read exactly once): non-temporal requests
                    __builtin_amdgcn_global_load_lds((global_cvoid*)(tbase + off), (lds_void*)(smem + buf_off + P * 1024u), 16, 0, NT ? 2 : 0);
                }
            }
        }
    };
    // this wave's DMA requests still allowed in flight: PRE - 1 whole tiles, or none
    auto dma_wait = [&](bool keep) {
        if (!keep) wait_vmcnt<0>();
        else if (full_wave) wait_vmcnt<(PRE - 1) * PPW>();
        else wait_vmcnt<(PRE - 1) * (PPW - 1)>();
    };

    f32x16 acc[RB];
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_void*)smem;
    const uint32_t lane_boff = (uint32_t)(lane & 31) * (uint32_t)ROW_B + (uint32_t)(lane >> 5) * 16u;
    // K loop. Fragment f + AHEAD is requested before MFMA f; the wait in front of MFMA f leaves at most min(AHEAD, NF - 1 - f)
    // younger reads outstanding. Every step is pinned (sched_barrier) — hipcc otherwise hoists an MFMA over the asm wait it depends
    // on. s_setprio 1 lets the multiplying wave win issue arbitration against its selecting partner.
    auto mfma_tile = [&](uint32_t baddr) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 fb[RING];
        static_for<0, (AHEAD < NF ? AHEAD : NF)>([&](auto F) {
            constexpr int f = decltype(F)::value;
            u32x4(&fbr)[RING] = fb;                  // (asm operands alone do not make a generic lambda capture)
            const uint32_t ba = baddr;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbr[f % RING]) : "v"(ba), "n"((f % RB) * 32 * ROW_B + (f / RB) * 32));
        });
        __builtin_amdgcn_s_setprio(1);
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
        __builtin_amdgcn_s_setprio(0);
    };
    // Fused selection on one tile. C/D layout of the 32x32 MFMA: column = lane & 31 (corpus row of the block), row =
    // (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query of the wave). A wave's 32 queries are ITS OWN: no other wave of the workgroup selects
    // for them. Hot path: per accumulator a subtraction and a funnel shift into a per-lane pass mask, one ballot per tile; nothing set
    // (the common case) = done. Cold path: lane-parallel, one returning LDS add per survivor (below).
    const uint32_t sim_addr = (uint32_t)(size_t)(lds_void*)sim_s + (uint32_t)(wave * 32 + 4 * (lane >> 5)) * 4u;
    // the lane's sixteen bounds: in registers for the whole launch where they fit (D <= 512; 768-d holds 192 registers of A fragments),
    // else re-read from LDS per tile — one LDS round trip, ~300 cycles beside eight waves' fragment reads, on the late waves' critical
    // path (round 6: -2 % cycles per tile, -1 ... -3 % wall at D = 384: profiles/r06/g_*)
    constexpr bool BOUNDS_REGS = !SAMPLE && D <= 512;
    f32x4 lo_keep[4];
    auto select_tile = [&](uint32_t tile) {
        if (SAMPLE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[0][r];
#pragma unroll
                for (int b = 1; b < RB; ++b) v = __builtin_fmaxf(v, acc[b][r]);
                const float m = group_max32(v);
                if ((lane & 31) == 31)
                    a.tile_max[(size_t)tile * (a.nqt * 128u) + q0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))] = m;
            }
            return;
        }
        f32x4 lo[4];
        if constexpr (BOUNDS_REGS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lo[j] = lo_keep[j];
        } else {
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                f32x4(&lor)[4] = lo;
                const uint32_t sa = sim_addr;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lor[j]) : "v"(sa), "n"(j * 32));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // Hot test (round 6). Per accumulator a subtraction and a funnel shift that pushes the sign of (acc - bound) into a per-lane
        // mask (1 = fails; a NaN may read as "passes" here — the exact comparison in `emit` decides), one ballot per tile. Rounds 1-5
        // compared every accumulator into a wave mask (v_cmp -> SGPR pair -> s_or): 47 cycles per accumulator beside the partner's K
        // loop, whatever the priorities (the VALU -> SGPR -> SALU hand-over, 32 of them per 64-row tile, plus SGPR spills), ~1 400 of
        // a tile's ~5 300 cycles on the late waves' critical path (profiles/r06/a_phase_budget_*). This form: ~800.
        constexpr int NM = (RB + 1) / 2;
        unsigned fm[NM];
#pragma unroll
        for (int m_ = 0; m_ < NM; ++m_) fm[m_] = 0u;
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                fm[b >> 1] = __builtin_amdgcn_alignbit(fm[b >> 1], __float_as_uint(acc[b][r] - lo[r >> 2][r & 3]), 31u);
        unsigned pm[NM], anyp = 0u;
#pragma unroll
        for (int m_ = 0; m_ < NM; ++m_) {
            const unsigned valid = (2 * m_ + 1 < RB) ? 0xFFFFFFFFu : 0xFFFFu;   // an odd last block fills 16 bits only
            pm[m_] = ~fm[m_] & valid;
            anyp |= pm[m_];
        }
        if (__ballot(anyp != 0u) == 0ull) return;
        // Cold path (round 6): every lane with a passing bit serves ITS OWN survivor — the accumulator picked by a five-level select
        // tree on the bit's index (31 v_cndmask; registers cannot be indexed by a lane's value), the exact test (a NaN fails it), a
        // slot from the (workgroup, query) counter in LDS by a returning add, the 8-byte store — all lanes at once, one LDS round trip
        // per pass, a second pass only when some lane holds two survivors. ~850 cycles per tile with survivors whatever their number.
        // Rounds 1-5 walked the accumulators one by one with a wave ballot, mbcnt and 16 packed SGPR counters: ~450 cycles PER survivor
        // behind ~600 of steering (1 470 per tile at top-10, 1 870 at top-100 where 93 % of the tiles have one), with the VALU -> SGPR
        // -> SALU hand-overs the hot test also had. A/B of the two builds (profiles/r06/h_*): -10 % cycles per tile and -6 % kernel
        // time at top-10, -16 % / -8 % at top-100, pipelined clustered corpus at top-100 -5.5 %.
        const unsigned int c0t = now();
        uint32_t seg_o = seg_lane0;                   // opaque: the per-query row offsets are computed HERE (cold path), not
        asm volatile("" : "+v"(seg_o));               // hoisted out of the tile loop into VGPRs that do not exist
        uint32_t cnt_o = cnt_lane0;
        asm volatile("" : "+v"(cnt_o));
#pragma unroll
        for (int m_ = 0; m_ < NM; ++m_) {
            const int nbits = (2 * m_ + 1 < RB) ? 32 : 16;
            unsigned pmv = pm[m_];
            while (__ballot(pmv != 0u) != 0ull) {
                const bool has = pmv != 0u;
                const unsigned pb = (unsigned)__builtin_ctz(pmv | 0x80000000u);
                const unsigned idx = (unsigned)(nbits - 1) - (pb < (unsigned)nbits ? pb : 0u);   // element in processing order: (block & 1) * 16 + r
                float t[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) t[i] = (i < nbits) ? acc[(2 * m_ + (i >> 4)) < RB ? 2 * m_ + (i >> 4) : 0][i & 15] : 0.f;
#pragma unroll
                for (int bit = 0; bit < 5; ++bit) {
                    const bool hi = (idx >> bit) & 1u;
#pragma unroll
                    for (int i = 0; i < (16 >> bit); ++i) t[i] = hi ? t[2 * i + 1] : t[2 * i];
                }
                const float val = t[0];
                float l16[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) l16[i] = lo[i >> 2][i & 3];
#pragma unroll
                for (int bit = 0; bit < 4; ++bit) {
                    const bool hi = (idx >> bit) & 1u;
#pragma unroll
                    for (int i = 0; i < (8 >> bit); ++i) l16[i] = hi ? l16[2 * i + 1] : l16[2 * i];
                }
                const float lov = l16[0];
                const unsigned r = idx & 15u;
                const uint32_t row0 = a.slab0 + tile * TROWS + (uint32_t)(2 * m_) * 32u + (idx >> 4) * 32u + (uint32_t)(lane & 31);
                const bool take = has && row0 < slab_end && val >= lov;
                if constexpr (PROF) ph[RQP_SURVIVORS] += (unsigned)__builtin_popcountll(__ballot(take));
                if (take) {
                    const unsigned ql = (r & 3u) + 8u * (r >> 2);   // + 4 (lane >> 5): folded into the lane bases
                    unsigned slot;       // (inline assembly: no LDS access the compiler can see inside the tile loop; the wait sits inside, the result is valid behind it)
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(cnt_o + ql * 4u), "v"(1u) : "memory");
                    // (the count is not clamped: one above seg_slots tells the finish kernel that survivors were dropped)
                    if (slot < seg_slots) a.cand[seg_o + ql * a.cand_cap + slot] = make_key((1.0f - val) + 0.0f, a.row_base + row0);
                }
                pmv &= pmv - 1u;
            }
        }
        if constexpr (PROF) { ph[RQP_COLD] += now() - c0t; ph[RQP_COLD_N] += 1u; }
    };

    uint32_t t = bidx;
    {   // prologue: the first PRE tiles of this workgroup
        bool all = true;
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const uint32_t ti = t + (uint32_t)i * blocks_per_group;
            if (ti < ntiles) dma_tile(ti, (uint32_t)(i * BUF_B)); else all = false;
        }
        dma_wait(all);
        __builtin_amdgcn_s_barrier();                         // also publishes sim_s / sync_s
        asm volatile("" ::: "memory");
    }
    if constexpr (BOUNDS_REGS) {
        static_for<0, 4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            f32x4(&lor)[4] = lo_keep;
            const uint32_t sa = sim_addr;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lor[j]) : "v"(sa), "n"(j * 32));
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned int k_loop0 = now();
    if constexpr (PROF) ph[RQP_PROLOGUE] = k_loop0 - k_entry;
    const bool late = wave >= 4;
    // Pace gate (advisory). With G > 1 query groups every corpus tile is wanted G times, by the G workgroups that share a `bidx` —
    // they sit on the same XCD (block -> XCD is blockIdx % 8 and G * blocks_per_group = 256), so the second to G-th reader hit in its
    // L2 as long as the groups stay within a few tiles of each other. Over a launch of milliseconds they do not (different survivor
    // loads): config 5 whole fetched 1.13 - 1.64 x the mirror (FETCH_SIZE, profiles/r04). Every GATE_EVERY tiles wave 0 adds its
    // progress to a word shared by the G workgroups of its bidx and, if it is more than GATE_WINDOW tiles ahead of the average of
    // those that are RUNNING, sleeps until they catch up — the other seven waves wait for it at the tile hand-over.
    // The word is {workgroups running: bits 24+, sum of their progress: bits 0-23}: a workgroup enters itself when it starts and takes
    // itself (and what it had added) out when it leaves, so nobody ever waits for a workgroup that is not on a CU — round 5: when
    // several persistent GEMMs share the GPU (two batches in flight; eight shards on one device) the G workgroups of a bidx start far
    // apart, and the round-4 form (average over all G, a 1 ms spin bound) made the early ones sleep through every check: the
    // 8-shards-on-one-GPU rehearsal of config 5 took 157 ms instead of 16. Bounded (64 polls, ~70 us) and advisory: a timeout just
    // proceeds. ("batch_debug" bit 12 switches the gate off.)
    // (round 6: every 4 tiles, window 3 — round 5 had 8 / 6: with three tiles in flight per workgroup, eight workgroup sets per XCD and a
    // window of six the shared tiles no longer fitted the XCD's 4 MB of L2 once the kernel got faster: 1.06 - 1.29 x the mirror fetched)
    constexpr uint32_t GATE_EVERY = 4u, GATE_WINDOW = 3u;
    const uint32_t ngroups = gridDim.x / blocks_per_group;
    // (launches of up to GATE_MIN_TILES tiles per workgroup — ~3 ms — do not drift far enough to lose the sharing: 1.25M x 768 reads
    // 1.01 x the mirror with or without the gate, and the gate's returning atomics cost 1.5 % there)
    constexpr uint32_t GATE_MIN_TILES = 1024u;
    const bool gate = !SAMPLE && a.progress != nullptr && ngroups > 1u && (blocks_per_group & 7u) == 0u && ngroups * blocks_per_group <= 256u &&
                      ntiles > GATE_MIN_TILES * blocks_per_group && !(a.debug & 4096u);
    uint32_t* gate_word = gate ? a.progress + (bidx & 7u) * 32u + (bidx >> 3) : a.progress;   // one word per bidx, one 128-byte line per XCD
    uint32_t gate_added = 0u;                                 // (wave 0) what this workgroup has added to the progress sum
    if (gate && tid == 0) __hip_atomic_fetch_add(gate_word, 1u << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // running
    auto pace = [&](uint32_t it) {
        if (gate && wave == 0 && (it & (GATE_EVERY - 1u)) == 0u && it > 0u) {
            // (returning atomics: the value comes from wherever agent-scope atomics execute, never from a stale cache line)
            unsigned int add = GATE_EVERY;
            gate_added += GATE_EVERY;
            for (uint32_t spins = 0; spins < 64u; ++spins) {
                unsigned int word = 0u;
                if (lane == 0) word = __hip_atomic_fetch_add(gate_word, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
                word = (unsigned int)__builtin_amdgcn_readfirstlane((int)word);
                add = 0u;
                const uint32_t running = word >> 24, sum = word & 0xFFFFFFu;
                // ahead of the running workgroups' average by more than the window?  running * it - sum > running * WINDOW
                if ((int)(running * it - sum) <= (int)(running * GATE_WINDOW)) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
    };
    // SPLIT: bounded spin on the arrival counter. The counter is touched through inline assembly only (point 1 above).
    bool gave_up = SPLIT && (a.debug & 16384u) != 0 && blockIdx.x == 1 && wave == 3;   // "batch_debug" bit 14: pretend one wave timed out (tests)
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
    // Tail pool (a.dyn). Fixed shares (tile = bidx + position * blocks_per_group) leave the launch waiting for its slowest workgroups:
    // the XCDs do not run at one speed (round 6: the odd ones 3 - 4 % behind at 384-d), 15 625 tiles do not divide by 256, and a
    // workgroup that met more survivors is behind by their cold paths — at 1M x 384 x 256 the mean workgroup idled 7.5 us of 170 at
    // the end. So only positions < n_static are fixed; the rest of the slab (the last twelfth, at least four tiles per workgroup) is
    // a pool per query group: wave 0 claims position p's tile with a returning add at the top of iteration p - PRE - 1 (the value is
    // back by that iteration's DMA wait — it is older than every request the wait lets stay in flight), publishes it in LDS in front
    // of the tile barrier, and every wave reads it at the top of iteration p - PRE, where it is the tile to request. A claim past the
    // end of the slab ends the walk (so does every later one: the counter only grows). Two LDS words, alternating: the next claim is
    // published before the slowest wave need have read this one.
    constexpr bool DYN_OK = !SAMPLE && !SPLIT;
    constexpr uint32_t DYN_MIN_TILES = 24u;
    const uint32_t tiles_min = ntiles / blocks_per_group;
    // One query group only: with G groups a tile is wanted G times, by the G workgroups of one bidx on one XCD, and fixed shares keep
    // them reading it together (one HBM fetch, G - 1 L2 hits); pool tiles go to whoever is free, on any XCD — measured at 768-d x 1 024:
    // kernel -2 %, but FETCH_SIZE 1.02 -> 1.18 x the mirror (profiles/r06/n_*). Not worth the re-reads.
    const bool dyn = DYN_OK && a.dyn != nullptr && tiles_min >= DYN_MIN_TILES && gridDim.x == blocks_per_group;
    const uint32_t dyn_keep = tiles_min / 12u > 4u ? tiles_min / 12u : 4u;
    const uint32_t n_static = dyn ? tiles_min - dyn_keep : 0xffffffffu;
    const uint32_t dyn_base = n_static * blocks_per_group;       // first tile of the pool (unused without)
    const uint32_t* dyn_word = a.dyn + group * 32u;
    const uint32_t dyn_addr = sync_addr + 16u;                    // sync_s[4], sync_s[5]
    bool dyn_done = false;                                        // wave 0: a claim came back past the end
    uint32_t t_up[PRE > 1 ? PRE - 1 : 1];                         // the tiles of positions it + 1 .. it + PRE - 1 (requested, not yet current)
#pragma unroll
    for (int i = 0; i + 1 < PRE; ++i) t_up[i] = bidx + (uint32_t)(i + 1) * blocks_per_group;
    uint32_t tn_fixed = bidx + PRE * blocks_per_group;            // position it + PRE's tile while that position is fixed
    uint32_t it = 0, cur_idx = 0, t_prev = 0;
    for (; t < ntiles; ++it) {
        const uint32_t baddr = smem_lds + cur_idx * (uint32_t)BUF_B + lane_boff;
        uint32_t tn = tn_fixed;
        uint32_t claim_v = 0u;
        if (DYN_OK && dyn && it + PRE + 1u >= n_static) {        // (uniform; the last twelfth of the walk only)
            if (it + PRE >= n_static) {
                unsigned int v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(dyn_addr + ((it & 1u) << 2)) : "memory");
                tn = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            }
            // the claim for position it + PRE + 1: lane 0 of wave 0 (EXEC narrowed inside the statement — the add's result register
            // is written long after it, and a branch around it would leave hipcc a copy of the stale value to make)
            const unsigned int mask = (wave == 0 && !dyn_done) ? 1u : 0u;   // (hipcc hands an "s" input over in a VGPR here: read it inside)
            unsigned int mask_s;
            unsigned long long saved;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "v_readfirstlane_b32 %[ms], %[m]\n\t"
                         "s_mov_b32 exec_lo, %[ms]\n\t"
                         "s_mov_b32 exec_hi, 0\n\t"
                         "global_atomic_add %[v], %[zero], %[one], %[base] sc0\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [v] "+v"(claim_v), [sv] "=&s"(saved), [ms] "=&s"(mask_s)
                         : [m] "v"(mask), [zero] "v"(0u), [one] "v"(1u), [base] "s"(dyn_word)
                         : "memory");
        }
        uint32_t pre_idx = cur_idx + PRE;
        pre_idx = pre_idx >= (uint32_t)NBUF ? pre_idx - NBUF : pre_idx;
        const bool issued = tn < ntiles;
        const unsigned int p0 = now();
        if (late && it > 0) select_tile(t_prev);
        const unsigned int p1 = now();
        // every wave is through K loop it - 1 (the buffer tile tn goes to is free) and has its pieces of tile `it` in LDS
        if (SPLIT && it > 0) wait_arrivals(8u * it);
        const unsigned int p2 = now();
        if (SPLIT) pace(it);                                  // (workgroup barrier: the gate sits in front of the barrier, below)
        // Three tile buffers (D = 128, 256, 768): tile t + 2 goes to the buffer tile t - 1 left at the last hand-over and need not land
        // before the hand-over after next, so an EARLY wave starts its K loop at once and requests behind it — the matrix pipe no
        // longer idles through the early waves' requests at the top of every tile (round 6: 768-d x 1 024 queries -5 % kernel time,
        // profiles/r06/k_*). A late wave requests in front of its K loop as before (its selection sits in front of both).
        constexpr bool EARLY_AFTER = NBUF >= 3 && !SAMPLE;
        if (!(EARLY_AFTER && !late) && issued) dma_tile(tn, pre_idx * BUF_B);
        const unsigned int p3 = now();
        mfma_tile(baddr);
        const unsigned int p4 = now();
        if (EARLY_AFTER && !late && issued) dma_tile(tn, pre_idx * BUF_B);
        // tile t + 1 must have landed before the others read it (every wave waits for its own pieces, the barrier / the arrival
        // counter joins them); the PRE - 1 younger tiles stay in flight. The wait sits in FRONT of an early wave's selection:
        // vmcnt counts the selection's survivor stores too (they are older than the requests that stay in flight and complete first)
        dma_wait(issued);
        const unsigned int p5 = now();
        if (SPLIT) arrive();
        if (!late) select_tile(t);
        t_prev = t;
        const unsigned int p6 = now();
        if (DYN_OK && dyn && it + PRE + 1u >= n_static) {
            // the claim is back (dma_wait above): publish position it + PRE + 1's tile for the top of the next iteration
            asm volatile("" : "+v"(claim_v));
            if (wave == 0) {
                uint32_t tile_c = 0xffffffffu;
                if (!dyn_done) {
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim_v);
                    tile_c = dyn_base + c;
                    dyn_done = tile_c >= ntiles;
                }
                if (lane == 0) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(dyn_addr + (((it + 1u) & 1u) << 2)), "v"(tile_c) : "memory");
            }
        }
        if (!SPLIT) {
            // the pace gate of the NEXT tile, in front of the barrier: while wave 0 sleeps the barrier holds all eight waves, so nobody
            // requests ahead of the gate (with the gate at the top of the loop the late waves — and, since the early waves request
            // behind their K loop, everybody — requested a tile before the gate could bite: config 5 whole fetched 1.06 - 1.29 x the
            // mirror, 1.98 x with the split barrier; profiles/r06/l_*)
            pace(it + 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if constexpr (PROF) {
            const unsigned int p7 = now();
            ph[RQP_SELECT] += (p1 - p0) + (p6 - p5);
            ph[RQP_WAIT_ARRIVALS] += (p2 - p1) + (p7 - p6);
            ph[RQP_DMA_ISSUE] += p3 - p2;
            ph[RQP_KLOOP] += p4 - p3;
            ph[RQP_DMA_WAIT] += p5 - p4;
        }
        cur_idx = cur_idx + 1 == (uint32_t)NBUF ? 0u : cur_idx + 1;
        if constexpr (PRE > 1) {
            t = t_up[0];
#pragma unroll
            for (int i = 0; i + 2 < PRE; ++i) t_up[i] = t_up[i + 1];
            t_up[PRE - 2] = tn;
        } else {
            t = tn;
        }
        tn_fixed += blocks_per_group;
    }
    const unsigned int k_loop1 = now();
    if (late && it > 0) select_tile(t_prev);
    if (gate && tid == 0) __hip_atomic_fetch_add(gate_word, 0u - gate_added - (1u << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leaving: out of the count and the sum
    if (SPLIT && gave_up && lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(sync_addr + 4u), "v"(1u) : "memory");
    __syncthreads();
    if (!SAMPLE && tid < 256) {
        const bool poisoned = SPLIT && sync_s[1] != 0u;       // a wave gave up waiting: nothing this workgroup selected can be trusted
        a.seg_count[(size_t)bidx * (a.nqt * 128u) + group * 256u + (uint32_t)tid] = poisoned ? 0x40000000u : cnt_s[tid];
    }
    if constexpr (PROF) {
        if (a.prof != nullptr) {
            const unsigned int k_exit = now();
            ph[RQP_LOOP] = k_loop1 - k_loop0;
            ph[RQP_EPILOGUE] = k_exit - k_loop1;
            ph[RQP_TILES] = it;
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
            ph[RQP_RT0_LO] = (unsigned int)rt0; ph[RQP_RT0_HI] = (unsigned int)(rt0 >> 32);
            ph[RQP_RT1_LO] = (unsigned int)rt1; ph[RQP_RT1_HI] = (unsigned int)(rt1 >> 32);
            unsigned int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            ph[RQP_XCC] = xcc;
            if (lane == 0) {
                unsigned int* dst = a.prof + ((size_t)blockIdx.x * 8u + (unsigned)wave) * RQ_PROF_WORDS;
#pragma unroll
                for (int i = 0; i < RQ_PROF_WORDS; ++i) dst[i] = ph[i];
            }
        }
    }
}

// Per-dimension geometry of the rq kernel: rows per tile (what the one-pass planner calls a tile), LDS tile buffers, read-ahead.
// D * TROWS is 16 - 32 K elements (32 - 64 MFMAs per wave and tile); three buffers where they fit beside the neighbouring batch's
// small kernels (a workgroup of ~100 KB leaves them a third of the CU's LDS; 150 KB does not: measured in round 3 and again in
// round 5 at D = 384, where two buffers lose nothing), except at D = 768 where a 32-row tile is already 48.5 KB and the third
// buffer is what lets a request stay in flight across a tile.
// (Round 6, D = 384: 96-row tiles — 2 x 75 KB — cut 7.6 % of the cycles per row and 4 % of the Q = 1 024 kernel, but a workgroup
// that fills the CU's LDS keeps the neighbouring batches' small kernels off it: pipelined 256-query batches +2 %, dense
// neighbourhoods +6 %. 64 rows stay: profiles/r06/g_*.)
// (Round 6, D = 384 with three buffers + the early waves requesting behind their K loop: +5 % cycles per tile, pipelined batches +4 ... 7 %.)
template <int D> struct RqGeom;
template <> struct RqGeom<128> { static constexpr int TROWS = 128, NBUF = 3, AHEAD = 3; };
template <> struct RqGeom<256> { static constexpr int TROWS = 64, NBUF = 3, AHEAD = 3; };
template <> struct RqGeom<384> { static constexpr int TROWS = 64, NBUF = 2, AHEAD = 3; };
template <> struct RqGeom<512> { static constexpr int TROWS = 64, NBUF = 2, AHEAD = 3; };
template <> struct RqGeom<768> { static constexpr int TROWS = 32, NBUF = 3, AHEAD = 3; };
static bool rq_dims(uint32_t dims) { return dims == 128 || dims == 256 || dims == 384 || dims == 512 || dims == 768; }
static uint32_t rq_tile_rows(uint32_t dims) { return dims == 128 ? 128u : dims == 768 ? 32u : 64u; }
template <int D>
constexpr size_t rq_smem() {
    return (size_t)RqGeom<D>::NBUF * (((RqGeom<D>::TROWS * (D * 2 + 16)) + 1023) / 1024 * 1024) + 2 * 8 * 32 * 4 + 64;   // tile buffers, counters, bounds, split-barrier words
}

static void rq_geometry(const GemmArgs& a, uint32_t* groups, uint32_t* per_group) {
    *groups = (a.nqt * 128 + 255) / 256;        // 256 queries per group of workgroups
    const uint32_t tr = rq_tile_rows(a.dims);
    const uint32_t ntiles = (a.slab_rows + tr - 1) / tr;
    uint32_t pg = 256 / *groups;                // one persistent workgroup per CU in total
    if (pg < 1) pg = 1;
    if (pg > ntiles) pg = ntiles;
    *per_group = pg;
}

static bool rq_eligible(const GemmArgs& a, int metric) {
    return a.dense == nullptr && a.use_rega && metric != BM_L2 && rq_dims(a.dims);
}

bool batch_gemm_segments(const GemmArgs& a, int metric, uint32_t* nseg, uint32_t* seg_slots) {
    if (!rq_eligible(a, metric)) return false;
    uint32_t groups, per_group;
    rq_geometry(a, &groups, &per_group);
    *nseg = per_group;
    *seg_slots = a.seg_area / per_group;
    return true;
}

template <int D, bool SAMPLE, bool SPLIT, bool NT = false, bool PROF = false>
static hipError_t launch_rq(const GemmArgs& a, uint32_t groups, uint32_t per_group, hipStream_t st) {
    using G = RqGeom<D>;
    constexpr size_t smem = rq_smem<D>();
    static_assert(smem <= 160 * 1024, "LDS budget of one CU");
    static std::atomic<uint64_t> configured{0};   // per device (ensure_dynamic_lds)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&batch_gemm_rq_kernel<D, G::TROWS, G::NBUF, G::AHEAD, SAMPLE, SPLIT, NT, PROF>), smem, configured);
        if (e != hipSuccess) return e;
    }
    launch_kernel((batch_gemm_rq_kernel<D, G::TROWS, G::NBUF, G::AHEAD, SAMPLE, SPLIT, NT, PROF>), dim3(groups * per_group), dim3(512), smem, st, a, per_group);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_rq_filter(const GemmArgs& a, hipStream_t st) {
    uint32_t groups, per_group;
    rq_geometry(a, &groups, &per_group);
    // "batch_rega" 1 (default since round 6): a workgroup barrier per tile; 5: the split tile barrier (arrive / wait on an LDS counter),
    // the default of round 5. With the selection at a third of its round-5 cost the early waves reach the hand-over long before the
    // late ones either way, and the split barrier's polls (an LDS round trip each beside eight waves' fragment reads) cost more than
    // the overlap buys: -2.2 % (384-d, 256 queries) ... -3.4 % (768-d) with s_barrier, same answers (profiles/r06/j_*).
    const bool split = a.use_rega == 5u;
    // one query group: every tile is requested exactly once, by one workgroup -> non-temporal requests (-1 ... -3 % at Q = 256; with
    // G > 1 groups sharing tiles through their XCD's L2 the same hint costs 3 - 5 %: profiles/r05/e_nontemporal_tile_requests.txt)
    const bool nt = groups == 1;
    // "batch_prof_ptr" (diagnosis): the phase-timing build of the same launch, at the two dimensions of BASELINE configs 3 / 5
    if constexpr (D == 384 || D == 768) {
        if (a.prof != nullptr) {
            if (split) return nt ? launch_rq<D, false, true, true, true>(a, groups, per_group, st) : launch_rq<D, false, true, false, true>(a, groups, per_group, st);
            return nt ? launch_rq<D, false, false, true, true>(a, groups, per_group, st) : launch_rq<D, false, false, false, true>(a, groups, per_group, st);
        }
    }
    if (split) return nt ? launch_rq<D, false, true, true>(a, groups, per_group, st) : launch_rq<D, false, true>(a, groups, per_group, st);
    return nt ? launch_rq<D, false, false, true>(a, groups, per_group, st) : launch_rq<D, false, false>(a, groups, per_group, st);
}

hipError_t launch_batch_gemm(const GemmArgs& a, int metric, hipStream_t st) {
    // fast path: queries resident in registers (needs the query block padded to a multiple of 256 rows)
    if (rq_eligible(a, metric)) {
        switch (a.dims) {
            case 128: return launch_rq_filter<128>(a, st);
            case 256: return launch_rq_filter<256>(a, st);
            case 384: return launch_rq_filter<384>(a, st);
            case 512: return launch_rq_filter<512>(a, st);
            case 768: return launch_rq_filter<768>(a, st);
            default: break;
        }
    }
    const uint32_t ctiles = (a.slab_rows + GN - 1) / GN;
    const dim3 grid(ctiles * a.nqt);
    switch (metric) {
        case BM_COS: launch_kernel((batch_gemm_kernel<BM_COS>), grid, dim3(256), 0, st, a); break;
        case BM_DOT: launch_kernel((batch_gemm_kernel<BM_DOT>), grid, dim3(256), 0, st, a); break;
        case BM_L2: launch_kernel((batch_gemm_kernel<BM_L2>), grid, dim3(256), 0, st, a); break;
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
//   filtering GEMM         batch_gemm_rq_kernel (or the LDS-tiled batch_gemm_kernel) over ALL tiles with that fixed tau;
//                          survivors into per-workgroup segments
//   batch_finish_kernel    per query: survivors -> best k' (approximate keys) -> exact f32 re-score (scan_kernel's
//                          arithmetic) -> top-k + certificate, one launch
// Exactness is unchanged: the candidate set is "the k' smallest approximate distances of the whole store" (or every
// row below tau when fewer than k' pass), every non-candidate's approximate distance is >= a_max, and the certificate
// a_max - eps > exact k-th decides whether the answer is provably exact; anything else is re-run on the exact path.

// The register-resident filtering GEMM (survivors into per-workgroup segments): cosine / dot at these dimensions.
bool batch_onepass_fast(uint32_t dims, int metric) { return metric != BM_L2 && rq_dims(dims); }
// Everything else the MFMA path serves (L2; any other multiple of 64, e.g. 1024 / 1536) runs the same one-pass pipeline on
// the LDS-tiled 128 x 128 kernel: survivors are appended to one counted list per query.
bool batch_onepass_dims(uint32_t dims, int metric) {
    return (dims % 64u) == 0 && dims >= 64 && metric >= BM_COS && metric <= BM_L2;
}
uint32_t batch_tile_rows(uint32_t dims, int metric) {
    if (!batch_onepass_fast(dims, metric)) return (uint32_t)GN;
    return rq_tile_rows(dims);
}

__global__ __launch_bounds__(256) void batch_prep_kernel(PrepArgs a) {
    constexpr uint32_t CHUNK = 1024;                     // floats staged per wave at a time
    __shared__ float stage_s[4][CHUNK];
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t q = blockIdx.x * 4 + (uint32_t)wave;
    if (q >= a.nq_pad) return;
    if (q == 0 && a.progress) for (uint32_t i = lane; i < BATCH_PROGRESS_WORDS; i += WAVE) a.progress[i] = 0u;   // the pace gate's words (one 128-byte line per XCD)
    if (q == 0 && a.dyn) for (uint32_t i = lane; i < BATCH_DYN_WORDS; i += WAVE) a.dyn[i] = 0u;                   // the tail pool's claim counters
    const uint32_t D = a.dims;
    unsigned short* out = a.qb + (size_t)q * D;
    // fragment-ordered copy (GemmArgs::qf): element c of query q -> ((q / 32 * D/16 + c / 16) * 64 + (q & 31) + 32 * ((c >> 3) & 1)) * 8 + (c & 7)
    unsigned short* outf = (a.qf != nullptr && (D & 15u) == 0u) ? a.qf + ((size_t)(q >> 5) * (D >> 4) * 64u + (q & 31u)) * 8u : nullptr;
    auto frag_at = [](uint32_t c) { return (size_t)(c >> 4) * 512u + (size_t)((c >> 3) & 1u) * 256u + (c & 7u); };
    if (q >= a.nq) {                                     // padding query: admits nothing, matches nothing
        for (uint32_t c = lane; c < D; c += WAVE) { out[c] = 0; if (outf) outf[frag_at(c)] = 0; }
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
        if (outf) outf[frag_at(c)] = b;
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
        // the mirror's measured bounds, where its conversions left them (device words: no host round trip between a conversion and this launch)
        const float max_norm = a.max_bits != nullptr ? __uint_as_float(a.max_bits[0]) : 0.f;
        const float max_row_err = (a.max_bits != nullptr && a.use_measured) ? __uint_as_float(a.max_bits[1]) : 0.f;
        const double vn_d = a.metric == BM_COS ? 1.0 + 1e-6 : (double)max_norm;
        const double u = 0.0078125 * (1.0 + 1.0 / 512.0) + (double)D * 5.97e-8 + 1e-6;          // worst case, relative to ||q|| max||v||
        double dot_err = u * qn_d * vn_d * 1.001;
        if (max_row_err > 0.f) {
            const double measured = sqrt(qe2) * vn_d * (1.0 + 1.0 / 256.0) + qn_d * (double)max_row_err * 1.001 +
                                    3.0 * (double)D * 5.97e-8 * qn_d * vn_d;
            if (measured < dot_err) dot_err = measured;
        }
        float eps;
        if (a.metric == BM_COS) eps = (float)(dot_err + 3e-6);
        else if (a.metric == BM_DOT) eps = (float)(dot_err + 1e-6 * (1.0 + qn_d * vn_d));
        else {   // L2: ||q||^2 + ||v||^2 - 2 q.v carries the factor 2 and the norms' own rounding
            const double ss = (double)c * (double)c + (double)max_norm * (double)max_norm;
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
static hipError_t launch_rq_sample(const GemmArgs& a, hipStream_t st) {
    const uint32_t groups = (a.nqt * 128 + 255) / 256;
    uint32_t pg = 256 / groups;
    if (pg < 1) pg = 1;
    if (pg > a.sample_tiles) pg = a.sample_tiles;
    return launch_rq<D, true, false>(a, groups, pg, st);
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
        case 128: return launch_rq_sample<128>(a, st);
        case 256: return launch_rq_sample<256>(a, st);
        case 384: return launch_rq_sample<384>(a, st);
        case 512: return launch_rq_sample<512>(a, st);
        case 768: return launch_rq_sample<768>(a, st);
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
