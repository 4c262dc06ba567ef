// kernels.h — host-callable launchers for the gfx950 kernels (kernels.hip).
#pragma once
#include <atomic>

#include <hip/hip_ext.h>

#include "common.h"

namespace wax {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: a multi-GPU handle launches the same
// kernel from one worker thread per device, so "configured once per process" leaves every device but the first at the
// 64 KB default and its > 64 KB launches fail. `done` (one per kernel instantiation) holds one bit per device ordinal;
// setting the attribute twice is harmless, so concurrent first launches need no lock.
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << ((unsigned)dev & 63u);
    if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (dev < 64) done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

// Kernel-bound timing ("time_kernels" = 2). hipEventRecord in front of and behind a launch brackets the kernel AND the packets
// around it (both record markers, the chain wait, the dispatch latency: 15-19 us over rocprofv3's duration of the same dispatch when
// scans are chained, which is 0.8 % of a 2.2 ms scan but 6 % of a 0.3 ms one). hipExtLaunchKernel binds a start / stop event pair
// to the DISPATCH: hipEventElapsedTime(start, stop) then runs from the marker the runtime puts in front of the dispatch to the
// dispatch's own completion — no trailing marker, no chain wait; measured 4-8 us above rocprofv3's per-dispatch duration of the same
// launch (profiles/r05/k_event_modes_vs_rocprofv3_same_launches.txt). The caller arms the pair for the next launch of this thread
// (one shot: the first launch_kernel() consumes it, later launches of the same call — the merge kernel behind a scan — are ordinary).
struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
};
inline LaunchTiming& launch_timing() {
    static thread_local LaunchTiming t;
    return t;
}
template <typename F, typename... Args>
inline void launch_kernel(F kernel, const dim3& grid, const dim3& block, uint32_t smem, hipStream_t st, Args... args) {
    LaunchTiming& t = launch_timing();
    if (t.start != nullptr && t.stop != nullptr) {
        const LaunchTiming ev = t;
        t = LaunchTiming{};
        hipExtLaunchKernelGGL(kernel, grid, block, smem, st, ev.start, ev.stop, 0u, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, smem, st, args...);
    }
}

// Arguments of one single-query scan over one shard.
struct ScanArgs {
    const float* store;      // [n_rows][dims] f32 row-major in HBM (a3: MetalVectorEngine vectorsBuffer)
    const float* query;      // [dims] f32 in HBM
    int64_t* partials;       // fused path: [grid][k] per-workgroup sorted keys
    float* dist_out;         // general path: [n_rows] distances (the reference's distances buffer)
    const uint32_t* gate;    // distance-writing launches only: non-null = return at once while *gate == 0 (the short selection answered; see SelectWork)
    uint32_t n_rows;
    uint32_t row_base;       // global row of local row 0 (shard offset)
    uint32_t dims;
    int32_t k;               // min(clamp(topK), n_rows)
    float q_norm;            // ||query||_2 (f64-accumulated on the host, rounded to f32)
    // Fused final merge (small grids: the merge launch costs more than the scan): non-null = the LAST workgroup to arrive
    // (device-scope ticket on *arrive, which it resets to 0) reduces all per-workgroup lists and writes the kpad hits
    // itself — one launch per query instead of two. launch_scan decides (SCAN_FUSE_MERGE_GRID) and reports it.
    wax_hip_hit* merge_out;
    const uint64_t* ids;     // frame ids by local row (merge_out only; may be null)
    uint32_t* arrive;
    int32_t kpad;
    // Host side only (never read on the device): non-null = the query's dims floats in HOST memory, to be passed in the kernel
    // arguments (launch_scan copies them into the launch packet's kernarg block) instead of being read through `query`.
    // Honoured for dims 384 / 768 on the fused path; `query` may then be null.
    const float* query_host;
    // Fused final merge only: non-null = after the kpad hits are written (merge_out is then pinned host memory) the last-arriving
    // workgroup stores done_value to *done_flag (pinned host memory) with system-scope release — the host polls this word instead
    // of waiting on an event recorded behind the kernel (one packet and one API call less per query, and the answer is visible
    // before the kernel has even retired). launch_scan clears it when the launch does not merge in the kernel.
    uint64_t* done_flag;
    uint64_t done_value;
    int32_t plain_loads;     // query-in-arguments kernels: != 0 = ordinary row loads (the store is expected to stay cached between queries), 0 = non-temporal
    int32_t no_kway;         // fused final merge: != 0 = always the wave-list merge (A/B; default 0 = the k-way merge of the list heads for k <= SCAN_KWAY_MAX_K)
};
// kernarg block of scan_kernel_qarg: the scan arguments followed by the query itself (16-byte aligned for float4 loads)
template <int DIMS>
struct alignas(16) ScanArgsQ {
    ScanArgs a;
    alignas(16) float q[DIMS];
};
static_assert(sizeof(ScanArgsQ<768>) <= 4096, "HIP kernel arguments are limited to 4 KB");
// whether launch_scan can take the query through the kernel arguments for this shape (scan_kernel_qarg instantiations)
inline bool scan_query_args_dims(uint32_t dims) { return dims == 384 || dims == 768; }
constexpr int SCAN_FUSE_MERGE_GRID = 160;   // largest grid whose last-arriving workgroup does the final merge (any k <= FUSED_MAX_K)
// k <= SCAN_KWAY_MAX_K: the last arriver merges the lists' HEADS (k rounds of a workgroup-wide minimum; cost independent of the number
// of lists), so every grid the engine launches by default (<= 512 workgroups = two lists per thread) can merge in the scan kernel:
// one launch per query. Measured (profiles/r04/m_*): a blocking call gets 12 - 14 us shorter at every store size (20K rows 38 -> 25 us,
// 100K 55 -> 43, 1M 250 -> 237), while a PIPELINED stream of queries on a large store loses the overlap of the small merge kernel
// with the next scan (the last arriver's tail, ~7 us at 505 workgroups, is serial): 10M x 384 2.162 -> 2.166 ms per query. So the
// larger grids merge in the kernel only up to SCAN_KWAY_MAX_BYTES of rows (a scan of <= ~0.3 ms, where 12 us is >= 4 %).
constexpr int SCAN_KWAY_MAX_K = 64;
constexpr int SCAN_KWAY_MERGE_GRID = 512;
constexpr uint64_t SCAN_KWAY_MAX_BYTES = 2ull << 30;
// will a scan of `grid` workgroups over n_rows x dims floats for top-`k` merge in its own last-arriving workgroup? (kway = "merge_kway")
inline bool scan_merges_in_kernel(int grid, int k, bool kway, uint32_t n_rows, uint32_t dims) {
    return grid <= SCAN_FUSE_MERGE_GRID || (kway && k <= SCAN_KWAY_MAX_K && grid <= SCAN_KWAY_MERGE_GRID &&
                                            (uint64_t)n_rows * dims * sizeof(float) <= SCAN_KWAY_MAX_BYTES);
}

struct ScanVariantInfo {
    int unroll;          // row groups in flight per wave iteration
    int nt;              // 1 = non-temporal (streaming) loads
    int group;           // lanes cooperating on one row
    int rows_per_chunk;  // rows one wave consumes per loop iteration
    int specialised;     // 1 = compile-time dims kernel, 0 = generic any-dims kernel
};

// Number of scan kernel variants available for (dims, metric); variant 0 is the default.
int scan_variant_count(uint32_t dims);
bool scan_variant_info(uint32_t dims, int variant, ScanVariantInfo* out);

// Launch the fused scan+select (write_dist=false: per-workgroup top-k keys into
// args.partials, *out_grid workgroups) or the distance-only scan (write_dist=true).
// cap: 128 (k <= 64) or 256 (k <= 192). grid_cap: max workgroups (0 = default).
// *out_merged (may be null): true = the kernel wrote args.merge_out itself (args.merge_out / arrive were set and the grid is
// small enough); false = the caller launches launch_merge_keys on args.partials as usual.
hipError_t launch_scan(const ScanArgs& args, int metric, int variant, int cap, bool write_dist, int grid_cap,
                       hipStream_t stream, int* out_grid, bool* out_merged = nullptr);
int scan_grid_for(uint32_t n_rows, uint32_t dims, int variant, int grid_cap);

// ---- multiscan.hip: the exact scan for a GROUP of queries in one pass over the store (fallback of the batched path) ----
struct ScanMultiArgs {
    const float* store;       // [n_rows][dims] f32
    const float* queries;     // query block in HBM, row-major x dims
    const uint32_t* qlist;    // [nq] rows of `queries` this launch answers
    const float* q_norm;      // [nq] exact ||q|| of those queries (by slot, not by query number)
    int64_t* partials;        // [nq][grid][k] per-(query, workgroup) sorted keys
    uint32_t n_rows, row_base, dims, nq;
    int32_t k;
};
bool scan_multi_dims(uint32_t dims);                    // dims the kernel is specialised for (= launch_scan's table)
uint32_t scan_multi_group(uint32_t dims, int k);        // queries per launch (0: not served — k > 192, other dims, lists too large for LDS)
int scan_multi_grid(uint32_t n_rows, uint32_t dims, int grid_cap);   // workgroups launch_scan_multi will use
hipError_t launch_scan_multi(const ScanMultiArgs& a, int metric, int grid_cap, hipStream_t stream, int* out_grid);
// Per query b < nq: the n_lists lists of k keys at d_in + b * n_lists * k -> the k smallest as hits (frame ids attached) in
// row (d_qlist ? d_qlist[b] : b) of d_out_base, rows out_stride hits wide and padded to it.
hipError_t launch_merge_keys_multi(const int64_t* d_in, uint32_t n_lists, int k, const uint64_t* d_ids, uint32_t row_base,
                                   uint32_t n_rows, wax_hip_hit* d_out_base, uint32_t out_stride, const uint32_t* d_qlist,
                                   uint32_t nq, hipStream_t stream);

// n_in sorted-or-not keys -> the k smallest, ascending, as hits (frame id looked up in d_ids
// by local row = key_row - row_base; d_ids may be null => frame_id = global row). kpad >= k
// slots are written (tail padded). cap as above. gate (may be null = run): the launch returns at once while *gate == 0 (a short
// selection in front of it answered: launch_select_short).
hipError_t launch_merge_keys(const int64_t* d_in, uint32_t n_in, int k, int kpad, const uint64_t* d_ids,
                             uint32_t row_base, uint32_t n_rows, wax_hip_hit* d_out, int cap, hipStream_t stream,
                             const uint32_t* gate = nullptr);
// gathered shard hits (n <= 16384) -> k smallest ascending (k <= 192)
hipError_t launch_merge_hits(const wax_hip_hit* d_in, uint32_t n, int k, wax_hip_hit* d_out, hipStream_t stream);
// Batched form: d_in = [n_shards][nq][kin] hits (an all-gather of per-shard batch results), d_out = [nq][out_stride]
// (k merged hits per row, the rest padded; out_stride = 0 means k).
hipError_t launch_merge_batch_hits(const wax_hip_hit* d_in, uint32_t n_shards, uint32_t nq, uint32_t kin, int k,
                                   wax_hip_hit* d_out, hipStream_t stream, uint32_t out_stride = 0);

// General (any k <= 10000) selection over a distance buffer: exact k-th key by 8-pass radix
// select on the 64-bit key, compaction, rank sort, id lookup. Work buffers are caller-owned.
struct SelectWork {
    uint32_t* hist;      // [256 bins + 1 arrival ticket], zero between launches
    uint64_t* state;     // [4]: prefix / threshold, rank left, resolved flag; behind them [4] more whose first 16 bytes are `flags`
    uint32_t* counter;   // [1]
    int64_t* keys_a;     // [kmax]
    int64_t* keys_b;     // [kmax]
    uint32_t* flags;     // [4] (inside `state`): [0] gate — 1 = the short selection could not certify its answer, the long path runs;
                         //     [1] short selections that failed, [2] short selections run (both since the workspace was allocated)
};
// `gate` (may be null = run): device word; every kernel of the chain returns at once while *gate == 0 (the short selection answered).
hipError_t launch_select_general(const float* d_dist, uint32_t n_rows, uint32_t row_base, int k, int kpad,
                                 const uint64_t* d_ids, const SelectWork& w, wax_hip_hit* d_out,
                                 hipStream_t stream, const uint32_t* gate = nullptr);
// Short selection over the fused scan's per-workgroup lists (`lists` x `per_list` keys, each ascending, KEY_PAD-padded: every
// workgroup's `per_list` best). ONE workgroup sorts the lists' first few entries in LDS to get a tight upper bound of the k-th key,
// gathers the lists' prefixes below it, sorts those and writes the k best as hits. per_list >= k (the fused path's final merge for
// 64 < k <= 192): always exact. per_list < k (k > FUSED_MAX_K): exact unless some workgroup dropped one of the answer — certified: a
// workgroup whose list is full dropped only keys above its last entry, so every full list's last entry must be >= the k-th selected
// key. d_flags[0] = 0 and hits written, or d_flags[0] = 1 and nothing written (certificate failed, or the prefixes overflowed the LDS
// buffer): the caller's gated launches behind it answer. d_flags[1] counts failures, d_flags[2] launches.
bool select_short_viable(int k, int lists, int per_list);
hipError_t launch_select_short(const int64_t* d_cand, uint32_t lists, uint32_t per_list, int k, int kpad, const uint64_t* d_ids,
                               uint32_t row_base, uint32_t n_rows, uint32_t* d_flags, wax_hip_hit* d_out, hipStream_t stream);
hipError_t alloc_select_work(SelectWork* w);   // on the current device; synchronous (hipMemset of the histogram)
void free_select_work(SelectWork* w);

// Streaming-read microbenchmark: sums every float4 of [bytes] (16-B multiple), one partial per workgroup.
hipError_t launch_stream_read(const float* d_src, uint64_t bytes, int nt, int grid, float* d_sink,
                              hipStream_t stream);

// Order-preserving in-place removal of one row (MetalVectorEngine.remove's memmove, :431-438),
// done through a bounce buffer in chunks; also shifts the id table.
hipError_t device_shift_down(void* base, uint64_t dst_off, uint64_t src_off, uint64_t bytes, void* bounce,
                             uint64_t bounce_bytes, hipStream_t stream);


// ---- batched queries (batch.hip): bf16 MFMA GEMM + select + exact f32 re-score + certificate ----
// Per-query append counters of the LDS-tiled GEMM are device-scope atomics from every CU; packed, 1024 of them share
// 32 cache lines and serialise there (a 16 K-row slab at Q = 1024, D = 768: 161 us). One counter per line spreads them
// over the memory channels.
constexpr uint32_t CAND_COUNT_STRIDE = 32;

struct GemmArgs {
    const unsigned short* qb;   // [nq_pad][dims] bf16 queries (zero padded to a multiple of 128)
    // The same queries in MFMA A-FRAGMENT order (register-resident GEMM only; may be null = read qb): for a block of 32 queries and
    // k-step ks, lane l's 16 bytes (query l & 31, elements 16 ks + 8 (l >> 5) .. + 7) sit at ((block * dims/16 + ks) * 64 + l) * 16 —
    // a wave's fragment load is then ONE contiguous 1-KB run instead of 32 rows x 32 bytes (dims*2 bytes apart), and a workgroup's
    // prologue (8 waves x dims/16 loads, every workgroup of the launch the same block of queries) stops thrashing the CU's L1.
    const unsigned short* qf;
    const unsigned short* cb;   // [n_rows][dims] bf16 corpus mirror
    const float* q_n2;          // [nq_pad] ||q||^2 (L2 epilogue)
    const float* v_n2;          // [n_rows] ||v||^2 (L2 epilogue)
    const float* tau;           // [nq_pad] per-query admission threshold (approx distance), +inf at the start
    float* dense;               // non-null (first slab only): store the whole tile [nq_pad][dense_ld] instead of filtering
    uint32_t dense_ld;
    int64_t* cand;              // [nq][cand_cap] appended candidate keys
    uint32_t* cand_count;       // [nq * CAND_COUNT_STRIDE]: one counter per 128-byte line
    uint32_t cand_cap;
    uint32_t row_base;
    uint32_t dims;
    uint32_t n_rows;
    uint32_t slab0;
    uint32_t slab_rows;
    uint32_t nq;
    uint32_t nqt;               // nq_pad / 128
    uint32_t debug;             // "batch_debug": bit 12 no pace gate, bit 14 one wave of workgroup 1 pretends its split-barrier wait timed out (tests)
    uint32_t use_rega;          // != 0: queries padded to 256 rows => register-resident-queries kernel allowed (1 = a workgroup barrier per tile instead of the split one)
    // Register-resident kernel only: survivors go to per-(workgroup, query) segments, no global atomics.
    // Query q's row of `cand` holds [0, seg_base) the best list and, from seg_base, `nseg` segments of `seg_slots`
    // keys; workgroup b of a 256-query group writes its survivors for q into segment b and its (unclamped) count
    // into seg_count[b * nq_pad + q].
    uint32_t* seg_count;
    uint32_t seg_base;
    uint32_t seg_area;          // slots available after seg_base (nseg * seg_slots <= seg_area)
    // One-pass pipeline, sampling launch only (launch_batch_gemm_sample): per-query best similarity of each of
    // `sample_tiles` tiles spread evenly over the slab, tile_max[sample_tiles][nq_pad].
    float* tile_max;
    uint32_t sample_tiles;
    // rq kernel, more than one query group (D = 768): non-null = [256] progress words (zeroed before the launch; one 128-byte line per
    // XCD) for the advisory pace gate that keeps the groups walking the corpus within a few tiles of each other, so that a tile
    // fetched for one group is still in the XCD's L2 when the others read it.
    uint32_t* progress;
    // rq filtering launch (workgroup-barrier form, one query group): non-null = [BATCH_DYN_WORDS] claim counters, zeroed before the launch.
    // Every workgroup walks its fixed share of the slab except the last twelfth; those tiles form a pool the workgroups claim one at a
    // time (a returning add), so that the launch ends within one tile of every workgroup instead of waiting for the slowest XCD's
    // fixed share. Null, or more than one query group (pool tiles would lose the groups' shared L2 reads) = fixed shares throughout.
    uint32_t* dyn;
    // Diagnosis ("batch_prof_ptr"): non-null = device buffer of [grid * 8 waves][RQ_PROF_WORDS] u32 that the PROF instantiation of the
    // filtering launch fills with per-wave phase cycle counts (indices RQP_*). Never set by the product path.
    uint32_t* prof;
};
// Words per wave of GemmArgs::prof and what they hold (shader cycles unless said otherwise).
enum : int {
    RQP_PROLOGUE = 0,       // kernel entry -> first tile hand-over (A fragments, bounds, first DMA requests landed)
    RQP_SELECT = 1,         // fused selection, hot + cold
    RQP_COLD = 2,           // ... of which: tiles with at least one survivor in this wave (the cold path)
    RQP_COLD_N = 3,         // number of such tiles
    RQP_WAIT_ARRIVALS = 4,  // split barrier: waiting for the other waves (or the workgroup barrier of "batch_rega" 1)
    RQP_DMA_ISSUE = 5,      // pace gate + LDS-DMA requests of the next tile
    RQP_KLOOP = 6,          // ds_read_b128 + MFMA
    RQP_DMA_WAIT = 7,       // counted vmcnt wait for this wave's pieces of the next tile
    RQP_LOOP = 8,           // the whole tile loop
    RQP_EPILOGUE = 9,       // after the loop (last selection, counters, final barrier)
    RQP_TILES = 10,         // tiles this workgroup multiplied
    RQP_SURVIVORS = 11,     // survivors this wave stored
    RQP_RT0_LO = 12, RQP_RT0_HI = 13, RQP_RT1_LO = 14, RQP_RT1_HI = 15,   // s_memrealtime (100 MHz) at entry / exit
    RQP_XCC = 16,           // HW_REG_XCC_ID
    RQ_PROF_WORDS = 20
};
// Geometry the register-resident kernel will use for these arguments (false: the LDS-tiled kernel runs instead,
// appending through cand_count).
bool batch_gemm_segments(const GemmArgs& a, int metric, uint32_t* nseg, uint32_t* seg_slots);
struct RescoreArgs {
    const float* store;
    const float* queries;       // [nq][dims] f32
    const float* q_norm;        // [nq]
    const int64_t* cand;        // [nq][cand_cap], first kp entries = the candidates (ascending approx key)
    int64_t* exact;             // [nq][kp]
    uint32_t n_rows, row_base, dims, nq, cand_cap;
    int kp;
    // filtered single-query search (nq = 1, kp = number of listed rows): rows come from `rows` (local indices,
    // ascending) instead of candidate keys, and plain f32 distances go to dist_out[i] instead of keys to `exact`
    const uint32_t* rows;
    float* dist_out;
    const uint32_t* qlist;      // candidate-key mode: non-null = slot s of the launch is query qlist[s] (arrays indexed by query)
    int by_slot;                // 1 (with qlist) = `cand` and `exact` rows are indexed by the launch slot (compact buffers of a
                                // retry round), only queries / q_norm by the query number
};
// Full retry, step 1: the live survivors of each listed query (segment s holds min(count_s, seg_slots) keys; count_s =
// seg_count[s * nq_pad + q], or seg_count[q * count_stride] for the one counted list) packed into dense[slot][0 .. live),
// the rest of the row (dense_stride keys) padded with KEY_PAD; live_out[slot] = live (pinned host memory: the host sizes the
// re-score by the largest).
struct CompactArgs {
    const int64_t* cand; uint32_t cand_cap;
    const uint32_t* seg_count; uint32_t nseg, seg_slots, nq_pad, count_stride;
    const uint32_t* qlist; uint32_t n_slots;
    int64_t* dense; uint32_t dense_stride;
    uint32_t* live_out;
};
hipError_t launch_compact_survivors(const CompactArgs& a, hipStream_t stream);
// Full retry, step 3 (step 2 = launch_rescore over the dense lists, by slot): exact keys of ALL survivors of each listed query
// (`area` keys per row, KEY_PAD = dead) -> per query the k best, ascending, as hits with frame ids, and the certificate
// "every row the filter rejected is provably worse": no segment overflowed, at least k survivors, tau - eps > the exact k-th.
struct FullRetryArgs {
    const int64_t* exact; uint32_t area;                 // [n_slots][area]
    const uint32_t* qlist; uint32_t n_slots;             // launch slot -> query number
    const uint32_t* seg_count; uint32_t nseg, seg_slots, nq_pad, count_stride;
    const float* tau; const float* eps; const uint32_t* overflow;
    const uint64_t* ids; uint32_t n_rows, row_base;
    int k;
    wax_hip_hit* out; uint32_t out_stride;               // rows indexed by query number
    uint32_t* certified;                                 // indexed by query number
};
hipError_t launch_full_retry_select(const FullRetryArgs& a, hipStream_t stream);
hipError_t launch_mirror(const float* src, uint32_t n_rows, uint32_t n_rows_padded, uint32_t dims, int normalize,
                         unsigned short* dst, float* norm2, unsigned int* max_norm_bits, hipStream_t stream);
// The same for `n_listed` rows named by d_rows[] (device), each converted in place (round 6: upserted rows).
hipError_t launch_mirror_rows(const float* src, const uint32_t* d_rows, uint32_t n_listed, uint32_t dims, int normalize,
                              unsigned short* dst, float* norm2, unsigned int* max_norm_bits, hipStream_t stream);
hipError_t launch_batch_gemm(const GemmArgs& a, int metric, hipStream_t stream);

// ---- one-pass batched pipeline (large stores; cosine / dot at the register-resident GEMM dims) ----
// prep -> sampling GEMM -> pick_tau -> ONE filtering GEMM over the whole store -> finish. No host round trip, no slabs.
bool batch_onepass_dims(uint32_t dims, int metric);
// true: the register-resident filtering GEMMs serve (dims, metric) — survivors in per-workgroup segments; false: the
// LDS-tiled kernel does (L2, other multiples of 64) — survivors in one counted list per query
bool batch_onepass_fast(uint32_t dims, int metric);
// rows per GEMM tile of the kernel that serves (dims, metric): 64, 32 for the K-split kernel, 128 for the LDS-tiled kernel
uint32_t batch_tile_rows(uint32_t dims, int metric);
constexpr uint32_t BATCH_MIRROR_SLACK_ROWS = 160;   // rows allocated behind the bf16 mirror: >= the largest GEMM tile (128 rows) + its over-reads
constexpr uint32_t BATCH_GROUP_QUERIES = 256;   // queries per group of workgroups of the register-resident filtering GEMM
bool batch_finish_fused_dims(uint32_t dims);
// Queries f32 [nq][dims] in HBM -> bf16 block (cosine: normalised; rows [nq, nq_pad) zero), exact ||q|| as the
// single-query path computes it (f64 accumulation, the host's summation order), certificate eps, and the per-batch
// state (tau = +inf / -inf for padding, overflow = 0).
struct PrepArgs {
    const float* queries; uint32_t nq, nq_pad, dims; int metric;
    const unsigned int* max_bits;   // device: {max ||v||, max over the mirror's rows of ||x - bf16(x)||} as f32 bits (mirror_kernel; x normalised for cosine)
    int use_measured;               // 0 = ignore the measured rounding error (worst-case bound: "batch_eps_measured" = 0)
    unsigned short* qb; float* q_n2; float* q_norm; float* eps; float* tau; uint32_t* overflow;
    unsigned short* qf;         // the bf16 queries once more in MFMA A-fragment order (GemmArgs::qf; dims % 16 == 0); may be null
    uint32_t* cand_count;       // slab pipeline: per-query append counters to zero (stride CAND_COUNT_STRIDE); may be null
    uint32_t* progress;         // one-pass pipeline: [BATCH_PROGRESS_WORDS] pace-gate words of the filtering GEMM to zero; may be null
    uint32_t* dyn;              // one-pass pipeline: [BATCH_DYN_WORDS] tile-claim counters of the filtering GEMM to zero; may be null
    float* q_norm_host;         // [nq] the exact norms once more, straight into pinned host memory (fallback queries need them there); may be null
};
constexpr uint32_t BATCH_PROGRESS_WORDS = 256;
constexpr uint32_t BATCH_DYN_WORDS = 128;          // 4 query groups x one 128-byte line
hipError_t launch_batch_prep(const PrepArgs& a, hipStream_t stream);
hipError_t launch_batch_gemm_sample(const GemmArgs& a, int metric, hipStream_t stream);
// tau[q] = 1 - (the rank-th largest of the sampled tiles' best similarities), rank <= 12; padding queries keep -inf.
hipError_t launch_pick_tau(const float* tile_max, uint32_t sample_tiles, uint32_t nq, uint32_t nq_pad, uint32_t rank,
                           float* tau, int metric, hipStream_t stream);
// Per query: survivors of the filtering GEMM (per-workgroup segments) -> best kp by approximate key -> exact f32
// re-score with the scan kernel's arithmetic -> top-k hits + exactness certificate. kp <= 192: one fused kernel;
// larger kp (k up to 464): segment select, re-score and finalize as three launches.
struct FinishArgs {
    const int64_t* cand; uint32_t cand_cap;              // [nq][cand_cap]: nseg segments of seg_slots keys from slot 0
    const uint32_t* seg_count; uint32_t nseg, seg_slots, nq_pad;
    uint32_t count_stride;                               // != 0: ONE counted list of seg_slots keys per query, its length at seg_count[q * count_stride]
    const float* tau; const uint32_t* overflow;          // admission threshold the GEMM used; pre-set overflow flags
    const float* store; const float* queries; const float* q_norm; const float* eps;
    const uint64_t* ids; uint32_t n_rows, row_base, dims, nq;
    int kp, k;
    int64_t* sel;                                        // [nq][kp] scratch: selected approximate keys (large-kp path)
    int64_t* exact;                                      // [nq][kp] scratch (large-kp path)
    wax_hip_hit* out; uint32_t out_stride;               // [nq][out_stride], k written per query
    uint32_t* certified;                                 // [nq]
    // large-kp path only: non-null = process queries qlist[0 .. nq); every array above stays indexed by the query's own number.
    // (Round 2's "wide retry" used it; the full retry of round 3 has its own kernels — CompactArgs / FullRetryArgs — and the
    // engine leaves this null.)
    const uint32_t* qlist;
    // fused finish kernel (kp <= 192): non-null = device-side copy of the flags for batch_retry_kernel: 1 certified, 0 = failed with
    // every survivor still in the segments (retryable), 2 = failed and a retry is pointless (something dropped / nothing beyond k')
    uint32_t* cert_dev;
    uint32_t retry_all;                                  // batch_retry_kernel: != 0 = re-score EVERY survivor (A/B: "batch_debug" bit 16); 0 = only those the first finish's k-th cannot exclude
};
// Device-side full retry behind launch_batch_finish (same arguments, cert_dev written by it): per uncertified query ALL survivors are
// re-scored exactly and the k best written; a query it certifies gets certified[q] = 2.
bool batch_retry_dims(uint32_t dims);
hipError_t launch_batch_retry(const FinishArgs& a, int metric, hipStream_t stream);
hipError_t launch_batch_finish(const FinishArgs& a, int metric, hipStream_t stream);
struct TightenArgs {
    int64_t* cand; uint32_t cand_cap; uint32_t* cand_count; int kp; uint32_t nq; float* tau; uint32_t* overflow;
    const float* dense; uint32_t dense_ld, dense_rows, dense_row0;      // first slab: dense score tile
    const uint32_t* seg_count; uint32_t nseg, seg_slots, seg_base, nq_pad;  // nseg > 0: segmented survivors
};
hipError_t launch_tighten(const TightenArgs& a, hipStream_t stream);
hipError_t launch_batch_reset(float* tau, uint32_t* cand_count, uint32_t* overflow, uint32_t nq, uint32_t nq_pad,
                              hipStream_t stream);
hipError_t launch_rescore(const RescoreArgs& a, int metric, hipStream_t stream);
hipError_t launch_finalize_batch(const int64_t* cand, uint32_t cand_cap, const uint32_t* overflow, const int64_t* exact,
                                 int kp, int k, const float* eps,
                                 const uint64_t* ids, uint32_t row_base, uint32_t n_rows, uint32_t nq,
                                 wax_hip_hit* out, uint32_t out_stride, uint32_t* certified, hipStream_t stream);


// ---- filter.hip: allow-list pre-filter on the device (id -> row table in HBM, bitmap, compaction) ----
hipError_t launch_idhash_build(const uint64_t* ids, uint32_t row0, uint32_t n, uint32_t* table, uint64_t slots, hipStream_t st);   // rows [row0, n); row0 = 0 clears the table first
uint32_t filter_bitmap_blocks(uint32_t n_rows);
// probe + per-block popcounts + exclusive scan; *total = number of distinct allowed rows present in the store
hipError_t launch_allow_probe(const uint64_t* d_allow, uint64_t n_allow, const uint64_t* ids, uint32_t n_rows,
                              const uint32_t* table, uint64_t slots, uint32_t* bitmap, uint32_t* block_sum, uint32_t* total,
                              hipStream_t st);
hipError_t launch_allow_emit(const uint32_t* bitmap, uint32_t n_rows, const uint32_t* block_off, const uint64_t* ids,
                             uint32_t* rows_out, uint64_t* ids_out, hipStream_t st);


// ---- rrf.hip: reciprocal-rank fusion of ranked id lists, one workgroup per query ----
hipError_t launch_rrf_fuse(const wax_hip_rrf_lane* lanes, uint32_t n_lanes, uint32_t nq, int32_t k, uint64_t* out_ids,
                           float* out_scores, uint32_t* out_best_rank, uint32_t* out_sources, uint32_t out_stride,
                           uint32_t* out_counts, hipStream_t st);

}  // namespace wax
