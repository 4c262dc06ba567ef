/*
 * wax_hip.h — C ABI of libwaxhip: the MI355X (gfx950 / CDNA4) brute-force vector
 * scan + top-k backend for Wax's `VectorSearchEngine`.
 *
 * This is the drop-in boundary. Every entry point replaces one
 * member of the reference's Swift surface; the reference file:line it stands in
 * for is cited on each declaration (paths relative to the reference checkout).
 * A Swift `actor HIPVectorEngine: VectorSearchEngine` binds these through a
 * module map; Python binds them with ctypes
 * (wax_amd/_abi.py); C++ callers can use include/wax_hip.hpp.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in any signature
 *     (`void* stream` is a hipStream_t passed opaquely, NULL = engine's own)
 *   - return 0 (WAX_HIP_OK) or a negative wax_hip_status; the message is in
 *     wax_hip_last_error() (thread-local)
 *   - the caller owns every in/out array and states its size: every result array comes with an
 *     explicit capacity (entries) and the library never writes past it, whatever the engine's row
 *     count has become by the time the call runs (a concurrent add may grow it between the caller's
 *     sizing and the search). wax_hip_result_capacity(top_k) = clamp(top_k, 1, 10000) entries always
 *     suffice. Buffers returned by wax_hip_serialize() are released with wax_hip_free()
 *   - search* calls are re-entrant (shared lock + per-call scratch slot),
 *     mutations take the exclusive lock — the same reader/writer contract as
 *     the reference's AsyncReadWriteLock (MetalVectorEngine.swift:56-81)
 *   - there is NO CPU fallback in this library: without a gfx950 device every
 *     engine call fails with WAX_HIP_ERR_NO_DEVICE.
 */
#ifndef WAX_HIP_H
#define WAX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAX_HIP_ABI_VERSION 2

/* MetalVectorEngine.maxResults (MetalVectorEngine.swift:18) */
#define WAX_HIP_MAX_RESULTS 10000
/* Constants.maxEmbeddingDimensions (WaxCore/Constants.swift:51) */
#define WAX_HIP_MAX_DIMENSIONS 1000000
/* MetalVectorEngine.initialReserve (MetalVectorEngine.swift:19) */
#define WAX_HIP_INITIAL_RESERVE 64

typedef struct wax_hip_engine wax_hip_engine;

/* Maps 1:1 onto the WaxError cases the Metal engine throws
 * (MetalVectorEngine.swift:154-169, 830-840, 858-860; WaxError.swift:4-18). */
typedef enum wax_hip_status {
    WAX_HIP_OK = 0,
    WAX_HIP_ERR_DIM_MISMATCH = -1,      /* WaxError.encodingError("vector dimension mismatch: expected X, got Y") */
    WAX_HIP_ERR_CAPACITY = -2,          /* WaxError.capacityExceeded(limit:requested:) */
    WAX_HIP_ERR_NO_DEVICE = -3,         /* WaxError.invalidToc("... device not available") */
    WAX_HIP_ERR_ALLOC = -4,             /* WaxError.invalidToc("Failed to allocate ...") */
    WAX_HIP_ERR_BAD_SEGMENT = -5,       /* WaxError.invalidToc(<segment decode reason>) */
    WAX_HIP_ERR_METRIC_UNSUPPORTED = -6,
    WAX_HIP_ERR_INVALID_ARGUMENT = -7,  /* WaxError.encodingError / invalidToc("dimensions must be > 0") */
    WAX_HIP_ERR_INTERNAL = -8
} wax_hip_status;

/* VecSimilarity raw values (WaxCore/FileFormat/MV2SEnums.swift:34-38) ==
 * VectorMetric cases (VectorMetric.swift:5-8). */
typedef enum wax_hip_metric {
    WAX_HIP_METRIC_COSINE = 0,
    WAX_HIP_METRIC_DOT = 1,
    WAX_HIP_METRIC_L2 = 2
} wax_hip_metric;

/* One ranked candidate as it lives in HBM and travels over RCCL between shards.
 * key = (int64)ordered(distance_f32) << 32 | global_row_u32, where ordered()
 * is the sign-magnitude -> two's-complement monotone map, so that a SIGNED
 * 64-bit ascending sort is exactly "(distance asc, row asc)" — the tie rule
 * SURVEY.md §8c adopts from MetalVectorEngine.topK's earlier-index-wins
 * boundary (MetalVectorEngine.swift:671). Replaces the reference's
 * TopKEntry{float distance; uint index} (TopKReduction.metal:15-18,
 * MetalVectorEngine.swift:26-29) plus the frameIds[index] lookup (:601). */
typedef struct wax_hip_hit {
    int64_t key;
    uint64_t frame_id;
} wax_hip_hit;

/* Counters; superset of MetalVectorEngine.BufferPoolStats
 * (MetalVectorEngine.swift:43-46, 119-121). */
typedef struct wax_hip_stats_t {
    uint64_t searches;             /* single-query scans issued */
    uint64_t rows_scanned;         /* sum of rows visited by those scans */
    uint64_t bytes_scanned;        /* algorithmic bytes: rows * dims * 4 */
    uint64_t transient_allocations;/* scratch slots created (BufferPoolStats.transientAllocations) */
    uint64_t reuse_count;          /* scratch slot reuses (BufferPoolStats.reuseCount) */
    uint64_t reserved_rows;        /* current capacity in rows (reservedCapacity) */
    double   last_scan_kernel_ms;  /* HIP-event time of the most recent timed scan kernel (0 if timing off) */
    double   scan_kernel_ms_total; /* sum of HIP-event scan-kernel times since "reset_stats" ("time_kernels"=1) */
    uint64_t scan_kernels_timed;   /* number of scan-kernel launches in that sum */
    double   batch_gemm_ms_total;  /* sum of HIP-event times of the batched path's filtering GEMM launches ("time_kernels"=1) */
    uint64_t batch_gemms_timed;    /* number of GEMM launches in that sum */
    uint64_t batch_gemm_rows;      /* corpus rows those launches covered, summed (algorithmic bytes = rows * dims * 2, bf16 mirror) */
    uint64_t batch_gemm_queries;   /* queries those launches served, summed (flops = 2 * queries * rows-per-launch * dims) */
} wax_hip_stats_t;

/* ---- availability ------------------------------------------------------- */

/* MetalVectorEngine.isAvailable (MetalVectorEngine.swift:144-146): 1 iff a gfx950 device is visible. */
int wax_hip_available(void);
int wax_hip_device_count(void);
uint32_t wax_hip_abi_version(void);
/* thread-local, never NULL */
const char* wax_hip_last_error(void);

/* ---- lifecycle ---------------------------------------------------------- */

/* MetalVectorEngine.init(metric:dimensions:) (MetalVectorEngine.swift:153-274).
 * metric: wax_hip_metric (the Metal engine accepts only cosine, :163-165; this
 * backend also implements dot and l2 with USearch's distance conventions,
 * VectorMetric.swift:21-43). device_id: HIP ordinal, -1 = current device. */
int wax_hip_engine_create(uint8_t metric, uint32_t dims, int device_id, wax_hip_engine** out);
/* The same engine over several GPUs of one node, in ONE process (the reference has no multi-device path):
 * the handle owns one single-device engine ("shard") per listed HIP ordinal (duplicates allowed: two shards on one GPU, for
 * tests). Rows are kept in contiguous blocks — shard g holds global rows [base_g, base_g + count_g), base_g = the rows before
 * it — so the concatenation of the shards is the insertion order of ONE engine: every entry point of this header that takes
 * a handle works on it with identical results (same tie rule, same serialize bytes). A shard is full at ceil(N / n_devices)
 * rows after wax_hip_reserve(N) (or the first batch); when every shard is full the block size doubles (peer-copy rebalance).
 * search: per shard on its own stream, query upload + fused scan + per-shard top-k, then the k hits (16 k bytes per shard)
 * are peer-copied to the first device and merged there by key (tuning "exchange" = 1: one ncclAllGather per query on a
 * single-process RCCL communicator instead). Batched search: every shard answers the batch on its rows through the
 * single-device submit / collect pair (one host thread drives all shards: nothing blocks between them, no thread and no
 * allocation per call), the first device merges per query; the device-resident forms (wax_hip_search_batch_hits_device,
 * _submit_device / _collect_device) work on the handle too — queries and hits then live in the FIRST device's HBM, the query
 * block is peer-copied to the other shards and only nq x k hits per shard come back. An unavailable ordinal fails with
 * WAX_HIP_ERR_NO_DEVICE. Single-device-only entry points (wax_hip_set_row_base, wax_hip_search_shard_device, the two
 * wax_hip_time_* probes) return WAX_HIP_ERR_INVALID_ARGUMENT on such a handle. */
int wax_hip_engine_create_sharded(uint8_t metric, uint32_t dims, const int* device_ids, int n_devices, wax_hip_engine** out);
/* 1 for a single-device engine; shard -> (device ordinal, first global row, rows). */
int wax_hip_shard_count(const wax_hip_engine* e);
int wax_hip_shard_info(const wax_hip_engine* e, int shard, int* out_device, uint64_t* out_row_base, uint64_t* out_rows);
void wax_hip_engine_destroy(wax_hip_engine* e);

/* VectorSearchEngine.dimensions (VectorSearchEngine.swift:11) */
uint32_t wax_hip_dimensions(const wax_hip_engine* e);
/* vectorCount (MetalVectorEngine.swift:51) */
uint64_t wax_hip_count(const wax_hip_engine* e);
uint8_t wax_hip_metric_of(const wax_hip_engine* e);
int wax_hip_device_of(const wax_hip_engine* e);

/* ---- store mutation (exclusive lock) ------------------------------------ */

/* add(frameId:vector:) — upsert by frame id (MetalVectorEngine.swift:330-357). */
int wax_hip_add(wax_hip_engine* e, uint64_t frame_id, const float* vector, uint32_t dims);
/* addBatch(frameIds:vectors:) (MetalVectorEngine.swift:359-402); rows is row-major n x dims.
 * addBatchStreaming (:404-421) is the same call made in chunks by the caller. */
int wax_hip_add_batch(wax_hip_engine* e, const uint64_t* frame_ids, const float* rows, uint64_t n, uint32_t dims);
/* Same, but `rows` is a DEVICE pointer on the engine's device (embeddings that
 * were produced in HBM never bounce through the host). frame_ids stays a host
 * pointer. All ids must be new (append-only fast path); WAX_HIP_ERR_INVALID_ARGUMENT otherwise. */
int wax_hip_add_batch_device(wax_hip_engine* e, const uint64_t* frame_ids, const float* d_rows, uint64_t n, uint32_t dims);
/* Pending-embedding replay (UnifiedSearchEngineCache.applyPendingEmbeddingsIfNeeded, UnifiedSearchEngineCache.swift:252-283;
 * MetalVectorEngine.load, MetalVectorEngine.swift:318-328): `payloads` is `len` bytes of WAL putEmbedding entry
 * payloads laid back to back exactly as WALEntryCodec.encode writes them (WALEntryCodec.swift:39-54):
 * u8 opcode 0x04, u64 frameId LE, u32 dimension LE, dimension x f32 LE. The stream is validated as a whole
 * (decode rules of WALEntryCodec.swift:104-129: unknown opcode, dimension > 1 000 000, truncated record ->
 * WAX_HIP_ERR_BAD_SEGMENT; dimension != engine dims -> WAX_HIP_ERR_DIM_MISMATCH) and then applied in order
 * as ONE addBatch (later records of the same frame overwrite earlier ones), so a bad stream changes nothing.
 * *out_applied (may be NULL) receives the number of records. */
int wax_hip_apply_put_embeddings(wax_hip_engine* e, const uint8_t* payloads, uint64_t len, uint64_t* out_applied);
/* remove(frameId:) — order-preserving delete, absent id is a no-op (MetalVectorEngine.swift:423-444). */
int wax_hip_remove(wax_hip_engine* e, uint64_t frame_id);
/* reserveIfNeeded(for:) (MetalVectorEngine.swift:857-871): capacity doubling from 64, cap UInt32.max rows. */
int wax_hip_reserve(wax_hip_engine* e, uint64_t rows);

/* ---- search (shared lock, re-entrant) ----------------------------------- */

/* clampTopK (MetalVectorEngine.swift:842-846): the number of entries a result array needs so that no
 * result is ever truncated, independent of the engine's current row count. */
uint32_t wax_hip_result_capacity(int32_t top_k);

/* VectorSearchEngine.search(vector:topK:) (VectorSearchEngine.swift:13;
 * MetalVectorEngine.swift:446-627). Blocking. out_ids/out_scores hold out_capacity
 * entries; *out_count receives how many were written = min(clamp(top_k,1,10000), count,
 * out_capacity) minus dropped non-finite entries (best first: descending score == ascending
 * distance, ties by ascending row; a capacity below the result count keeps the best ones).
 * Empty engine => OK with *out_count = 0 (:448). */
int wax_hip_search(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k,
                   uint64_t* out_ids, float* out_scores, uint32_t out_capacity, uint32_t* out_count);

/* Pipelined form of the same call: submit enqueues H2D + kernels + D2H on one
 * of the engine's scratch slots and returns immediately with a ticket;
 * collect blocks on that slot and fills the outputs exactly like
 * wax_hip_search. Tickets must be collected exactly once, in any order (also from another
 * thread). A thread that already holds tickets never waits for a scratch slot (it gets a fresh one,
 * up to 256 outstanding), so pipelining callers cannot deadlock each other. A ticket holds the engine's
 * shared lock until it is collected: a mutation (add / remove / reserve / deserialize / set_row_base)
 * called by a thread that still holds uncollected tickets would wait for itself forever, so it fails
 * with WAX_HIP_ERR_INVALID_ARGUMENT ("collect outstanding search tickets first") instead. */
int wax_hip_search_submit(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint64_t* out_ticket);
int wax_hip_search_collect(wax_hip_engine* e, uint64_t ticket,
                           uint64_t* out_ids, float* out_scores, uint32_t out_capacity, uint32_t* out_count);

/* nq queries, row-major nq x dims. out_ids/out_scores are nq rows of out_stride entries (query q's
 * results start at q * out_stride; at most out_stride are written per query); out_counts[nq]. */
int wax_hip_search_batch(wax_hip_engine* e, const float* queries, uint32_t nq, uint32_t dims, int32_t top_k,
                         uint64_t* out_ids, float* out_scores, uint32_t out_stride, uint32_t* out_counts);

/* Same search, but the raw ranked candidates are returned: out_hits is nq rows of out_stride wax_hip_hit
 * (ascending key, every row padded to out_stride with key = INT64_MAX). This is what a row-sharded deployment exchanges between ranks and
 * merges by key (exact tie order by GLOBAL row, see wax_hip_set_row_base). Large batches run Q x D^T as a
 * bf16 MFMA GEMM with an exact f32 re-score; results are identical to nq calls of wax_hip_search. */
int wax_hip_search_batch_hits(wax_hip_engine* e, const float* queries, uint32_t nq, uint32_t dims, int32_t top_k,
                              wax_hip_hit* out_hits, uint32_t out_stride, uint32_t* out_counts);

/* Device-resident form of the same search: d_queries is nq x dims f32 in HBM on the engine's device (embeddings that
 * were produced on the GPU never bounce through the host), d_out_hits receives nq rows of out_stride hits in HBM
 * (ascending key, padded). Blocking. `stream` (a hipStream_t, NULL = the null stream) is the stream whose earlier
 * work produced d_queries; the library orders its own work behind it, and on return all results are complete.
 * Inside, the batch runs without a host round trip (certificate flags come back once, at the end); uncertified
 * queries are re-run on the exact path in place. This is the entry point the batched BASELINE configs are timed on
 * (inputs resident in HBM), and what a row-sharded deployment feeds to its RCCL all-gather. */
int wax_hip_search_batch_hits_device(wax_hip_engine* e, const float* d_queries, uint32_t nq, uint32_t dims, int32_t top_k,
                                     wax_hip_hit* d_out_hits, uint32_t out_stride, void* stream);

/* Pipelined form (the batched counterpart of wax_hip_search_submit / _collect, i.e. of the command-buffer overlap the
 * reference gets from MTLCommandQueue): submit enqueues the whole batch on one of the engine's batch workspaces, ordered
 * behind `stream`, and returns at once with a ticket; collect waits for it, re-runs uncertified queries on the exact path
 * in place, and reports how many that were (out_fallbacks may be NULL). With two or more tickets in flight the next batch's
 * launches and the host's wake-up hide under the running batch. d_queries / d_out_hits must stay valid and untouched until
 * collect. At most "batch_workspaces" (default 4) tickets per engine; a ticket holds the engine's read lock like a
 * single-query ticket (writers are refused with WAX_HIP_ERR_INVALID_ARGUMENT while the calling thread holds one). */
int wax_hip_search_batch_submit_device(wax_hip_engine* e, const float* d_queries, uint32_t nq, uint32_t dims, int32_t top_k,
                                       wax_hip_hit* d_out_hits, uint32_t out_stride, void* stream, uint64_t* out_ticket);
int wax_hip_search_batch_collect_device(wax_hip_engine* e, uint64_t ticket, uint32_t* out_fallbacks);

/* ---- rank fusion on the device ----
 *
 * Weighted reciprocal-rank fusion of ranked frame-id lists for nq queries at once: HybridSearch.rrfFusion(lists:k:)
 * (HybridSearch.swift:25-52) == UnifiedSearch.rrfFusionResults (UnifiedSearch.swift:590-699) without the diagnostics.
 * Lanes are taken in order; a lane with weight <= 0 is skipped; the entry at 0-based position p of a lane adds
 * weight / Float(max(0, k) + p + 1) to its frame's f32 score (lane order, then position order: scores are bit-identical
 * to the Swift loop); bestRank = the smallest p + 1 over the lanes that name the frame; sources = bit l set when lane l
 * names it. Per query the output row holds every frame seen, best first by (score desc, bestRank asc, frameId asc),
 * truncated / padded (frame id UInt64.max, score 0) to out_stride; d_out_counts[q] = real entries.
 *
 * Lane l of query q: entry p is d_ids[(q * stride + p) * pitch] (pitch 1 = a plain id array; pitch 2 with d_ids pointing at
 * the frame_id field = a wax_hip_hit array straight out of wax_hip_search_batch_hits_device), d_counts[q] entries (NULL:
 * stride entries); entries equal to UInt64.max (the padding of a hit list) are skipped. All pointers are device memory on
 * the current device; the call enqueues on `stream` and returns. Limits: n_lanes <= 8, sum of the strides <= 4096,
 * k <= 2^30. d_out_best_rank / d_out_sources / d_out_counts may be NULL. */
#define WAX_HIP_RRF_MAX_LANES 8
#define WAX_HIP_RRF_MAX_ENTRIES 4096
typedef struct wax_hip_rrf_lane {
    const uint64_t* d_ids;
    const uint32_t* d_counts;
    uint32_t stride;
    uint32_t pitch;
    float weight;
} wax_hip_rrf_lane;
int wax_hip_rrf_fuse_batch_device(const wax_hip_rrf_lane* lanes, uint32_t n_lanes, uint32_t nq, int32_t k,
                                  uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_best_rank,
                                  uint32_t* d_out_sources, uint32_t out_stride, uint32_t* d_out_counts, void* stream);
/* One query from host memory (what UnifiedSearch holds when the text lane comes from FTS5): uploads the lists, fuses on
 * `device_id` (-1 = current), downloads. lists[l] has list_counts[l] ids; at most out_capacity results are written. */
int wax_hip_rrf_fuse(const float* weights, const uint64_t* const* lists, const uint32_t* list_counts, uint32_t n_lists,
                     int32_t k, int device_id, uint64_t* out_ids, float* out_scores, uint32_t* out_best_rank,
                     uint32_t* out_sources, uint32_t out_capacity, uint32_t* out_count);

/* ---- sharded search: per-shard top-k left in HBM for the RCCL exchange ---- */

/* Declares that this engine holds rows [row_base, row_base+count) of a corpus
 * that is row-sharded over several engines/GPUs. row_base is
 * folded into wax_hip_hit.key so ties break by GLOBAL row on every shard count. */
int wax_hip_set_row_base(wax_hip_engine* e, uint64_t row_base);
/* Scan this shard for one query and write exactly kpad = clamp(top_k) hits,
 * sorted ascending by key, to the DEVICE buffer d_out_hits (entries past the
 * shard's own count are padded with key = INT64_MAX, frame_id = UINT64_MAX).
 * Work is enqueued on `stream` (a hipStream_t; NULL = the null stream) and NOT
 * synchronised: the caller all-gathers d_out_hits over RCCL on the same stream
 * order and then calls wax_hip_merge_hits_device. top_k must be <= 192 here. */
int wax_hip_search_shard_device(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k,
                                wax_hip_hit* d_out_hits, void* stream);
/* Stateless G*k -> k merge of gathered shard results on the current device:
 * d_out receives the k smallest keys of d_in[0..n), ascending. n <= 16384. */
int wax_hip_merge_hits_device(const wax_hip_hit* d_in, uint32_t n, uint32_t k, wax_hip_hit* d_out, void* stream);
/* Batched form for the sharded batched path (BASELINE config 5): d_in = [n_shards][nq][k_in] hits (the all-gather
 * of every shard's wax_hip_search_batch_hits result, copied to the device), d_out = [nq][k]: per query the k
 * smallest keys, ascending. One workgroup per query. n_shards * k_in <= 16384, k <= 192. */
int wax_hip_merge_batch_hits_device(const wax_hip_hit* d_in, uint32_t n_shards, uint32_t nq, uint32_t k_in, uint32_t k,
                                    wax_hip_hit* d_out, void* stream);
/* Host-side tail of search (MetalVectorEngine.swift:592-611 + VectorMetric.swift:32-43):
 * drop padded / non-finite entries, distance -> score, emit (frameId, score); out arrays hold n entries. */
int wax_hip_hits_to_results(uint8_t metric, const wax_hip_hit* hits, uint32_t n,
                            uint64_t* out_ids, float* out_scores, uint32_t* out_count);

/* ---- filtered search (the vector lane's candidate filters, applied on the device) -------- *
 * UnifiedSearch applies FrameFilter.frameIds (an allow-list) and SearchRequest.minScore to the vector lane's
 * candidates AFTER the engine returned them (passesFrameFilter, UnifiedSearch.swift:1241-1258), so it asks the engine
 * for candidateLimit = max(topK, min(3 topK, 1000)) results (:1195-1200) and can still come back short when the
 * allowed frames rank deeper (the reference test only covers a 4-document corpus,
 * UnifiedSearchTests.swift:133-158). Here the allow-list is a PRE-filter: only the allowed rows are scored
 * (cost ~ allowed rows, not corpus rows) and the top_k of THEM comes back, then results with score < min_score are
 * dropped. has_allow = 0: no allow-list (allow_frame_ids ignored); has_allow = 1 with n_allow = 0: nothing is
 * allowed. Ids not in the engine are ignored. has_min_score = 0: no score cut. Distances are bit-identical to
 * wax_hip_search's for the same rows. */
int wax_hip_search_filtered(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k,
                            int has_allow, const uint64_t* allow_frame_ids, uint64_t n_allow,
                            int has_min_score, float min_score,
                            uint64_t* out_ids, float* out_scores, uint32_t out_capacity, uint32_t* out_count);

/* ---- persistence: "MV2V" vec segment, encoding 2 ------------------------- */

/* serialize() (MetalVectorEngine.swift:682-714): byte-identical layout
 * "MV2V" u16 ver=1 u8 enc=2 u8 similarity u32 dim u64 count u64 vectorBytes 8x0
 * | count*dim f32 LE | u64 idBytes | count u64 LE. */
int wax_hip_serialize(wax_hip_engine* e, uint8_t** out_bytes, size_t* out_len);
/* deserialize(_:) (MetalVectorEngine.swift:716-815), same validation order and reasons. */
int wax_hip_deserialize(wax_hip_engine* e, const uint8_t* bytes, size_t len);
void wax_hip_free(void* p);

/* ---- observability / tuning --------------------------------------------- */

int wax_hip_stats(wax_hip_engine* e, wax_hip_stats_t* out);
/* Tunables (all optional). No key can change an answer: every setting selects between paths that return the same hits.
 *
 * single-query scan
 *   "grid_blocks" (0 = auto), "variant" (scan kernel variant index, -1 = auto), "stream_nt", "force_general" (1 = the distance-buffer +
 *   radix-select path even for small k), "slots" (scratch-slot pool size), "streams" (1..4 in-order streams the slots rotate over),
 *   "time_kernels" (wax_hip_stats' kernel times: 1 = HIP events recorded in front of and behind the scan / filtering-GEMM launch on its own
 *   stream — the interval holds the kernel and the packets around it; 2 = the event pair bound to the dispatch itself (hipExtLaunchKernel):
 *   no trailing marker and no chain wait inside the interval; 4-8 us above rocprofv3's duration of the same dispatch), "reset_stats" (any value: zero the counters),
 *   "fuse_merge" (1 (default) = on grids of at most 160 workgroups the scan kernel's last-arriving workgroup does the final merge),
 *   "merge_kway" (1 (default) = top_k <= 64: that workgroup merges the per-workgroup lists by their heads, which lets every default
 *   grid of a store of up to 2 GiB of rows finish in ONE launch; 0 = the round-3 rule),
 *   "merge_overlap_mb" (default 400: a scan submitted while other single-query tickets of the engine are out — a stream of scans — over a
 *   store of at least this many MB leaves the merge to a second launch, which overlaps the next scan; 0 = never; a query submitted alone
 *   always keeps the single launch), "overlap_scans" (read-only: scans that took that form),
 *   "query_args" (dims 384 / 768: 1 (default) = a scan that merges in its own kernel takes the query in its kernel arguments; 2 = every
 *   store; 0 = never), "scan_plain_mb" (such scans read stores of at most this many MB with ordinary instead of non-temporal loads, default 32),
 *   "done_flag" (1 (default) = such a scan publishes a completion word in coherent pinned memory behind its hits and collect polls it
 *   instead of an event; never while "time_kernels" != 0),
 *   "scan_chain" (pipelined scans of different streams: 1 = chained through an event so that a per-launch duration is one scan alone;
 *   0 = free to overlap; -1 (default) = chained exactly while "time_kernels" != 0), "share_timing" (1 = a chained scan reuses its
 *   predecessor's end event as its start event),
 *   "select_short" (1 (default): top_k > 192 = the fused scan leaves every workgroup's 192 best, ONE workgroup selects the top_k among
 *   them in LDS and certifies that no workgroup dropped one of the answer — otherwise, on the device, the distance pass + radix
 *   selection behind it answers; 64 < top_k <= 192 on grids that do not merge in the scan kernel = the same workgroup is the final
 *   merge (nothing can have been dropped: no certificate), the wave-list merge stays behind it, gated; 0 = the long path / the wave-list
 *   merge at once; 2 = like 1 with 192-entry lists whatever top_k (1 keeps 64-entry lists while top_k <= 8 per workgroup); "force_general"
 *   implies 0 for top_k > 192), "short_selects" / "short_select_failures" (read-only: short selections
 *   enqueued / that left the answer to the launches behind them),
 *   "filter_device_min" (wax_hip_search_filtered: allow-lists at least this long are resolved by the id -> row table in HBM, default 4096; -1 = never).
 * batched queries (bf16 MFMA GEMM + fused selection + exact re-score; exact answers whatever the setting)
 *   "batch_mode" (0 = never use the MFMA path), "batch_min" (smallest batch that may use it, default 1; below 16 queries a cost model
 *   picks between one GEMM pass over the bf16 mirror and nq f32 scans), "batch_workspaces" (concurrent batched searches per engine, default 4),
 *   "batch_qfrag" (1 (default) = the register-resident GEMM loads its query fragments from a fragment-ordered copy the prep kernel writes —
 *   coalesced 1-KB runs; 0 = from the row-major block),
 *   "batch_rega" (1 (default) = the register-resident-queries GEMM with a workgroup barrier per tile; 5 = the same with the split
 *   tile barrier (round 5's default); 0 = the LDS-tiled GEMM only), "batch_dyn_tail" (1 (default) = with up to 256 queries per pass the workgroups of the
 *   filtering GEMM claim the last twelfth of the store's tiles from a pool, so the launch ends within one tile of every workgroup; 0 = fixed shares), "batch_onepass" (0 = always the slab pipeline), "batch_onepass_tiles" (smallest store, in
 *   units of 64 rows — 32 at D = 768 —, that the one-pass pipeline takes; default 64 = 4 096 rows (1024 before round 6), at least 32), "batch_survivors" (floor of the expected survivors per query as a
 *   multiple of k', default 3), "batch_sample_div" (1 / this of the tiles are sampled for the thresholds, default 32),
 *   "batch_retry" (an uncertified query gets ALL of its survivors re-scored before the exact path: 1 (default) = by a device-side
 *   kernel while "retry_hint" > 0 — armed for 16 batches by any such query, settable 0..1024 — and from the host for what is left;
 *   2 = from the host only; 0 = never), "batch_kp_fused" (1 (default) = top_k 81 .. 128 re-scores k' = 192 candidates in the fused finish kernel),
 *   "batch_multi" (1 (default) = uncertified queries share passes over the f32 store, up to 16 per pass, bit-identical to the
 *   single-query kernel), "batch_eps_measured" (1 (default) = the certificate bound uses the measured bf16 rounding errors; 0 = the
 *   worst case 2^-7 per product), "batch_slab_mb" / "batch_growth" / "batch_first" (slab schedule of the slab pipeline),
 *   "batch_debug" (test / diagnosis bits: 4096 = no pace gate in the 768-d filtering GEMM, 16384 = one wave of workgroup 1 pretends its
 *   split-barrier wait timed out (its workgroup's queries take the exact path), 65536 = the device-side retry re-scores every
 *   survivor; any other bit is refused), "batch_prof_ptr" (diagnosis: device address of a [256 x 8][20] u32 buffer — the filtering GEMM at
 *   D = 384 / 768 then runs its phase-timing build, which leaves per-wave s_memtime cycle counts there, same answers, ~10 % slower;
 *   0 (default) = the product kernel; tools/gemm_phase_budget.py),
 *   "batch_host_multi" (host-pointer batches of at least 16 queries that the MFMA pipelines do not take — top_k 81 .. 192 on a store below
 *   the one-pass floor, "batch_mode" 0 —: 1 (default) = up to 16 queries share one exact pass over the f32 store, bit-identical to the
 *   single-query kernel; 0 = one scan per query),
 *   "batch_in_wait" (device-resident batches: 0 (default) = the library's stream is ordered behind the caller's `stream` by an event
 *   only while that stream still has work pending — a drained stream costs no marker / barrier packet; 1 = always, as before round 6).
 * get-only
 *   "variant_count", "scan_grid", "store_ptr" (device address of the f32 slab), "fused_max_k", "batch_queries", "query_args_scans", "merged_scans", "done_flag_waits",
 *   "batch_inline_retries", "batch_max_row_err_e9", "batch_fallbacks", "onepass_queries", "batch_max_k", "batch_retries",
 *   "batch_multi_passes", "batch_multi_queries", "batch_multi_group" / "batch_multi_group_big", "filter_device_searches",
 *   "mirror_rows_converted" / "mirror_conversions" (bf16 mirror: rows converted / conversions enqueued so far — an append converts its own
 *   rows only), "idhash_rows_inserted", "batch_max_norm_e6".
 * sharded handles: every key above is forwarded to all shards; plus
 *   "ticket_path" (single queries: 1 (default) = every shard that holds rows answers through its own ticket — one launch per shard,
 *   submitted side by side by the handle's worker threads, hits straight to pinned memory — and the host merges G x k keys;
 *   0 = the device gather below), "exchange" (device gather: 0 = peer copies + merge on the first device, 1 = one RCCL all-gather per
 *   query; 1 implies the device gather), "gather" (0 = device; 2 = through the host: what a handle falls back to when a device pair
 *   lacks peer access), "shard_min_mb" (a shard's block holds at least this many MB of rows, default 64: a store below it lives on
 *   the first device only; 0 = spread from the first row; env WAX_HIP_SHARD_MIN_MB sets the default),
 *   get-only "shards", "block_rows", "rebalances", "rccl_collectives", "rccl_ranks" (ncclCommCount of the in-library communicator),
 *   "peer_pairs" / "peer_enabled", "ticket_searches" / "single_shard_searches" / "inline_fanouts" (single-query fan-outs submitted from the caller's thread because the workers were busy with a long job), "parallel_submits", "parallel_collects". */
int wax_hip_set_tuning(wax_hip_engine* e, const char* key, int64_t value);
int64_t wax_hip_get_tuning(wax_hip_engine* e, const char* key);
/* Times `iters` back-to-back launches of ONLY the scan kernel for `query`
 * with HIP events on the engine's stream; returns average ms per launch
 * (kernel sweeps; bench.py's roofline uses the per-launch events of its timed
 * region instead: "time_kernels" + wax_hip_stats). */
int wax_hip_time_scan_kernel(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k,
                             uint32_t iters, double* out_avg_ms);
/* Pure streaming-read microbenchmark over the engine's own store (sum of all
 * float4s, no top-k): the node's achievable read bandwidth for this layout. */
int wax_hip_time_stream_read(wax_hip_engine* e, uint32_t iters, double* out_avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* WAX_HIP_H */
