// engine.hip — host side of libwaxhip: the HBM-resident store, the scratch-slot pool,
// the reader/writer lock and the C ABI declared in include/wax_hip.h.
//
// Behaviour follows MetalVectorEngine (Sources/WaxVectorSearch/MetalVectorEngine.swift),
// re-designed for a discrete MI355X: the store is one hipMalloc slab [capacity x dims] f32
// row-major (+ a u64 frame-id table beside it so id mapping also happens on device), queries
// travel through pinned staging, every search runs on a pooled scratch slot with its own HIP
// stream (the analogue of the transient buffer pool, :84-117), and there is no CPU fallback.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"

using namespace wax;

namespace {

thread_local std::string g_last_error;


int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr, code, what)                                                         \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail((code), std::string(what) + ": " + hipGetErrorString(_e));        \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) {
            changed = (hipSetDevice(dev) == hipSuccess);
        }
    }
    ~DeviceGuard() {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
};

// Writer-preferring reader/writer lock with thread-agnostic unlock (a ticket may be
// collected on another thread). Same contract as AsyncReadWriteLock
// (WaxCore/Concurrency/ReadWriteLock.swift:79-156).
class RWLock {
  public:
    // reentrant = the caller already holds a shared lock (an uncollected search ticket): it must not queue
    // behind a waiting writer, or the writer (waiting for readers == 0) and the reader (waiting for the
    // writer) deadlock. With readers_ > 0 no writer can be active, so skipping the preference is safe.
    void lock_shared(bool reentrant = false) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !writer_ && (reentrant || writers_waiting_ == 0); });
        ++readers_;
    }
    void unlock_shared() {
        std::unique_lock<std::mutex> g(m_);
        if (--readers_ == 0) cv_.notify_all();
    }
    void lock() {
        std::unique_lock<std::mutex> g(m_);
        ++writers_waiting_;
        cv_.wait(g, [&] { return !writer_ && readers_ == 0; });
        --writers_waiting_;
        writer_ = true;
    }
    void unlock() {
        std::unique_lock<std::mutex> g(m_);
        writer_ = false;
        cv_.notify_all();
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    int readers_ = 0;
    int writers_waiting_ = 0;
    bool writer_ = false;
};

struct WriteGuard {
    RWLock& l;
    explicit WriteGuard(RWLock& l_) : l(l_) { l.lock(); }
    ~WriteGuard() { l.unlock(); }
};

// frameId -> row. Open addressing, linear probing, backward-shift deletion. Replaces the
// reference's O(N) `frameIds.firstIndex(of:)` (MetalVectorEngine.swift:334, 385, 426).
class IdMap {
  public:
    IdMap() { resize_table(1024); }
    int64_t find(uint64_t id) const {
        size_t i = hash(id) & mask_;
        while (used_[i]) {
            if (keys_[i] == id) return (int64_t)vals_[i];
            i = (i + 1) & mask_;
        }
        return -1;
    }
    void put(uint64_t id, uint32_t row) {
        if ((size_ + 1) * 10 > (mask_ + 1) * 6) grow();
        size_t i = hash(id) & mask_;
        while (used_[i]) {
            if (keys_[i] == id) { vals_[i] = row; return; }
            i = (i + 1) & mask_;
        }
        used_[i] = 1; keys_[i] = id; vals_[i] = row; ++size_;
    }
    // remove `id` (stored at row `row`) and renumber every row above it down by one
    void erase_row(uint64_t id, uint32_t row) {
        erase_only(id);
        for (size_t s = 0; s <= mask_; ++s)
            if (used_[s] && vals_[s] > row) --vals_[s];
    }
    void erase_only(uint64_t id) {
        size_t i = hash(id) & mask_;
        while (used_[i] && keys_[i] != id) i = (i + 1) & mask_;
        if (used_[i]) {
            size_t hole = i, j = i;
            for (;;) {
                j = (j + 1) & mask_;
                if (!used_[j]) break;
                const size_t home = hash(keys_[j]) & mask_;
                // can entry j move into the hole? yes iff hole lies cyclically in [home, j)
                const bool movable = (hole <= j) ? (home <= hole || home > j) : (home <= hole && home > j);
                if (movable) {
                    keys_[hole] = keys_[j]; vals_[hole] = vals_[j];
                    hole = j;
                }
            }
            used_[hole] = 0;
            --size_;
        }
    }
    void clear() { resize_table(1024); }
    void reserve(size_t n) {
        size_t want = 1024;
        while (want * 6 < n * 10 + 10) want <<= 1;
        if (want > mask_ + 1) rehash(want);
    }
    size_t size() const { return size_; }

  private:
    static uint64_t hash(uint64_t x) {
        x += 0x9e3779b97f4a7c15ull;
        x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
        x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
        return x ^ (x >> 31);
    }
    void resize_table(size_t cap) {
        keys_.assign(cap, 0); vals_.assign(cap, 0); used_.assign(cap, 0);
        mask_ = cap - 1; size_ = 0;
    }
    void rehash(size_t cap) {
        std::vector<uint64_t> ok; std::vector<uint32_t> ov; std::vector<uint8_t> ou;
        ok.swap(keys_); ov.swap(vals_); ou.swap(used_);
        resize_table(cap);
        for (size_t s = 0; s < ou.size(); ++s)
            if (ou[s]) put(ok[s], ov[s]);
    }
    void grow() { rehash((mask_ + 1) * 2); }
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> vals_;
    std::vector<uint8_t> used_;
    size_t mask_ = 0, size_ = 0;
};

constexpr int kShardRing = 8;

// A scan's stage buffer: per-workgroup partial top-k lists, followed by one cache line holding the arrival ticket of the
// fused final merge (scan_epilogue: zero between launches — the last arriver re-arms it).
constexpr size_t kPartialsBytes = (size_t)MAX_GRID_BLOCKS * FUSED_MAX_K * sizeof(int64_t) + 128;
inline uint32_t* partials_ticket(int64_t* d_partials) {
    return reinterpret_cast<uint32_t*>(d_partials + (size_t)MAX_GRID_BLOCKS * FUSED_MAX_K);
}

constexpr int kMaxStreams = 4;

struct Slot {
    int index = 0;
    hipStream_t stream = nullptr;  // one of the engine's streams (not owned)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    uint64_t* h_done = nullptr;    // pinned: the completion word a fused scan publishes behind its hits ("done_flag")
    bool coherent = false;         // h_hits / h_done are coherent host memory (a condition of the completion-word path)
    uint64_t done_seq = 0;         // value the slot's current query publishes
    bool flag_wait = false;        // this ticket completes through h_done (no event was recorded)
    hipEvent_t t_start = nullptr, t_end = nullptr;   // the events that bracket this ticket's scan kernel (not owned)
    float* d_query = nullptr;
    float* h_query = nullptr;  // pinned
    int64_t* d_partials = nullptr;
    wax_hip_hit* d_hits = nullptr;
    wax_hip_hit* h_hits = nullptr;  // pinned
    // general path, allocated on first use
    float* d_dist = nullptr;
    uint64_t dist_cap = 0;
    SelectWork sw{};
    bool sw_ready = false;
    // per-ticket state
    int k_eff = 0;
    bool timed = false;
    std::thread::id owner;         // the submitting thread (its outstanding-ticket count drops at collect, whoever collects)
};

// Workspace of one wax_hip_search_filtered call: pooled (filter_max per engine, allocated on demand) with its own stream,
// so filtered searches run concurrently with each other and with every other read entry point.
struct FilterWork {
    hipStream_t stream = nullptr;
    uint32_t* d_rows = nullptr;      // [cap] allowed local rows, ascending
    uint64_t* d_ids = nullptr;       // [cap] their frame ids
    float* d_dist = nullptr;         // [cap] their distances
    uint64_t cap = 0;
    uint64_t* d_allow = nullptr;     // [allow_cap] the caller's allow-list (device-side probe)
    uint64_t allow_cap = 0;
    uint32_t* d_bitmap = nullptr;    // [bitmap_words] one bit per store row
    uint64_t bitmap_words = 0;
    uint32_t* d_block_sum = nullptr; // [block_cap] per-block popcounts -> exclusive offsets
    uint64_t block_cap = 0;
    uint32_t* d_total = nullptr;     // [1]
    uint32_t* h_total = nullptr;     // pinned [1]
    float* d_query = nullptr;        // [dims]
    float* d_qnorm = nullptr;        // [1]
    wax_hip_hit* d_hits = nullptr;   // [WAX_HIP_MAX_RESULTS]
    wax_hip_hit* h_hits = nullptr;   // pinned [WAX_HIP_MAX_RESULTS]
    SelectWork sw{};
};

void free_filter_work(FilterWork* f);

// id -> row table in HBM (filter.hip): built at the first long allow-list, rebuilt lazily after a mutation.
struct IdHash {
    std::mutex mu;                   // serialises the (re)build only
    uint32_t* d_table = nullptr;
    uint64_t slots = 0;
    std::atomic<bool> valid{false};  // the table holds exactly rows [0, count)
    // Round 6: appends do not discard the table — rows [rows, count) are inserted by the next long allow-list (an upsert keeps id -> row;
    // a removal shifts every later row and a deserialize replaces them all: those start over, like the reference's firstIndex(of:) scan
    // they are O(N) anyway).
    uint64_t rows = 0;               // rows [0, rows) are in the table (guarded by mu / the exclusive engine lock)
    bool stale = true;               // the table must be rebuilt from row 0
};

// Batched (bf16 MFMA) path. The corpus mirror is shared by every batch (read-only during searches, rebuilt lazily by
// the first batch after a mutation); everything a call writes lives in a pooled per-call workspace with its own
// stream, so batched searches run concurrently with each other like every other read entry point.
struct BatchMirror {
    std::mutex mu;                       // serialises the (re)build only
    unsigned short* d_cb = nullptr;      // [mirror_cap][dims] bf16 (cosine rows pre-normalised)
    float* d_vn2 = nullptr;              // [mirror_cap] ||v||^2
    unsigned int* d_maxnorm = nullptr;
    uint64_t mirror_cap = 0;
    std::atomic<bool> mirror_valid{false};   // fast path: nothing to convert (rows == count, no dirty row, not stale, large enough)
    // Round 6: a mutation no longer throws the mirror away (rounds 1-5 re-converted the WHOLE store after any add / remove: 30.7 GB read
    // + 15.4 GB written and a blocking sync at 10M x 768, per mutation, on an ingest-while-serving workload — the reference's add is one
    // row copy, MetalVectorEngine.swift:330-357). Appends leave rows [rows, count) to convert; an upsert lists its row in `dirty`; a
    // removal moves the mirror's tail down like the store's (:431-438); only reallocation and deserialize convert everything again.
    // max ||v|| and max ||x - bf16(x)|| stay on the device (d_maxnorm, read there by the prep kernel) and only ever grow between full
    // conversions: a bound that is too large by a row that has since been overwritten or removed is still a bound.
    // Writers hold the exclusive engine lock; ensure_mirror holds `mu` under the shared one.
    uint64_t rows = 0;                   // rows [0, rows) of the mirror match the store, except `dirty`
    std::vector<uint32_t> dirty;         // rows < `rows` overwritten since (upserts); more than kMirrorMaxDirty of them = stale
    bool stale = true;                   // convert everything again
    uint32_t* d_dirty = nullptr;         // device copy of `dirty` for the listed-rows conversion
    hipEvent_t ev_ready = nullptr;       // recorded behind the last conversion: other workspaces' streams wait for it
    bool ev_pending = false;
    std::atomic<uint64_t> rows_converted{0}, conversions{0};   // statistics ("mirror_rows_converted" / "mirror_conversions")
    std::atomic<int> mirror_wanted{0};   // small batches since the last mutation that a VALID mirror would have made cheaper
};

constexpr size_t kMirrorMaxDirty = 4096; // upserted rows remembered one by one; beyond that the mirror is converted again as a whole
constexpr uint32_t kBatchMaxQ = 1024;   // queries per GEMM pass
constexpr uint32_t kBatchSegBase = FUSED_MAX_K;   // slab pipeline: a candidate row = [0, kBatchSegBase) best list, then the survivor area
constexpr uint32_t kBatchSegArea = 4096;          // survivors per query between two tighten passes (slab pipeline) / of the one pass
constexpr uint32_t kBatchCandCap = kBatchSegBase + kBatchSegArea;
constexpr uint32_t kBatchMaxSegs = 256;           // GEMM workgroups per query group
constexpr uint32_t kBatchFirstSlab = 2048;  // rows of the first slab (everything passes: tau = +inf)
constexpr int kBatchMaxKSlab = 80;      // largest k served by the slab pipeline (k' = 2k+32 <= 192)
constexpr int kBatchMaxK = 464;         // largest k served by the one-pass pipeline (k' = 2k+32 <= 960): covers the real caller's
                                        // candidateLimit = max(topK, min(3 topK, 1000)) up to topK 154 (UnifiedSearch.swift:1195-1200)

struct BatchCtx {
    hipStream_t stream = nullptr;        // owned
    hipEvent_t ev_in = nullptr;          // caller-stream -> ctx-stream ordering for device-resident inputs
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;   // "time_kernels": around the filtering GEMM of the last enqueued block
    hipEvent_t ev_gc = nullptr;                    // "time_kernels" = 2: the chain event behind it (ev_g0 / ev_g1 are bound to the dispatch)
    bool gemm_timed = false;
    uint32_t gemm_rows = 0, gemm_queries = 0;
    float* d_q = nullptr;                // [q_cap][dims] staging for host queries
    uint64_t q_cap = 0;
    unsigned short* d_qb = nullptr;
    unsigned short* d_qf = nullptr;                // the same queries in MFMA A-fragment order (GemmArgs::qf)
    float* d_qn2 = nullptr;
    float* d_qnorm = nullptr;
    float* d_eps = nullptr;
    float* d_tau = nullptr;              // [kBatchMaxQ] admission thresholds
    float* d_dense = nullptr;            // slab pipeline: [kBatchMaxQ][kBatchFirstSlab] first-slab score tile
    uint32_t* d_cand_count = nullptr;
    uint32_t* d_overflow = nullptr;
    int64_t* d_cand = nullptr;           // [rows of a block][slots per query]; cand_slots = capacity in keys
    uint64_t cand_slots = 0;
    uint32_t* d_seg_count = nullptr;     // [kBatchMaxSegs][kBatchMaxQ] survivors per (GEMM workgroup, query)
    int64_t* d_exact = nullptr;          // [kBatchMaxQ][kp_cap]
    int64_t* d_sel = nullptr;            // [kBatchMaxQ][kp_cap] (one-pass pipeline, k' > 192)
    uint64_t kp_cap = 0;
    float* d_tile_max = nullptr;         // one-pass pipeline: [tile_max_rows][kBatchMaxQ]
    uint64_t tile_max_rows = 0;
    wax_hip_hit* d_hits = nullptr;       // [hits_cap] hits of a whole host-pointer call
    uint64_t hits_cap = 0;
    uint64_t cert_cap = 0;
    wax_hip_hit* h_hits = nullptr;       // pinned [hits_cap]
    // full retry of uncertified queries (single-block one-pass batches): the finish arguments of the last block, the failed
    // queries' numbers (pinned + device)
    FinishArgs last_finish{};
    int last_metric = 0;
    bool last_finish_valid = false;
    uint32_t* h_qlist = nullptr;         // pinned [kBatchMaxQ]
    uint32_t* d_qlist = nullptr;         // [kBatchMaxQ]
    // shared exact pass over the uncertified queries (multiscan.hip): their norms by slot, per-(query, workgroup) partial top-k
    float* h_fnorm = nullptr;            // pinned [kBatchMaxQ]
    float* d_fnorm = nullptr;            // [kBatchMaxQ]
    int64_t* d_mpart = nullptr;          // [group][grid][k]
    uint64_t mpart_cap = 0;
    int64_t* d_rescue = nullptr;         // full retry: [queries of a round][survivor area] dense survivor keys
    uint64_t rescue_cap = 0;
    int64_t* d_rescue_exact = nullptr;   // full retry: [queries of a round][largest live count] their exact keys
    uint64_t rescue_exact_cap = 0;
    uint32_t* h_live = nullptr;          // pinned [kBatchMaxQ]: live survivors per query of a retry round
    uint32_t* h_cert = nullptr;          // pinned [cert_cap]
    uint32_t* d_cert = nullptr;          // device [cert_cap]: the finish kernel's flags for the device-side retry kernel
    float* h_qnorm = nullptr;            // pinned [cert_cap]: exact norms (the exact-path fallback needs them on the host)
};

struct ShardedState;   // sharded.inc: the multi-GPU handle's state (null for a single-device engine)

}  // namespace

struct wax_hip_engine {
    ShardedState* sh = nullptr;
    int device = 0;
    uint8_t metric = 0;
    uint32_t dims = 0;
    uint64_t count = 0;           // vectorCount
    uint64_t capacity = 0;        // reservedCapacity, rows
    float* d_store = nullptr;     // [capacity][dims]
    uint64_t* d_ids = nullptr;    // [capacity]
    std::vector<uint64_t> ids;    // frameIds (host mirror; row order)
    IdMap idmap;
    uint64_t row_base = 0;
    RWLock lock;

    hipStream_t streams[kMaxStreams] = {};
    int n_streams = 2;            // slots are spread round-robin over this many in-order streams
    // Scan kernels are chained across streams through this event so that they never overlap each
    // other (each one owns the whole HBM pipe and its HIP-event duration stays meaningful) while
    // the merge kernel, the query upload and the result write of neighbouring queries do overlap.
    hipEvent_t scan_done = nullptr;
    hipEvent_t chain_event = nullptr;  // event the next chained scan waits on: scan_done or the last scan's end-of-kernel timing event
    bool scan_done_valid = false;
    // End-of-kernel timing events of chained scans come from this ring (not from the slot): the end of scan i is also
    // the START of scan i+1 when i is still in flight at i+1's submit, so only two packets (record, wait) sit between
    // two scans instead of three. A ring entry is re-recorded kTimingRing chained scans later — more than the
    // kHardSlotCap tickets that can be outstanding — so both tickets that read it have been collected by then.
    static constexpr int kTimingRing = 1024;
    hipEvent_t tev[kTimingRing] = {};
    uint32_t tev_next = 0;
    bool chain_is_timing = false;
    std::atomic<int64_t> share_timing{1};
    // Pipelined scans of different streams: 1 = chained through an event (they never overlap: a per-launch HIP-event
    // duration is one scan alone), 0 = free to overlap (no idle HBM between two scans, ramp and tail of neighbouring
    // kernels hidden: +3 .. +19 % queries/s), -1 (default) = chained exactly when the kernels are being timed
    // ("time_kernels" = 1), so the product path is the fast one and a measurement pass still gets clean per-launch times.
    std::atomic<int64_t> scan_chain{-1};
    std::mutex chain_mu;

    std::mutex slot_mu;
    std::condition_variable slot_cv;
    std::vector<Slot*> all_slots;
    std::vector<Slot*> free_slots;
    int max_slots = 4;
    std::map<uint64_t, Slot*> tickets;
    uint64_t next_ticket = 1;
    // Uncollected search tickets per SUBMITTING thread (a ticket holds the shared lock until it is collected, possibly
    // by another thread): lets a thread that already holds the lock re-enter past a queued writer, never wait for a
    // scratch slot, and be refused by the mutating entry points instead of dead-locking on itself.
    std::mutex out_mu;
    std::map<std::thread::id, int> outstanding;

    // shard-search scratch ring (caller-stream async work)
    float* ring_d_query[kShardRing] = {};
    float* ring_h_query[kShardRing] = {};
    int64_t* ring_d_partials[kShardRing] = {};
    hipEvent_t ring_ev0[kShardRing] = {}, ring_ev1[kShardRing] = {};
    hipEvent_t ring_t0[kShardRing] = {}, ring_t1[kShardRing] = {};   // the events that bracket the entry's scan (not owned)
    bool ring_ev_pending[kShardRing] = {};
    // Completion of everything the entry's last use enqueued on the caller's stream (query upload, scan, merge):
    // waited for before the entry is reused (its pinned query and partials are still being read until then) and by
    // every writer (the caller-stream work runs after the shared lock was released).
    hipEvent_t ring_done[kShardRing] = {};
    bool ring_busy[kShardRing] = {};
    std::mutex ring_mu[kShardRing];
    std::atomic<uint32_t> ring_next{0};

    float* d_sink = nullptr;
    void* d_bounce = nullptr;

    // tuning
    std::atomic<int64_t> grid_blocks{0};
    std::atomic<int64_t> variant{-1};
    std::atomic<int64_t> time_kernels{0};
    std::atomic<int64_t> force_general{0};
    std::atomic<int64_t> stream_nt{1};
    std::atomic<int64_t> scan_plain_mb{32};  // single-query scans with the query in their arguments: stores up to this many MB read their rows with ordinary loads (-1 = grids <= 160 workgroups)
    std::atomic<int64_t> merge_kway{1};      // fused final merge, k <= 64: 1 (default) = k-way merge of the per-workgroup lists' heads; 0 = stream them through the wave lists
    std::atomic<int64_t> done_flag{1};       // single-query scans that merge in the kernel publish a completion word in pinned memory; collect polls it instead of an event (0 = always an event)
    std::atomic<uint64_t> st_flag_waits{0};
    std::atomic<int64_t> query_args{1};      // single-query scans: 1 (default) = stores whose scan grid is small enough for the fused merge (the launch-latency-bound ones) get the query in the kernel arguments (no upload copy); 2 = every store; 0 = always upload
    std::atomic<int64_t> fuse_merge{1};      // 1 = grids of <= SCAN_FUSE_MERGE_GRID workgroups merge in the scan kernel's last-arriving workgroup
    // A scan submitted while other tickets of this engine are still out (a caller that keeps several queries in flight, or several
    // callers at once) runs in a stream of scans: there the separate merge launch overlaps the NEXT scan, while the last arriver's
    // tail (ticket, k-way merge, id gather, ~8 us) is serial inside the kernel. Stores of at least this many MB then take the
    // two-launch form (upload, scan, merge; 0 = never). A query submitted alone keeps the single launch (12 - 14 us less latency).
    // Pipelined, depth 4, 384-d (profiles/r05/k_pipelined_merge_forms.txt): 100K rows 24.8 us per query in one launch against 30.1
    // in two, 200K 45.3 / 45.6, 400K 88.9 / 87.0, 700K 154.0 / 150.1, 1M 224.3 / 214.3, 1.25M 274.3 / 268.7.
    std::atomic<int64_t> merge_overlap_mb{400};
    std::atomic<uint64_t> st_overlap_scans{0};
    std::atomic<int64_t> batch_qfrag{1};     // 1 (default) = the prep kernel also writes the bf16 queries in MFMA A-fragment order and the register-resident GEMM loads them from there (coalesced); 0 = row-major reads
    std::atomic<int64_t> batch_min{1};       // fewer queries than this: always pipelined single-query scans (1..15: cost model below)
    std::atomic<int64_t> batch_mode{1};      // 0 = never use the MFMA path
    std::atomic<int64_t> batch_slab_mb{64};  // cap on slab size, in units of 16 384 rows
    std::atomic<int64_t> batch_growth{8};    // next slab = growth x rows seen so far
    std::atomic<int64_t> batch_first{2048};  // rows of the dense first slab (<= kBatchFirstSlab)
    std::atomic<int64_t> batch_prof_ptr{0};  // diagnosis: device address of the phase-timing buffer of the filtering GEMM (GemmArgs::prof); 0 = the product kernel
    std::atomic<int64_t> batch_debug{0};     // test / diagnosis bits, none of which can change an answer: 4096 = no pace gate, 16384 = one wave of workgroup 1 pretends its split-barrier wait timed out, 65536 = the device-side retry re-scores every survivor
    std::atomic<int64_t> batch_rega{5};      // register-resident-queries GEMM where it applies: 5 (default) split tile barrier, 1 workgroup barrier per tile; 0 = the LDS-tiled kernel instead
    std::atomic<uint64_t> st_batch_queries{0}, st_batch_fallbacks{0}, st_idhash_rows{0};
    BatchMirror batch;
    std::mutex bctx_mu;
    std::condition_variable bctx_cv;
    std::vector<BatchCtx*> bctx_all, bctx_free;
    int bctx_max = 4;
    std::atomic<int64_t> batch_onepass{1};        // 0 = always the slab pipeline
    std::atomic<int64_t> batch_onepass_tiles{1024};   // smallest store (in GEMM tiles) the one-pass pipeline takes
    std::atomic<int64_t> batch_survivors{3};      // one-pass pipeline: expected survivors per query = this x k'
    std::atomic<int64_t> batch_kp_fused{1};       // one-pass pipeline, k in 81 .. 128: 1 = k' capped at 192 (fused finish kernel + device retry); 0 = k' = 2k + 32 (three-launch finish)
    std::atomic<int64_t> batch_retry{1};          // one-pass pipeline: uncertified queries get a full retry (ALL their survivors re-scored) before the exact path
    std::atomic<int64_t> batch_multi{1};          // exact path of a batch: 1 = uncertified queries share passes over the f32 store (multiscan.hip), 0 = one scan each
    std::atomic<uint64_t> st_multi_passes{0}, st_multi_queries{0};
    std::atomic<uint64_t> st_batch_retries{0};
    std::atomic<int64_t> batch_sample_div{32};    // one-pass pipeline: 1 / this of the tiles are sampled (at least 256)
    std::atomic<uint64_t> st_onepass_queries{0};
    std::atomic<int64_t> batch_eps_measured{1};       // cosine certificate bound from the MEASURED bf16 rounding errors (per query, max over rows) instead of the worst case
    std::atomic<int> retry_hint{0};                   // > 0: batches carry the device-side retry kernel behind their finish kernel
    std::atomic<uint64_t> st_batch_inline_retries{0};  // ... of which inside the finish kernel (no host round trip)
    std::atomic<uint64_t> st_merged_scans{0};        // single-query scans whose last-arriving workgroup did the final merge (one launch per query)
    std::atomic<uint64_t> st_query_args{0};          // single-query scans that took their query through the kernel arguments
    // wax_hip_search_batch_submit_device tickets (guarded by bticket_mu)
    struct BatchTicket {
        BatchCtx* c = nullptr;           // null: the batch was answered at submit time (empty engine / loop path)
        const float* d_queries = nullptr;
        wax_hip_hit* d_out = nullptr;
        uint32_t nq = 0, out_stride = 0;
        int k_eff = 0;
        std::thread::id owner;
    };
    std::mutex gemm_chain_mu;                     // "time_kernels": filtering GEMMs of concurrent batches run one after another
    hipEvent_t gemm_chain_ev = nullptr;           // the last one's end-of-kernel event (owned by its workspace)
    std::mutex bticket_mu;
    std::map<uint64_t, BatchTicket> btickets;
    uint64_t next_bticket = 1;
    std::mutex filter_mu;                         // pool of filtered-search workspaces
    std::condition_variable filter_cv;
    std::vector<FilterWork*> filter_all, filter_free;
    int filter_max = 4;
    IdHash idhash;
    std::atomic<int64_t> filter_device_min{4096}; // allow-lists at least this long are resolved on the device
    std::atomic<uint64_t> st_filter_device{0};    // filtered searches whose allow-list was resolved on the device
    // Write-combining of single-frame appends (the reference appends into a unified-memory buffer and the GPU simply
    // sees it, MetalVectorEngine.swift:340-351; with discrete HBM the analogue is a pinned staging area that the NEXT
    // reader — or a full staging area — uploads in one copy). The last `pend_rows` rows of [0, count) live only here.
    float* h_pend = nullptr;                 // pinned, pend_cap rows x dims
    uint64_t pend_cap = 0;
    std::atomic<uint64_t> pend_rows{0};
    std::mutex pend_mu;

    // stats
    std::atomic<uint64_t> st_searches{0}, st_rows{0}, st_bytes{0}, st_alloc{0}, st_reuse{0};
    std::mutex st_mu;
    double st_last_ms = 0.0, st_total_ms = 0.0;
    uint64_t st_timed = 0;
    double st_gemm_ms = 0.0;
    uint64_t st_gemm_timed = 0, st_gemm_rows = 0, st_gemm_queries = 0;
};

namespace {

inline void cpu_relax() {   // a spin-wait hint, per architecture
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

constexpr uint64_t kBounceBytes = 64ull << 20;

// Wait until no shard-path work (wax_hip_search_shard_device: enqueued on caller streams, not synchronised by the
// call) is in flight. Called by writers under the exclusive lock, so no new shard work can start meanwhile.
void sync_shard_work(wax_hip_engine* e) {
    for (int r = 0; r < kShardRing; ++r) {
        std::unique_lock<std::mutex> g(e->ring_mu[r]);
        if (e->ring_busy[r]) {
            (void)hipEventSynchronize(e->ring_done[r]);
            e->ring_busy[r] = false;
        }
    }
}

int holding(wax_hip_engine* e) {   // uncollected tickets submitted by the calling thread
    std::unique_lock<std::mutex> g(e->out_mu);
    auto it = e->outstanding.find(std::this_thread::get_id());
    return it == e->outstanding.end() ? 0 : it->second;
}
void note_submit_id(wax_hip_engine* e, std::thread::id* owner) {
    *owner = std::this_thread::get_id();
    std::unique_lock<std::mutex> g(e->out_mu);
    e->outstanding[*owner] += 1;
}
void note_collect_id(wax_hip_engine* e, std::thread::id owner) {
    std::unique_lock<std::mutex> g(e->out_mu);
    auto it = e->outstanding.find(owner);
    if (it != e->outstanding.end() && --it->second <= 0) e->outstanding.erase(it);
}
void note_submit(wax_hip_engine* e, Slot* s) { note_submit_id(e, &s->owner); }
void note_collect(wax_hip_engine* e, Slot* s) { note_collect_id(e, s->owner); }

// sharded.inc (included at the end of this file)
int sh_add_batch(wax_hip_engine* e, const uint64_t* frame_ids, const float* rows, uint64_t n, uint32_t dims);
int sh_add_batch_device(wax_hip_engine* e, const uint64_t* frame_ids, const float* d_rows, uint64_t n, uint32_t dims);
int sh_remove(wax_hip_engine* e, uint64_t frame_id);
int sh_reserve(wax_hip_engine* e, uint64_t rows);
int sh_submit(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint64_t* out_ticket);
int sh_collect(wax_hip_engine* e, uint64_t ticket, uint64_t* out_ids, float* out_scores, uint32_t capacity, uint32_t* out_count);
int sh_search_batch_hits(wax_hip_engine* e, const float* queries, uint32_t nq, uint32_t dims, int32_t top_k, wax_hip_hit* out_hits,
                         uint32_t stride, uint32_t* out_counts);
int sh_batch_device(wax_hip_engine* e, const float* d_queries, uint32_t nq, uint32_t dims, int32_t top_k, wax_hip_hit* d_out_hits,
                    uint32_t out_stride, void* stream, uint64_t* ticket);
int sh_batch_collect_device(wax_hip_engine* e, uint64_t ticket, uint32_t* out_fallbacks);
int sh_search_filtered(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, int has_allow, const uint64_t* allow,
                       uint64_t n_allow, int has_min, float min_score, uint64_t* out_ids, float* out_scores, uint32_t capacity,
                       uint32_t* out_count);
int sh_serialize(wax_hip_engine* e, uint8_t** out_bytes, size_t* out_len);
int sh_deserialize(wax_hip_engine* e, const uint8_t* data, size_t len);
void sh_destroy(wax_hip_engine* e);
uint64_t sh_total_count(const wax_hip_engine* e);
int sh_stats(wax_hip_engine* e, wax_hip_stats_t* out);
int sh_set_tuning(wax_hip_engine* e, const std::string& k, int64_t value);
int64_t sh_get_tuning(wax_hip_engine* e, const std::string& k);
#define SHARDED_UNSUPPORTED(e, what)                                                                                 \
    do {                                                                                                             \
        if ((e) && (e)->sh) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, what " is a single-device entry point: not available on a sharded engine"); \
    } while (0)
// Mutating entry points: a thread holding uncollected tickets holds the shared lock and would wait for itself.
#define REFUSE_IF_HOLDING(e)                                                                                        \
    do {                                                                                                            \
        if (holding(e) > 0)                                                                                         \
            return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "collect outstanding search tickets first (a ticket holds the engine's read lock)"); \
    } while (0)

int clamp_topk(int64_t top_k) {  // MetalVectorEngine.swift:842-846
    if (top_k < 1) return 1;
    if (top_k > WAX_HIP_MAX_RESULTS) return WAX_HIP_MAX_RESULTS;
    return (int)top_k;
}

std::string dim_mismatch_msg(uint32_t expected, uint64_t got) {  // MetalVectorEngine.swift:832
    return "vector dimension mismatch: expected " + std::to_string(expected) + ", got " + std::to_string(got);
}

int alloc_slot(wax_hip_engine* e, Slot** out) {
    Slot* s = new Slot();
    auto bail = [&](int code, const char* what, hipError_t err) {
        std::string msg = std::string("Failed to allocate ") + what + ": " + hipGetErrorString(err);
        if (s->ev0) (void)hipEventDestroy(s->ev0);
        if (s->ev1) (void)hipEventDestroy(s->ev1);
        if (s->ev_done) (void)hipEventDestroy(s->ev_done);
        (void)hipFree(s->d_query); (void)hipHostFree(s->h_query); (void)hipFree(s->d_partials);
        (void)hipFree(s->d_hits); (void)hipHostFree(s->h_hits); (void)hipHostFree(s->h_done);
        delete s;
        return fail(code, msg);
    };
    hipError_t err;
    // kernel-timing events: device-scope release (no system-scope cache flush folded into the interval or into the
    // gap before the next scan); ev_done is what the host waits on for results in pinned memory: default (system) scope
    if ((err = hipEventCreateWithFlags(&s->ev0, hipEventReleaseToDevice)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "event", err);
    if ((err = hipEventCreateWithFlags(&s->ev1, hipEventReleaseToDevice)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "event", err);
    if ((err = hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "event", err);
    const size_t qbytes = (size_t)e->dims * sizeof(float);
    if ((err = hipMalloc(&s->d_query, qbytes)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "transient query buffer", err);
    if ((err = hipHostMalloc(&s->h_query, qbytes, hipHostMallocDefault)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "pinned query buffer", err);
    if ((err = hipMalloc(&s->d_partials, kPartialsBytes)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "top-k stage buffer", err);
    if ((err = hipMemset(partials_ticket(s->d_partials), 0, 128)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "top-k stage buffer", err);
    // hipMemset on device memory may return before the fill has run, and the slot's scans run on non-blocking streams the
    // null stream does not order against: wait once, here, so the first fused scan can only ever see an armed (zero) ticket
    if ((err = hipStreamSynchronize(nullptr)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "top-k stage buffer", err);
    if ((err = hipMalloc(&s->d_hits, (size_t)WAX_HIP_MAX_RESULTS * sizeof(wax_hip_hit))) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "top-k results buffer", err);
    // The completion-word protocol ("done_flag") needs the kernel's writes — hits, then the word — visible to the host IN ORDER while
    // the kernel is still running: coherent (fine-grained) host memory, asked for explicitly (the default is coherent on this stack
    // today, but HIP_HOST_COHERENT=0 or another platform changes that silently). Where a coherent allocation is refused the slot falls
    // back to default pinned memory and never takes the completion-word path (collect waits on the event behind the kernel).
    s->coherent = true;
    if ((err = hipHostMalloc(&s->h_hits, (size_t)WAX_HIP_MAX_RESULTS * sizeof(wax_hip_hit), hipHostMallocCoherent)) != hipSuccess) {
        (void)hipGetLastError();
        s->coherent = false;
        if ((err = hipHostMalloc(&s->h_hits, (size_t)WAX_HIP_MAX_RESULTS * sizeof(wax_hip_hit), hipHostMallocDefault)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "pinned results buffer", err);
    }
    if ((err = hipHostMalloc(&s->h_done, 64, s->coherent ? hipHostMallocCoherent : hipHostMallocDefault)) != hipSuccess) {
        (void)hipGetLastError();
        s->coherent = false;
        if ((err = hipHostMalloc(&s->h_done, 64, hipHostMallocDefault)) != hipSuccess) return bail(WAX_HIP_ERR_ALLOC, "pinned completion word", err);
    }
    *s->h_done = 0;
    *out = s;
    return WAX_HIP_OK;
}

void free_slot(Slot* s) {
    if (!s) return;
    (void)hipFree(s->d_query); (void)hipHostFree(s->h_query); (void)hipFree(s->d_partials);
    (void)hipFree(s->d_hits); (void)hipHostFree(s->h_hits); (void)hipHostFree(s->h_done); (void)hipFree(s->d_dist);
    if (s->sw_ready) free_select_work(&s->sw);
    (void)hipEventDestroy(s->ev0); (void)hipEventDestroy(s->ev1); (void)hipEventDestroy(s->ev_done);
    delete s;
}

// acquireTransientBuffers (MetalVectorEngine.swift:84-113): reuse a pooled slot or create one.
constexpr int kSlotBusy = 1;  // internal: try_only and every slot is in use

// `holding`: the calling thread already owns tickets (= slots). It must never WAIT for a slot — two threads that each
// hold half of the pool and want one more would wait for each other forever — so, like the reference's transient
// buffer pool (MetalVectorEngine.swift:84-117: an empty pool allocates), it gets a freshly allocated slot beyond
// `max_slots` (counted in transient_allocations; the slot stays pooled afterwards), up to kHardSlotCap.
constexpr int kHardSlotCap = 256;

int acquire_slot(wax_hip_engine* e, Slot** out, bool try_only = false, bool holding = false) {
    std::unique_lock<std::mutex> g(e->slot_mu);
    for (;;) {
        if (!e->free_slots.empty()) {
            Slot* s = e->free_slots.back();
            e->free_slots.pop_back();
            s->stream = e->streams[s->index % e->n_streams];  // idle slot: safe to re-home
            e->st_reuse++;
            *out = s;
            return WAX_HIP_OK;
        }
        if ((int)e->all_slots.size() < e->max_slots) {
            Slot* s = nullptr;
            int rc = alloc_slot(e, &s);
            if (rc != WAX_HIP_OK) return rc;
            s->index = (int)e->all_slots.size();
            s->stream = e->streams[s->index % e->n_streams];
            e->all_slots.push_back(s);
            e->st_alloc++;
            *out = s;
            return WAX_HIP_OK;
        }
        if (try_only) return kSlotBusy;
        if (holding) {
            if ((int)e->all_slots.size() >= kHardSlotCap)
                return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "too many outstanding search tickets (collect some first)");
            Slot* s = nullptr;
            int rc = alloc_slot(e, &s);
            if (rc != WAX_HIP_OK) return rc;
            s->index = (int)e->all_slots.size();
            s->stream = e->streams[s->index % e->n_streams];
            e->all_slots.push_back(s);
            e->st_alloc++;
            *out = s;
            return WAX_HIP_OK;
        }
        e->slot_cv.wait(g);
    }
}

void release_slot(wax_hip_engine* e, Slot* s) {  // releaseTransientBuffers (:115-117)
    std::unique_lock<std::mutex> g(e->slot_mu);
    e->free_slots.push_back(s);
    e->slot_cv.notify_one();
}

int ensure_general(wax_hip_engine* e, Slot* s) {
    if (s->dist_cap < e->capacity) {
        (void)hipStreamSynchronize(s->stream);
        (void)hipFree(s->d_dist);
        s->d_dist = nullptr; s->dist_cap = 0;
        HIP_TRY(hipMalloc(&s->d_dist, (size_t)e->capacity * sizeof(float)), WAX_HIP_ERR_ALLOC,
                "Failed to allocate transient distances buffer");
        s->dist_cap = e->capacity;
    }
    if (!s->sw_ready) {
        HIP_TRY(alloc_select_work(&s->sw), WAX_HIP_ERR_ALLOC, "Failed to allocate the selection workspace");
        s->sw_ready = true;
    }
    return WAX_HIP_OK;
}

// ||q|| in f64, four independent partial sums (a single chain is add-latency-bound: 0.16 us per 384-d query,
// 160 us for a 1024-query batch). Every path (single query, batch re-score) takes its norm from here.
float query_norm(const float* q, uint32_t dims) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    uint32_t j = 0;
    for (; j + 4 <= dims; j += 4) {
        s0 += (double)q[j] * (double)q[j];
        s1 += (double)q[j + 1] * (double)q[j + 1];
        s2 += (double)q[j + 2] * (double)q[j + 2];
        s3 += (double)q[j + 3] * (double)q[j + 3];
    }
    for (; j < dims; ++j) s0 += (double)q[j] * (double)q[j];
    return (float)std::sqrt((s0 + s1) + (s2 + s3));
}

// Upload the staged appends. Callers hold the engine lock (shared or exclusive); readers may race each other here,
// never a writer (staging is only filled under the exclusive lock). Rows >= count - pend_rows are beyond the n_rows of
// every scan already in flight, so writing them concurrently with those scans is safe.
int flush_pending(wax_hip_engine* e) {
    if (e->pend_rows.load(std::memory_order_acquire) == 0) return WAX_HIP_OK;
    std::unique_lock<std::mutex> g(e->pend_mu);
    const uint64_t n = e->pend_rows.load(std::memory_order_relaxed);
    if (n == 0) return WAX_HIP_OK;
    const uint64_t first = e->count - n;
    HIP_TRY(hipMemcpy(e->d_store + first * e->dims, e->h_pend, (size_t)n * e->dims * sizeof(float), hipMemcpyHostToDevice),
            WAX_HIP_ERR_INTERNAL, "vector upload");
    HIP_TRY(hipMemcpy(e->d_ids + first, e->ids.data() + first, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice),
            WAX_HIP_ERR_INTERNAL, "frame id upload");
    e->pend_rows.store(0, std::memory_order_release);
    return WAX_HIP_OK;
}

// resizeBuffersIfNeeded (MetalVectorEngine.swift:873-890): new slab + copy of the live rows.
int resize_store(wax_hip_engine* e, uint64_t new_cap) {
    if (new_cap <= e->capacity) return WAX_HIP_OK;
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) return frc; }   // the copy below reads device rows
    float* ns = nullptr;
    uint64_t* ni = nullptr;
    const size_t row_bytes = (size_t)e->dims * sizeof(float);
    hipError_t err = hipMalloc(&ns, (size_t)new_cap * row_bytes);
    if (err != hipSuccess) return fail(WAX_HIP_ERR_ALLOC, std::string("Failed to resize vectors buffer: ") + hipGetErrorString(err));
    err = hipMalloc(&ni, (size_t)new_cap * sizeof(uint64_t));
    if (err != hipSuccess) { (void)hipFree(ns); return fail(WAX_HIP_ERR_ALLOC, std::string("Failed to resize frame id buffer: ") + hipGetErrorString(err)); }
    if (e->count > 0) {
        err = hipMemcpy(ns, e->d_store, (size_t)e->count * row_bytes, hipMemcpyDeviceToDevice);
        if (err == hipSuccess) err = hipMemcpy(ni, e->d_ids, (size_t)e->count * sizeof(uint64_t), hipMemcpyDeviceToDevice);
        if (err != hipSuccess) { (void)hipFree(ns); (void)hipFree(ni); return fail(WAX_HIP_ERR_INTERNAL, std::string("store copy failed: ") + hipGetErrorString(err)); }
    }
    (void)hipFree(e->d_store); (void)hipFree(e->d_ids);
    e->d_store = ns; e->d_ids = ni; e->capacity = new_cap;
    return WAX_HIP_OK;
}

// reserveIfNeeded (MetalVectorEngine.swift:857-871)
int reserve_rows(wax_hip_engine* e, uint64_t required) {
    if (required > 0xffffffffull)
        return fail(WAX_HIP_ERR_CAPACITY, "capacity exceeded: limit 4294967295, requested " + std::to_string(required));
    if (required <= e->capacity) return WAX_HIP_OK;
    uint64_t next = e->capacity == 0 ? WAX_HIP_INITIAL_RESERVE : e->capacity;
    while (required > next) {
        uint64_t doubled = next * 2;
        next = doubled > 0xffffffffull ? 0xffffffffull : doubled;
        if (next == 0xffffffffull) break;
    }
    return resize_store(e, next);
}

struct Enqueued { int k_eff; };

// The scan + select chain for one query on `stream`; leaves kpad hits in d_hits.
// Does a scan of this engine for k_eff results take its query through the kernel arguments ("query_args")? Decided BEFORE the
// query would be uploaded: the fused path only (the general selection reads the query through its pointer), the default kernel
// variant, dimensions scan_kernel_qarg exists for.
// Will a scan submitted now run in a stream of scans whose merge is better left to a second launch ("merge_overlap_mb")?
bool scan_overlaps_merge(wax_hip_engine* e, bool others_in_flight) {
    const int64_t mb = e->merge_overlap_mb.load();
    return others_in_flight && mb > 0 && e->count * (uint64_t)e->dims * sizeof(float) >= (uint64_t)mb << 20;
}

bool scan_uses_query_args(wax_hip_engine* e, int k_eff, bool has_general_slot, bool overlap_merge = false) {
    const int64_t mode = e->query_args.load();
    if (mode == 0 || !scan_query_args_dims(e->dims) || k_eff > FUSED_MAX_K) return false;
    if (e->force_general.load() && has_general_slot) return false;
    const int variant = (int)e->variant.load();
    if (variant > 0) return false;
    if (mode >= 2) return true;
    const int grid = scan_grid_for((uint32_t)e->count, e->dims, 0, (int)e->grid_blocks.load());
    if (grid <= SCAN_FUSE_MERGE_GRID) return true;
    // larger stores: where the scan is the query's only packet (it merges in its own kernel: k <= SCAN_KWAY_MAX_K, <= 2 GiB of rows)
    return e->fuse_merge.load() != 0 && !overlap_merge && scan_merges_in_kernel(grid, k_eff, e->merge_kway.load() != 0, (uint32_t)e->count, e->dims);
}

// d_query == nullptr: the query is `h_query` (host memory, read during this call) and travels in the kernel arguments.
int enqueue_scan(wax_hip_engine* e, const float* d_query, float q_norm, int k_eff, int kpad, int64_t* d_partials,
                 Slot* general_slot, wax_hip_hit* d_hits, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                 bool chain = false, hipEvent_t* used_start = nullptr, hipEvent_t* used_end = nullptr,
                 const float* h_query = nullptr, uint64_t* done_flag = nullptr, uint64_t done_value = 0, bool* out_flagged = nullptr,
                 bool overlap_merge = false, int tk_mode = 0) {
    // tk_mode: the caller's ONE read of "time_kernels" for this submit (a concurrent set_tuning between two reads could otherwise arm
    // ev0 / ev1 one way and mark the slot as timed the other: advisor, round 5)
    ScanArgs a{};
    a.store = e->d_store;
    a.query = d_query;
    a.query_host = d_query == nullptr ? h_query : nullptr;
    a.done_flag = done_flag;
    a.done_value = done_value;
    a.no_kway = e->merge_kway.load() != 0 ? 0 : 1;
    {
        // ordinary (cacheable) instead of non-temporal row loads: "scan_plain_mb" = stores of at most N MB (default 32); -1 = stores whose
        // grid fits the wave-list fused merge (<= 160 workgroups). Measured on one box (profiles/r04/n_latency_small_stores_*): ordinary loads
        // are 0.7-0.9 us per query faster at 5K / 10K / 20K rows x 384 (7.7 / 15 / 31 MB), equal from 60 to 230 MB, 10 % slower beyond.
        // (FETCH_SIZE still shows the whole store crossing the L2 -> fabric boundary on every query: the gain is on the memory side.)
        const int64_t mb = e->scan_plain_mb.load();
        a.plain_loads = mb < 0 ? (scan_grid_for((uint32_t)e->count, e->dims, 0, (int)e->grid_blocks.load()) <= SCAN_FUSE_MERGE_GRID)
                               : ((uint64_t)e->count * e->dims * sizeof(float) <= (uint64_t)mb << 20);
    }
    if (out_flagged) *out_flagged = false;
    a.partials = d_partials;
    a.dist_out = nullptr;
    a.n_rows = (uint32_t)e->count;
    a.row_base = (uint32_t)e->row_base;
    a.dims = e->dims;
    a.k = k_eff;
    a.q_norm = q_norm;
    const bool fused = k_eff <= FUSED_MAX_K && !(e->force_general.load() && general_slot != nullptr);
    int grid = 0;
    std::unique_lock<std::mutex> chain_guard(e->chain_mu, std::defer_lock);
    if (chain) {
        chain_guard.lock();
        if (e->scan_done_valid) HIP_TRY(hipStreamWaitEvent(stream, e->chain_event, 0), WAX_HIP_ERR_INTERNAL, "scan chain wait");
    }
    if (used_start) *used_start = ev0;
    if (used_end) *used_end = ev1;
    // "time_kernels" = 2: the event pair is bound to the scan's dispatch itself (kernels.h: launch_kernel) — no trailing marker and no
    // chain wait inside the interval; 1: the pair is recorded in front of and behind the launch.
    const bool bound = ev0 != nullptr && ev1 != nullptr && tk_mode == 2;
    if (fused) {
        const int cap = k_eff <= 64 ? 128 : 256;
        bool record_start = ev0 != nullptr && !bound;
        if (!bound && chain_guard.owns_lock() && ev0 && ev1 && used_start && used_end && e->share_timing.load() != 0) {
            hipEvent_t& slot_ev = e->tev[e->tev_next % wax_hip_engine::kTimingRing];
            if (!slot_ev && hipEventCreateWithFlags(&slot_ev, hipEventReleaseToDevice) != hipSuccess) slot_ev = nullptr;
            if (slot_ev) {
                ++e->tev_next;
                ev1 = slot_ev;
                // previous scan still in flight: this one starts when that one ends, and that moment is already recorded
                if (e->scan_done_valid && e->chain_is_timing && hipEventQuery(e->chain_event) == hipErrorNotReady) {
                    ev0 = e->chain_event;
                    record_start = false;
                }
                *used_start = ev0;
                *used_end = ev1;
            }
        }
        if (record_start) HIP_TRY(hipEventRecord(ev0, stream), WAX_HIP_ERR_INTERNAL, "event record");
        // small grids: the last-arriving workgroup does the final merge itself (one launch per query instead of two)
        bool merged = false;
        // (overlap_merge: a large store in a stream of scans — the merge launch behind this scan overlaps the next one)
        const bool small_grid = scan_grid_for((uint32_t)e->count, e->dims, 0, (int)e->grid_blocks.load()) <= SCAN_FUSE_MERGE_GRID;
        if (e->fuse_merge.load() != 0 && (!overlap_merge || small_grid)) { a.merge_out = d_hits; a.ids = e->d_ids; a.arrive = partials_ticket(d_partials); a.kpad = kpad; }
        else if (overlap_merge) e->st_overlap_scans++;
        if (bound) launch_timing() = LaunchTiming{ev0, ev1};
        const hipError_t lerr = launch_scan(a, e->metric, (int)e->variant.load(), cap, false, (int)e->grid_blocks.load(), stream, &grid, &merged);
        launch_timing() = LaunchTiming{};
        HIP_TRY(lerr, WAX_HIP_ERR_INTERNAL, "scan kernel launch");
        if (ev1 && !bound) HIP_TRY(hipEventRecord(ev1, stream), WAX_HIP_ERR_INTERNAL, "event record");
        if (chain_guard.owns_lock()) {
            // the next scan (on the other stream) starts when this one ends: its end-of-kernel timing event doubles
            // as the chain event when kernels are timed — every packet between two scans costs microseconds
            // (kernel-bound timing: the packets between two scans are outside the interval, so the chain has its own event)
            if (ev1 && !bound) {
                e->chain_event = ev1;
                e->chain_is_timing = true;
            } else {
                HIP_TRY(hipEventRecord(e->scan_done, stream), WAX_HIP_ERR_INTERNAL, "scan chain record");
                e->chain_event = e->scan_done;
                e->chain_is_timing = false;
            }
            e->scan_done_valid = true;
            chain_guard.unlock();
        }
        if (merged) e->st_merged_scans++;
        if (out_flagged) *out_flagged = merged && done_flag != nullptr;   // the kernel itself publishes the completion word
        if (!merged)
            HIP_TRY(launch_merge_keys(d_partials, (uint32_t)grid * (uint32_t)k_eff, k_eff, kpad, e->d_ids, a.row_base,
                                      a.n_rows, d_hits, cap, stream),
                    WAX_HIP_ERR_INTERNAL, "merge kernel launch");
    } else {
        if (!general_slot) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "top_k too large for the device-resident shard path (max 192)");
        int rc = ensure_general(e, general_slot);
        if (rc != WAX_HIP_OK) return rc;
        a.dist_out = general_slot->d_dist;
        if (ev0 && !bound) HIP_TRY(hipEventRecord(ev0, stream), WAX_HIP_ERR_INTERNAL, "event record");
        if (bound) launch_timing() = LaunchTiming{ev0, ev1};
        const hipError_t lerr = launch_scan(a, e->metric, 0, 128, true, (int)e->grid_blocks.load(), stream, &grid);
        launch_timing() = LaunchTiming{};
        HIP_TRY(lerr, WAX_HIP_ERR_INTERNAL, "distance kernel launch");
        if (ev1 && !bound) HIP_TRY(hipEventRecord(ev1, stream), WAX_HIP_ERR_INTERNAL, "event record");
        HIP_TRY(launch_select_general(general_slot->d_dist, a.n_rows, a.row_base, k_eff, kpad, e->d_ids,
                                      general_slot->sw, d_hits, stream),
                WAX_HIP_ERR_INTERNAL, "select kernel launch");
    }
    e->st_searches++;
    e->st_rows += e->count;
    e->st_bytes += e->count * (uint64_t)e->dims * 4ull;
    return WAX_HIP_OK;
}

int hits_to_results(uint8_t metric, const wax_hip_hit* hits, uint32_t n, uint64_t* out_ids, float* out_scores,
                    uint32_t capacity, uint32_t* out_count) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n && m < capacity; ++i) {
        if (hits[i].key == KEY_PAD) continue;                    // idx == UInt32.max (:597)
        const float d = key_distance(hits[i].key);
        if (!std::isfinite(d)) continue;                         // !distance.isFinite (:597)
        if (hits[i].frame_id == ID_PAD) continue;                // index >= frameIds.count (:599)
        out_ids[m] = hits[i].frame_id;
        // VectorMetric.score(fromDistance:) (VectorMetric.swift:32-43)
        out_scores[m] = (metric == WAX_HIP_METRIC_COSINE) ? (1.0f - d) : (-d);
        ++m;
    }
    *out_count = m;
    return WAX_HIP_OK;
}

// Fold a finished shard-path scan's event pair into the kernel-time statistics.
void harvest_ring_event(wax_hip_engine* e, int r) {
    std::unique_lock<std::mutex> sg(e->st_mu);
    if (!e->ring_ev_pending[r]) return;
    e->ring_ev_pending[r] = false;
    float ms = 0.f;
    if (e->ring_t0[r] && e->ring_t1[r] && hipEventSynchronize(e->ring_t1[r]) == hipSuccess &&
        hipEventElapsedTime(&ms, e->ring_t0[r], e->ring_t1[r]) == hipSuccess) {
        e->st_last_ms = ms; e->st_total_ms += ms; e->st_timed += 1;
    }
}


// ---------------------------------------------------------------------------
// Batched path: Q x D^T on the matrix cores (batch.hip). Everything here is called with the shared lock held.

// Rows the next ensure_mirror would convert (a planning figure: read without the mirror's mutex).
static double mirror_rows_to_convert(wax_hip_engine* e) {
    const BatchMirror& b = e->batch;
    if (b.mirror_valid.load(std::memory_order_acquire)) return 0.0;
    if (b.stale || b.d_cb == nullptr || b.mirror_cap < e->capacity) return (double)e->count;
    const uint64_t cnt = e->count;
    return (double)(cnt > b.rows ? cnt - b.rows : 0) + (double)b.dirty.size();
}

// Mutation hooks of the bf16 mirror and the id -> row table (exclusive engine lock held).
static void mirror_note_upsert(wax_hip_engine* e, uint64_t row) {
    BatchMirror& b = e->batch;
    b.mirror_valid = false;
    if (b.stale || row >= b.rows) return;                    // not mirrored yet: converted with the appended range
    if (b.dirty.size() >= kMirrorMaxDirty) { b.stale = true; b.dirty.clear(); return; }
    b.dirty.push_back((uint32_t)row);
}
static void mirror_note_append(wax_hip_engine* e) { e->batch.mirror_valid = false; e->idhash.valid = false; }
static void mirror_note_replaced(wax_hip_engine* e) {       // deserialize: every row is new
    e->batch.mirror_valid = false; e->batch.stale = true; e->batch.dirty.clear(); e->batch.rows = 0;
    e->idhash.valid = false; e->idhash.stale = true; e->idhash.rows = 0;
}
// remove(frameId:) moved store rows (idx, count) down by one (MetalVectorEngine.swift:431-438): the mirror's tail follows (half the
// bytes of the store's own move), and the listed dirty rows move with it. Called BEFORE count is decremented.
static int mirror_note_remove(wax_hip_engine* e, uint64_t idx) {
    BatchMirror& b = e->batch;
    e->idhash.valid = false; e->idhash.stale = true; e->idhash.rows = 0;     // every later row changed its number
    b.mirror_valid = false;
    if (b.stale || b.d_cb == nullptr || idx >= b.rows) return WAX_HIP_OK;
    if (b.ev_pending) { (void)hipEventSynchronize(b.ev_ready); b.ev_pending = false; }
    const uint64_t after = b.rows - 1 - idx;
    if (after > 0) {
        if (!e->d_bounce)
            HIP_TRY(hipMalloc(&e->d_bounce, kBounceBytes), WAX_HIP_ERR_ALLOC, "Failed to allocate bounce buffer");
        const uint64_t rb = (uint64_t)e->dims * sizeof(unsigned short);
        HIP_TRY(device_shift_down(b.d_cb, idx * rb, (idx + 1) * rb, after * rb, e->d_bounce, kBounceBytes, nullptr), WAX_HIP_ERR_INTERNAL, "mirror row shift");
        HIP_TRY(device_shift_down(b.d_vn2, idx * 4, (idx + 1) * 4, after * 4, e->d_bounce, kBounceBytes, nullptr), WAX_HIP_ERR_INTERNAL, "mirror norm shift");
        HIP_TRY(hipStreamSynchronize(nullptr), WAX_HIP_ERR_INTERNAL, "mirror shift sync");
    }
    b.rows -= 1;
    size_t w = 0;
    for (size_t i = 0; i < b.dirty.size(); ++i) {
        const uint32_t r = b.dirty[i];
        if (r == idx) continue;
        b.dirty[w++] = r > idx ? r - 1 : r;
    }
    b.dirty.resize(w);
    return WAX_HIP_OK;
}

// The bf16 mirror of the store (+ ||v||^2, max ||v||, max rounding error): allocated at the first batched search, then kept in step
// with the store incrementally — see BatchMirror. Concurrent batches serialise on the conversion's ENQUEUE only: nothing here waits
// for the device (the converting stream records an event that every other workspace's stream waits for).
int ensure_mirror(wax_hip_engine* e, hipStream_t st) {
    BatchMirror& b = e->batch;
    if (b.mirror_valid.load(std::memory_order_acquire) && b.mirror_cap >= e->capacity) {
        // converted, but perhaps still in flight on another workspace's stream
        std::unique_lock<std::mutex> g(b.mu);
        if (b.ev_pending) {
            if (hipEventQuery(b.ev_ready) == hipSuccess) b.ev_pending = false;
            else HIP_TRY(hipStreamWaitEvent(st, b.ev_ready, 0), WAX_HIP_ERR_INTERNAL, "mirror ready wait");
        }
        return WAX_HIP_OK;
    }
    std::unique_lock<std::mutex> g(b.mu);
    const uint32_t D = e->dims;
    if (!b.d_maxnorm) {
        HIP_TRY(hipMalloc(&b.d_maxnorm, 2 * sizeof(unsigned int)), WAX_HIP_ERR_ALLOC, "Failed to allocate batch scalars");
        HIP_TRY(hipMalloc(&b.d_dirty, kMirrorMaxDirty * sizeof(uint32_t)), WAX_HIP_ERR_ALLOC, "Failed to allocate batch scalars");
        HIP_TRY(hipEventCreateWithFlags(&b.ev_ready, hipEventDisableTiming), WAX_HIP_ERR_INTERNAL, "event create");
        b.stale = true;
    }
    if (b.ev_pending) HIP_TRY(hipStreamWaitEvent(st, b.ev_ready, 0), WAX_HIP_ERR_INTERNAL, "mirror ready wait");   // order behind the previous conversion
    if (b.mirror_cap < e->capacity) {
        // The store was reallocated (capacity doubles, MetalVectorEngine.swift:857-871). No batch can be reading the old mirror: a
        // smaller mirror is only possible after a mutation, which took the exclusive lock after every reader had finished. The
        // converted rows move over (a device copy of half the store's bytes, not a re-conversion from f32).
        unsigned short* ncb = nullptr; float* nvn = nullptr;
        // + slack rows: the filtering GEMM requests whole tiles without clamps — its last tile may run up to (tile rows - 1) rows
        // past the store, plus one row for the pad slots and the slack behind the tile image (never used: masked by the selection)
        HIP_TRY(hipMalloc(&ncb, ((size_t)e->capacity + BATCH_MIRROR_SLACK_ROWS) * D * sizeof(unsigned short)), WAX_HIP_ERR_ALLOC, "Failed to allocate bf16 mirror");
        if (hipMalloc(&nvn, (size_t)e->capacity * sizeof(float)) != hipSuccess) { (void)hipFree(ncb); return fail(WAX_HIP_ERR_ALLOC, "Failed to allocate row norms"); }
        hipError_t err = hipMemsetAsync(ncb + (size_t)e->capacity * D, 0, (size_t)BATCH_MIRROR_SLACK_ROWS * D * sizeof(unsigned short), st);
        if (err == hipSuccess && !b.stale && b.rows > 0 && b.d_cb) {
            err = hipMemcpyAsync(ncb, b.d_cb, (size_t)b.rows * D * sizeof(unsigned short), hipMemcpyDeviceToDevice, st);
            if (err == hipSuccess) err = hipMemcpyAsync(nvn, b.d_vn2, (size_t)b.rows * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (err == hipSuccess) err = hipStreamSynchronize(st);                     // the old buffers are freed next
        } else {
            b.stale = true;
        }
        if (err != hipSuccess) { (void)hipFree(ncb); (void)hipFree(nvn); return fail(WAX_HIP_ERR_INTERNAL, std::string("mirror move: ") + hipGetErrorString(err)); }
        (void)hipFree(b.d_cb); (void)hipFree(b.d_vn2);
        b.d_cb = ncb; b.d_vn2 = nvn; b.mirror_cap = e->capacity;
    }
    const uint64_t count = e->count;
    if (b.stale) {
        b.rows = 0; b.dirty.clear(); b.stale = false;
        HIP_TRY(hipMemsetAsync(b.d_maxnorm, 0, 2 * sizeof(unsigned int), st), WAX_HIP_ERR_INTERNAL, "batch memset");
    }
    if (b.rows > count) b.rows = count;                       // (cannot happen: removals move `rows` with them)
    const int normalize = e->metric == WAX_HIP_METRIC_COSINE ? 1 : 0;
    uint64_t converted = 0;
    if (b.rows < count) {                                     // appended since: rows [rows, count)
        const uint64_t n_new = count - b.rows;
        HIP_TRY(launch_mirror(e->d_store + b.rows * D, (uint32_t)n_new, (uint32_t)n_new, D, normalize, b.d_cb + b.rows * D, b.d_vn2 + b.rows,
                              b.d_maxnorm, st), WAX_HIP_ERR_INTERNAL, "mirror kernel launch");
        converted += n_new;
        b.rows = count;
    }
    if (!b.dirty.empty()) {                                   // overwritten since: the listed rows
        std::sort(b.dirty.begin(), b.dirty.end());
        b.dirty.erase(std::unique(b.dirty.begin(), b.dirty.end()), b.dirty.end());
        HIP_TRY(hipMemcpyAsync(b.d_dirty, b.dirty.data(), b.dirty.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "dirty row list upload");
        HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "dirty row list upload");   // (pageable source: the vector is cleared next; a rare path)
        HIP_TRY(launch_mirror_rows(e->d_store, b.d_dirty, (uint32_t)b.dirty.size(), D, normalize, b.d_cb, b.d_vn2, b.d_maxnorm, st),
                WAX_HIP_ERR_INTERNAL, "mirror kernel launch");
        converted += b.dirty.size();
        b.dirty.clear();
    }
    if (converted) {
        HIP_TRY(hipEventRecord(b.ev_ready, st), WAX_HIP_ERR_INTERNAL, "mirror ready record");
        b.ev_pending = true;
        b.rows_converted += converted;
        b.conversions += 1;
    }
    b.mirror_valid.store(true, std::memory_order_release);
    return WAX_HIP_OK;
}

void free_bctx(BatchCtx* c) {
    if (!c) return;
    (void)hipFree(c->d_q); (void)hipFree(c->d_qb); (void)hipFree(c->d_qf); (void)hipFree(c->d_qn2); (void)hipFree(c->d_qnorm); (void)hipFree(c->d_eps);
    (void)hipFree(c->d_tau); (void)hipFree(c->d_dense); (void)hipFree(c->d_cand_count); (void)hipFree(c->d_overflow);
    (void)hipFree(c->d_cand); (void)hipFree(c->d_seg_count); (void)hipFree(c->d_exact); (void)hipFree(c->d_sel);
    (void)hipFree(c->d_tile_max); (void)hipFree(c->d_hits);
    (void)hipHostFree(c->h_hits); (void)hipHostFree(c->h_cert); (void)hipFree(c->d_cert); (void)hipHostFree(c->h_qnorm); (void)hipHostFree(c->h_qlist);
    (void)hipFree(c->d_qlist); (void)hipHostFree(c->h_fnorm); (void)hipFree(c->d_fnorm); (void)hipFree(c->d_mpart); (void)hipFree(c->d_rescue); (void)hipFree(c->d_rescue_exact); (void)hipHostFree(c->h_live);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_g0) (void)hipEventDestroy(c->ev_g0);
    if (c->ev_g1) (void)hipEventDestroy(c->ev_g1);
    if (c->ev_gc) (void)hipEventDestroy(c->ev_gc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int alloc_bctx(wax_hip_engine* e, BatchCtx** out) {
    BatchCtx* c = new BatchCtx();
    const uint32_t D = e->dims;
    (void)D;
    hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    auto A = [&](auto** p, size_t bytes) { if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(p), bytes); };
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->ev_g0, hipEventReleaseToDevice);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->ev_g1, hipEventReleaseToDevice);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->ev_gc, hipEventDisableTiming);
    A(&c->d_qb, (size_t)kBatchMaxQ * D * sizeof(unsigned short));
    A(&c->d_qf, (size_t)kBatchMaxQ * D * sizeof(unsigned short));
    A(&c->d_qn2, kBatchMaxQ * sizeof(float));
    A(&c->d_qnorm, kBatchMaxQ * sizeof(float));
    A(&c->d_eps, kBatchMaxQ * sizeof(float));
    A(&c->d_tau, kBatchMaxQ * sizeof(float));
    A(&c->d_cand_count, (size_t)kBatchMaxQ * CAND_COUNT_STRIDE * sizeof(uint32_t));
    A(&c->d_overflow, (kBatchMaxQ + BATCH_PROGRESS_WORDS) * sizeof(uint32_t));   // + the filtering GEMM's pace-gate words
    A(&c->d_seg_count, (size_t)kBatchMaxSegs * kBatchMaxQ * sizeof(uint32_t));
    A(&c->d_qlist, (size_t)kBatchMaxQ * sizeof(uint32_t));
    A(&c->d_fnorm, (size_t)kBatchMaxQ * sizeof(float));
    if (err == hipSuccess) err = hipHostMalloc(&c->h_qlist, (size_t)kBatchMaxQ * sizeof(uint32_t), hipHostMallocDefault);
    if (err == hipSuccess) err = hipHostMalloc(&c->h_fnorm, (size_t)kBatchMaxQ * sizeof(float), hipHostMallocDefault);
    if (err == hipSuccess) err = hipHostMalloc(&c->h_live, (size_t)kBatchMaxQ * sizeof(uint32_t), hipHostMallocDefault);
    if (err != hipSuccess) {
        free_bctx(c);
        return fail(WAX_HIP_ERR_ALLOC, std::string("Failed to allocate batch workspace: ") + hipGetErrorString(err));
    }
    *out = c;
    return WAX_HIP_OK;
}

int acquire_bctx(wax_hip_engine* e, BatchCtx** out, bool try_only = false) {
    std::unique_lock<std::mutex> g(e->bctx_mu);
    for (;;) {
        if (!e->bctx_free.empty()) {
            *out = e->bctx_free.back();
            e->bctx_free.pop_back();
            return WAX_HIP_OK;
        }
        if ((int)e->bctx_all.size() < e->bctx_max) {
            BatchCtx* c = nullptr;
            int rc = alloc_bctx(e, &c);
            if (rc != WAX_HIP_OK) return rc;
            e->bctx_all.push_back(c);
            *out = c;
            return WAX_HIP_OK;
        }
        if (try_only)
            return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "every batch workspace is in use: collect outstanding batch tickets first (or raise \"batch_workspaces\")");
        e->bctx_cv.wait(g);
    }
}

void release_bctx(wax_hip_engine* e, BatchCtx* c) {
    std::unique_lock<std::mutex> g(e->bctx_mu);
    e->bctx_free.push_back(c);
    e->bctx_cv.notify_one();
}

// Grow-on-demand buffers of a workspace (idle on its stream when called: a call owns its workspace and starts here).
template <typename T>
int grow_dev(T** p, uint64_t* cap, uint64_t want, size_t elem, const char* what) {
    if (*cap >= want) return WAX_HIP_OK;
    (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), (size_t)want * elem), WAX_HIP_ERR_ALLOC, what);
    *cap = want;
    return WAX_HIP_OK;
}

// ---- filtered-search workspaces and the id -> row table ----

void free_filter_work(FilterWork* f) {
    if (!f) return;
    (void)hipFree(f->d_rows); (void)hipFree(f->d_ids); (void)hipFree(f->d_dist); (void)hipFree(f->d_allow); (void)hipFree(f->d_bitmap);
    (void)hipFree(f->d_block_sum); (void)hipFree(f->d_total); (void)hipFree(f->d_query); (void)hipFree(f->d_qnorm); (void)hipFree(f->d_hits);
    free_select_work(&f->sw);
    if (f->h_total) (void)hipHostFree(f->h_total);
    if (f->h_hits) (void)hipHostFree(f->h_hits);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

static int alloc_filter_work(wax_hip_engine* e, FilterWork** out) {
    FilterWork* f = new FilterWork();
    struct Guard { FilterWork* f; ~Guard() { if (f) free_filter_work(f); } } guard{f};
    HIP_TRY(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking), WAX_HIP_ERR_INTERNAL, "Failed to create filter stream");
    HIP_TRY(hipMalloc(&f->d_query, (size_t)e->dims * sizeof(float)), WAX_HIP_ERR_ALLOC, "Failed to allocate filter query buffer");
    HIP_TRY(hipMalloc(&f->d_qnorm, sizeof(float)), WAX_HIP_ERR_ALLOC, "Failed to allocate filter scalars");
    HIP_TRY(hipMalloc(&f->d_total, sizeof(uint32_t)), WAX_HIP_ERR_ALLOC, "Failed to allocate filter scalars");
    HIP_TRY(hipHostMalloc(&f->h_total, sizeof(uint32_t), hipHostMallocDefault), WAX_HIP_ERR_ALLOC, "Failed to allocate filter scalars");
    HIP_TRY(hipMalloc(&f->d_hits, (size_t)WAX_HIP_MAX_RESULTS * sizeof(wax_hip_hit)), WAX_HIP_ERR_ALLOC, "Failed to allocate filter hits");
    HIP_TRY(hipHostMalloc(&f->h_hits, (size_t)WAX_HIP_MAX_RESULTS * sizeof(wax_hip_hit), hipHostMallocDefault), WAX_HIP_ERR_ALLOC,
            "Failed to allocate filter hits staging");
    HIP_TRY(alloc_select_work(&f->sw), WAX_HIP_ERR_ALLOC, "Failed to allocate the selection workspace");
    guard.f = nullptr;
    *out = f;
    return WAX_HIP_OK;
}

static int acquire_filter_work(wax_hip_engine* e, FilterWork** out) {
    std::unique_lock<std::mutex> g(e->filter_mu);
    for (;;) {
        if (!e->filter_free.empty()) {
            *out = e->filter_free.back();
            e->filter_free.pop_back();
            return WAX_HIP_OK;
        }
        if ((int)e->filter_all.size() < e->filter_max) {
            FilterWork* f = nullptr;
            int rc = alloc_filter_work(e, &f);
            if (rc != WAX_HIP_OK) return rc;
            e->filter_all.push_back(f);
            *out = f;
            return WAX_HIP_OK;
        }
        e->filter_cv.wait(g);
    }
}

static void release_filter_work(wax_hip_engine* e, FilterWork* f) {
    std::unique_lock<std::mutex> g(e->filter_mu);
    e->filter_free.push_back(f);
    e->filter_cv.notify_one();
}

// The id -> row table of the store as it is now (shared lock held; pending rows flushed). Concurrent filtered searches
// serialise on the rebuild only.
static int ensure_idhash(wax_hip_engine* e, hipStream_t st) {
    IdHash& h = e->idhash;
    if (h.valid.load(std::memory_order_acquire)) return WAX_HIP_OK;
    std::unique_lock<std::mutex> g(h.mu);
    if (h.valid.load(std::memory_order_acquire)) return WAX_HIP_OK;
    uint64_t want = 1024;
    while (want < 2 * e->count) want *= 2;   // load factor <= 0.5
    if (h.slots < want) {
        (void)hipFree(h.d_table);
        h.d_table = nullptr; h.slots = 0;
        want *= 2;                                // room for the appends to come: a table that must grow is rebuilt from row 0
        HIP_TRY(hipMalloc(&h.d_table, (size_t)want * sizeof(uint32_t)), WAX_HIP_ERR_ALLOC, "Failed to allocate id table");
        h.slots = want;
        h.stale = true;
    }
    if (h.stale) { h.rows = 0; h.stale = false; }
    if (h.rows > e->count) { h.rows = 0; }        // (cannot happen: removals start the table over)
    // rows [h.rows, count): all of them after a removal / deserialize / growth (the table is cleared first), the appended ones otherwise
    HIP_TRY(launch_idhash_build(e->d_ids, (uint32_t)h.rows, (uint32_t)e->count, h.d_table, h.slots, st), WAX_HIP_ERR_INTERNAL, "id table kernel launch");
    HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "id table build failed on device");
    e->st_idhash_rows += e->count - h.rows;
    h.rows = e->count;
    h.valid.store(true, std::memory_order_release);
    return WAX_HIP_OK;
}


// [kBatchMaxQ][kp] scratch of the re-score / large-k' selection (the workspace's stream is idle when this is called)
int bctx_reserve_kp(BatchCtx* c, uint64_t kp) {
    if (c->kp_cap >= kp) return WAX_HIP_OK;
    (void)hipFree(c->d_exact); (void)hipFree(c->d_sel);
    c->d_exact = nullptr; c->d_sel = nullptr; c->kp_cap = 0;
    HIP_TRY(hipMalloc(&c->d_exact, (size_t)kBatchMaxQ * kp * sizeof(int64_t)), WAX_HIP_ERR_ALLOC, "Failed to allocate batch re-score keys");
    HIP_TRY(hipMalloc(&c->d_sel, (size_t)kBatchMaxQ * kp * sizeof(int64_t)), WAX_HIP_ERR_ALLOC, "Failed to allocate batch selection");
    c->kp_cap = kp;
    return WAX_HIP_OK;
}

int bctx_reserve(BatchCtx* c, uint64_t cand_slots, uint64_t kp, uint64_t tile_rows, uint64_t n_queries, bool dense) {
    // [rows of the largest block][cand_slots] keys; capacity tracked in keys
    const uint64_t rows_blk = n_queries < kBatchMaxQ ? ((n_queries + 255ull) & ~255ull) : (uint64_t)kBatchMaxQ;
    int rc = grow_dev(&c->d_cand, &c->cand_slots, rows_blk * cand_slots, sizeof(int64_t), "Failed to allocate batch candidates");
    if (rc != WAX_HIP_OK) return rc;
    rc = bctx_reserve_kp(c, kp);
    if (rc != WAX_HIP_OK) return rc;
    rc = grow_dev(&c->d_tile_max, &c->tile_max_rows, tile_rows, (size_t)kBatchMaxQ * sizeof(float), "Failed to allocate sample maxima");
    if (rc != WAX_HIP_OK) return rc;
    if (dense && !c->d_dense)
        HIP_TRY(hipMalloc(&c->d_dense, (size_t)kBatchMaxQ * kBatchFirstSlab * sizeof(float)), WAX_HIP_ERR_ALLOC, "Failed to allocate first-slab tile");
    if (c->cert_cap < n_queries) {
        uint64_t want = 1024;
        while (want < n_queries) want *= 2;
        (void)hipHostFree(c->h_cert); (void)hipHostFree(c->h_qnorm); (void)hipFree(c->d_cert);
        c->h_cert = nullptr; c->h_qnorm = nullptr; c->d_cert = nullptr; c->cert_cap = 0;
        HIP_TRY(hipHostMalloc(&c->h_cert, want * sizeof(uint32_t), hipHostMallocDefault), WAX_HIP_ERR_ALLOC, "Failed to allocate pinned batch flags");
        HIP_TRY(hipMalloc(&c->d_cert, want * sizeof(uint32_t)), WAX_HIP_ERR_ALLOC, "Failed to allocate batch flags");
        HIP_TRY(hipHostMalloc(&c->h_qnorm, want * sizeof(float), hipHostMallocDefault), WAX_HIP_ERR_ALLOC, "Failed to allocate pinned batch norms");
        c->cert_cap = want;
    }
    return WAX_HIP_OK;
}

// Staging of a host-pointer call: the query block in HBM and the hits on their way back.
int bctx_reserve_host(wax_hip_engine* e, BatchCtx* c, uint64_t nq, uint64_t hits) {
    int rc = grow_dev(&c->d_q, &c->q_cap, nq, (size_t)e->dims * sizeof(float), "Failed to allocate batch queries");
    if (rc != WAX_HIP_OK) return rc;
    if (c->hits_cap < hits) {
        uint64_t want = (uint64_t)kBatchMaxQ * 16;
        while (want < hits) want *= 2;
        (void)hipFree(c->d_hits); (void)hipHostFree(c->h_hits);
        c->d_hits = nullptr; c->h_hits = nullptr; c->hits_cap = 0;
        HIP_TRY(hipMalloc(&c->d_hits, want * sizeof(wax_hip_hit)), WAX_HIP_ERR_ALLOC, "Failed to allocate batch hits");
        HIP_TRY(hipHostMalloc(&c->h_hits, want * sizeof(wax_hip_hit), hipHostMallocDefault), WAX_HIP_ERR_ALLOC, "Failed to allocate pinned batch hits");
        c->hits_cap = want;
    }
    return WAX_HIP_OK;
}

int batch_kp(int k_eff, int kp_max) {
    int kp = 2 * k_eff + 32;
    if (kp < 64) kp = 64;
    if (kp > kp_max) kp = kp_max;
    return kp;
}

// ---- one-pass pipeline: plan -------------------------------------------------------------------------------------
struct OnepassPlan {
    uint32_t tile_rows, ntiles, sample_tiles, rank, seg_area;
    int kp;
    double expect;       // expected survivors per query
};

// tau_sim = the j-th largest of the S sampled tiles' best similarities (pick_tau_kernel). With well-mixed rows the
// number of store rows above it is Gamma(j) / f, f = S * tile_rows / n the sampled fraction: mean j / f, relative spread
// 1 / sqrt(j). The filter fails a query (exact path, ~0.25 ms) when fewer than m ~ 2.2 k rows pass — k for the answer
// plus the rows within the bf16 error band of the k-th — i.e. with probability P(Gamma(j) < m f). The plan takes the
// rank and sample size of least modelled cost among those whose failure probability is below 1e-6 per query (and whose
// expected survivors are at least `batch_survivors` x k', a floor for stores that are not well mixed): survivors cost the filtering GEMM its cold path
// (~10 % of the kernel at 10 k' survivors) and the finish kernel its gather. The minimum of G = 4 group maxima used
// before needed ~10 k' survivors for 1e-5. Sampling costs a tile round (~3 us) per `workgroups per group` tiles and
// visits scattered tiles, so S is held to a few rounds.
constexpr uint32_t kPickJ = 12;      // == PICK_J (batch.hip)
double gamma_cdf_below(uint32_t j, double x) {   // P(Gamma(j, 1) < x) = P(Poisson(x) >= j)
    double term = std::exp(-x), sum = 0.0;
    for (uint32_t i = 1; i <= j + 60; ++i) {
        term *= x / (double)i;               // e^-x x^i / i!
        if (i >= j) sum += term;
    }
    return sum;
}
bool plan_onepass(wax_hip_engine* e, uint32_t n, int k_eff, uint32_t nq, OnepassPlan* p) {
    if (e->batch_onepass.load() == 0 || !batch_onepass_dims(e->dims, e->metric) || k_eff > kBatchMaxK) return false;
    // fast: a register-resident GEMM filters (survivors in per-workgroup segments); otherwise (L2, other multiples of 64)
    // the LDS-tiled kernel does, appending to ONE counted list per query
    const bool fast = batch_onepass_fast(e->dims, e->metric);
    p->tile_rows = batch_tile_rows(e->dims, e->metric);
    p->ntiles = (n + p->tile_rows - 1) / p->tile_rows;
    // too small to sample: slab pipeline ("batch_onepass_tiles" counts units of min(tile rows, 64) rows: 32 at D = 768, 64 elsewhere)
    const uint32_t unit = fast ? std::min<uint32_t>(p->tile_rows, 64u) : 64u;
    if ((int64_t)n < e->batch_onepass_tiles.load() * (int64_t)unit || (fast ? n < 1024u * unit : p->ntiles < 512)) return false;
    p->kp = batch_kp(k_eff, 960);
    // k in 81 .. 128: k' = 2k + 32 would need the three-launch finish (select 120-190 us + re-score + finalize at 256 queries on a dense
    // corpus) where the fused finish kernel takes ~58 us. With the device-side retry behind it — which re-scores exactly the survivors the
    // first finish's k-th cannot exclude, whatever k' was — k' = 192 (>= 64 candidates beyond k) loses nothing but a few more retried queries.
    if (e->batch_kp_fused.load() != 0 && p->kp > FUSED_MAX_K && k_eff <= 128 && e->batch_retry.load() == 1 && batch_retry_dims(e->dims))
        p->kp = FUSED_MAX_K;
    const uint32_t nq_blk = nq < kBatchMaxQ ? nq : kBatchMaxQ;
    const uint32_t nq_pad = (nq_blk + 255u) & ~255u;
    const uint32_t groups = nq_pad / BATCH_GROUP_QUERIES;
    uint32_t nseg = fast ? 256 / (groups ? groups : 1) : 1u;   // workgroups per query group == survivor segments per query (1 = counted list)
    if (nseg < 1) nseg = 1;
    if (nseg > p->ntiles) nseg = p->ntiles;
    // sampled tiles that run at once (one "round" ~ 3 us): the persistent workgroups of a query group, or — LDS-tiled kernel,
    // one workgroup per (tile, 128 queries), ~3 resident per CU — 768 workgroups over the batch's query tiles
    const uint32_t samp_par = fast ? nseg : std::max<uint32_t>(1u, 768u / (nq_pad / 128u));
    const double floor_e = (double)e->batch_survivors.load() * (double)p->kp;
    uint64_t s_pref = p->ntiles / (uint64_t)e->batch_sample_div.load();
    if (s_pref < 192) s_pref = 192;
    if (s_pref > 2048) s_pref = 2048;
    if (s_pref > 8ull * samp_par) s_pref = 8ull * samp_par > 192 ? 8ull * samp_par : 192;   // at most 8 tile rounds
    if (s_pref > p->ntiles / 2) s_pref = p->ntiles / 2;
    const double rows = (double)p->tile_rows, nn = (double)n;
    const double need = 2.2 * (double)k_eff;             // rows that must pass: k + those inside the bf16 error band of the k-th
    // Per rank j: the expected survivors E that keep P(Gamma(j) < need * j / E) below 1e-6 (a fallback costs ~0.25 ms: 1e-6 x 1024 queries = 0.3 us per batch), the sample size that gives
    // that E — in general E(j, S) = n (1 - (1 - j/S)^(1/tile_rows)): a fraction j/S of the tiles holds a row above the
    // threshold; ~ j n / (S tile_rows) while j << S, and still a valid (looser) threshold when a small store or a large
    // k forces j close to S — and a cost in microseconds: sampling rounds + cold-path survivors (profiles/r02).
    double best_cost = 0.0;
    p->rank = 0;
    // x_j: P(Gamma(j) < x_j) = 1e-6 — constants (40 bisection steps of a 70-term series each, for eight ranks: ~35 us of host
    // time that used to sit in front of the first launch of EVERY batch; a blocking call paid it in full). Computed once.
    static const std::array<double, kPickJ + 1> kGammaX = [] {
        std::array<double, kPickJ + 1> x{};
        for (uint32_t j = 5; j <= kPickJ; ++j) {
            double lo = 0.0, hi = (double)j;
            for (int it = 0; it < 40; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (gamma_cdf_below(j, mid) > 1e-6) hi = mid; else lo = mid;
            }
            x[j] = lo;
        }
        return x;
    }();
    for (uint32_t j = 5; j <= kPickJ; ++j) {
        const double lo = kGammaX[j];
        double e_need = need * (double)j / lo;
        if (e_need < floor_e) e_need = floor_e;
        if (e_need / nn > 0.25) continue;                 // a quarter of the store as candidates: not a filter any more
        double s_exact = (double)j / (1.0 - std::pow(1.0 - e_need / nn, rows));
        double st = std::floor(s_exact);
        if (st > (double)s_pref) st = (double)s_pref;
        if (st < (double)j + 1.0) st = (double)j + 1.0;
        const double expect = nn * (1.0 - std::pow(1.0 - (double)j / st, 1.0 / rows));
        if (expect / nn > 0.25) continue;
        const double cost = std::ceil(st / (double)samp_par) * 3.0 + expect * 0.031 * 256.0 / (double)(fast ? nseg : 256u);
        if (p->rank == 0 || cost < best_cost) {
            best_cost = cost;
            p->rank = j;
            p->sample_tiles = (uint32_t)st;
            p->expect = expect;
        }
    }
    if (p->rank == 0) return false;
    if (p->expect < 2.0 * (double)k_eff + 8.0) return false;   // cannot reach enough candidates for the certificate: other paths
    // Survivor segments: one per (GEMM workgroup of the query's group, query). A query's survivor count spreads around
    // `expect` like Gamma(rank) (x 2.3 at 1e-4 for rank 12; clustered stores spread more), and a segment overflows when
    // ITS fill does: size every segment for 4 x expect spread over the group's workgroups, + 6 sigma of a Poisson fill
    // (an overflow only costs that query the exact path, but at 0.25 ms each two of them per batch doubled the batch
    // time: profiles/r02/c_onepass_diag.txt).
    // ... and at least 128 slots per segment (8 192 for the one counted list): memory is not the constraint (32 K keys per
    // query = 64 MB for a 256-query block), and a run of near-duplicate rows — consecutive chunks of one document — lands in
    // the few segments whose workgroups own those tiles: 64 survivors from one 64-row tile, not the ~1 of well-mixed rows.
    // Roomy segments are what lets the full retry (re-score every survivor) answer such queries without a pass over the store.
    const double fill = 4.0 * p->expect / (double)nseg;
    uint64_t slots = fast ? 128 : 8192;
    while ((double)slots < fill + 6.0 * std::sqrt(fill) + 4.0) slots *= 2;
    const uint64_t area = slots * nseg;
    if (area > 262144) return false;
    p->seg_area = (uint32_t)area;
    return true;
}

// Enqueue the whole batch pipeline for <= kBatchMaxQ device-resident queries on the workspace's stream: hits land in
// d_out[q * out_stride .. + out_stride) (k_eff real entries, the rest padded), certificate flags in ctx->h_cert +
// cert_off, exact norms in ctx->h_qnorm + cert_off (pinned host memory the kernels write directly). Nothing is synchronised here.
int batch_enqueue(wax_hip_engine* e, BatchCtx* c, const float* d_queries, uint32_t qn, int k_eff, const OnepassPlan* plan,
                  wax_hip_hit* d_out, uint32_t out_stride, uint32_t cert_off) {
    BatchMirror& b = e->batch;
    hipStream_t st = c->stream;
    const uint32_t D = e->dims;
    const uint32_t n = (uint32_t)e->count;
    const uint32_t nq_pad = (qn + 255u) & ~255u;  // 256: the register-resident-queries GEMM works on groups of 256
    PrepArgs pa{};
    pa.queries = d_queries; pa.nq = qn; pa.nq_pad = nq_pad; pa.dims = D; pa.metric = e->metric;
    pa.max_bits = b.d_maxnorm;                                   // {max ||v||, max ||x - bf16(x)||} as the mirror's conversions left them, read on the device
    pa.use_measured = e->batch_eps_measured.load() != 0 ? 1 : 0;
    const bool frag_order = e->batch_qfrag.load() != 0 && (D % 16u) == 0;   // "batch_qfrag" (default 1): fragment-ordered query copy for the rq GEMM
    pa.qf = frag_order ? c->d_qf : nullptr;
    pa.qb = c->d_qb; pa.q_n2 = c->d_qn2; pa.q_norm = c->d_qnorm; pa.eps = c->d_eps; pa.tau = c->d_tau; pa.overflow = c->d_overflow;
    const bool counted = plan != nullptr && !batch_onepass_fast(D, e->metric);   // one-pass on the LDS-tiled kernel: one counted list per query
    pa.cand_count = (plan && !counted) ? nullptr : c->d_cand_count;
    pa.q_norm_host = c->h_qnorm + cert_off;
    // the words behind the overflow flags: the progress words of the 768-d filtering GEMM's pace gate — zeroed by the prep kernel
    const bool pace_gate = plan != nullptr && D == 768 && nq_pad > 256;
    pa.progress = pace_gate ? c->d_overflow + kBatchMaxQ : nullptr;
    HIP_TRY(launch_batch_prep(pa, st), WAX_HIP_ERR_INTERNAL, "batch prep launch");
    GemmArgs g{};
    g.qf = frag_order ? c->d_qf : nullptr;
    g.qb = c->d_qb; g.cb = b.d_cb; g.q_n2 = c->d_qn2; g.v_n2 = b.d_vn2; g.tau = c->d_tau;
    g.cand = c->d_cand; g.cand_count = c->d_cand_count; g.row_base = (uint32_t)e->row_base;
    g.dims = D; g.n_rows = n; g.nq = qn; g.nqt = nq_pad / 128;
    g.use_rega = (uint32_t)e->batch_rega.load();  // 0 LDS-tiled kernel, 1 workgroup barrier per tile, 5 split barrier (default)
    g.debug = (uint32_t)e->batch_debug.load();
    g.prof = reinterpret_cast<uint32_t*>((uintptr_t)e->batch_prof_ptr.load());
    g.seg_count = c->d_seg_count;
    if (plan) {
        // ---- one pass: sample -> threshold -> filter everything -> finish ----
        g.slab0 = 0; g.slab_rows = n; g.dense = nullptr; g.dense_ld = 0;
        g.cand_cap = plan->seg_area; g.seg_base = 0; g.seg_area = plan->seg_area;
        if (g.use_rega == 0) g.use_rega = 5;
        GemmArgs gs = g;
        gs.tile_max = c->d_tile_max; gs.sample_tiles = plan->sample_tiles;
        g.progress = pa.progress;
        HIP_TRY(launch_batch_gemm_sample(gs, e->metric, st), WAX_HIP_ERR_INTERNAL, "sampling gemm launch");
        const int tk_mode = (int)e->time_kernels.load();   // read once per batch
        const bool timed = tk_mode != 0;
        std::unique_lock<std::mutex> cg(e->gemm_chain_mu, std::defer_lock);
        if (timed) {
            // (the chain wait sits in front of the threshold kernel: the sampling GEMM before it could not get a CU
            // until the previous batch's filtering GEMM left anyway, and one packet less separates threshold and GEMM)
            cg.lock();
            if (e->gemm_chain_ev && e->gemm_chain_ev != c->ev_g1 && e->gemm_chain_ev != c->ev_gc)
                HIP_TRY(hipStreamWaitEvent(st, e->gemm_chain_ev, 0), WAX_HIP_ERR_INTERNAL, "gemm chain wait");
        }
        HIP_TRY(launch_pick_tau(c->d_tile_max, plan->sample_tiles, qn, nq_pad, plan->rank, c->d_tau, e->metric, st), WAX_HIP_ERR_INTERNAL,
                "threshold kernel launch");
        if (timed) {
            // With several batches in flight (submit / collect) the filtering GEMMs of different workspaces would queue
            // for the same CUs (one workgroup's LDS fills a CU) and an event interval would include that wait: chain
            // them, like the single-query scans, so that a timed interval is one GEMM running alone. The small
            // kernels before this point (prep, sampling, thresholds) still overlap the previous batch's GEMM tail
            // and finish kernel.
            if (tk_mode == 2) {
                // kernel-bound pair (kernels.h: launch_kernel)
                launch_timing() = LaunchTiming{c->ev_g0, c->ev_g1};
                const hipError_t lerr = launch_batch_gemm(g, e->metric, st);
                launch_timing() = LaunchTiming{};
                HIP_TRY(lerr, WAX_HIP_ERR_INTERNAL, "gemm kernel launch");
                HIP_TRY(hipEventRecord(c->ev_gc, st), WAX_HIP_ERR_INTERNAL, "event record");
                e->gemm_chain_ev = c->ev_gc;
            } else {
                HIP_TRY(hipEventRecord(c->ev_g0, st), WAX_HIP_ERR_INTERNAL, "event record");
                HIP_TRY(launch_batch_gemm(g, e->metric, st), WAX_HIP_ERR_INTERNAL, "gemm kernel launch");
                HIP_TRY(hipEventRecord(c->ev_g1, st), WAX_HIP_ERR_INTERNAL, "event record");
                e->gemm_chain_ev = c->ev_g1;
            }
            c->gemm_timed = true; c->gemm_rows = n; c->gemm_queries = qn;
        } else {
            HIP_TRY(launch_batch_gemm(g, e->metric, st), WAX_HIP_ERR_INTERNAL, "gemm kernel launch");
        }
        FinishArgs f{};
        f.cand = c->d_cand; f.cand_cap = plan->seg_area; f.seg_count = c->d_seg_count; f.nq_pad = nq_pad;
        if (counted) {
            f.nseg = 1; f.seg_slots = plan->seg_area; f.seg_count = c->d_cand_count; f.count_stride = CAND_COUNT_STRIDE;
        } else if (!batch_gemm_segments(g, e->metric, &f.nseg, &f.seg_slots)) {
            return fail(WAX_HIP_ERR_INTERNAL, "one-pass plan without a segment kernel");
        }
        f.tau = c->d_tau; f.overflow = c->d_overflow; f.store = e->d_store; f.queries = d_queries; f.q_norm = c->d_qnorm;
        f.eps = c->d_eps; f.ids = e->d_ids; f.n_rows = n; f.row_base = (uint32_t)e->row_base; f.dims = D; f.nq = qn;
        f.kp = plan->kp; f.k = k_eff; f.sel = c->d_sel; f.exact = c->d_exact; f.out = d_out; f.out_stride = out_stride;
        f.certified = c->h_cert + cert_off;   // pinned host memory, written by the kernel: no copy launch behind the finish kernel
        f.cert_dev = c->d_cert + cert_off;
        f.retry_all = (e->batch_debug.load() & 65536) != 0 ? 1u : 0u;
        HIP_TRY(launch_batch_finish(f, e->metric, st), WAX_HIP_ERR_INTERNAL, "finish kernel launch");
        // "batch_retry" 1 (default): while recent batches had queries the first finish could not certify (`retry_hint`, set at
        // collect), the device-side full retry rides behind the finish kernel — uncertified queries get ALL their survivors
        // re-scored without a host round trip; certified ones cost their workgroup one flag read. A store whose batches certify
        // never pays for the launch. 2: the host-driven full retry of round 3 only; 0: neither (exact path at once).
        if (e->batch_retry.load() == 1 && e->retry_hint.load() > 0 && k_eff <= FUSED_MAX_K && batch_retry_dims(D) && !counted)
            HIP_TRY(launch_batch_retry(f, e->metric, st), WAX_HIP_ERR_INTERNAL, "retry kernel launch");
        c->last_finish = f; c->last_metric = e->metric;
        c->last_finish_valid = cert_off == 0;   // the candidate segments survive until collect only for a one-block batch
        e->st_onepass_queries += qn;
    } else {
        // ---- slab pipeline (small stores, L2, other dims): thresholds tightened between geometrically growing slabs ----
        c->last_finish_valid = false;
        const int kp = batch_kp(k_eff, FUSED_MAX_K);
        uint64_t max_slab = (uint64_t)e->batch_slab_mb.load() * 16384ull;  // "slab_mb" MB of f32 scores per 256 queries
        if (max_slab < 2048) max_slab = 2048;
        g.cand_cap = kBatchCandCap; g.seg_base = kBatchSegBase; g.seg_area = kBatchSegArea; g.dense_ld = kBatchFirstSlab;
        // Slabs grow geometrically: with tau tightened after each slab, a slab of s rows appends about
        // kp * s / rows_seen candidates per query, so "next slab = growth x rows seen" keeps every list ~growth*kp long.
        uint32_t s0 = 0;
        while (s0 < n) {
            uint64_t want = (s0 == 0) ? (uint64_t)e->batch_first.load() : (uint64_t)e->batch_growth.load() * s0;
            if (want > max_slab) want = max_slab;
            want = (want + 127ull) & ~127ull;
            const uint32_t rows = (n - s0 < want) ? n - s0 : (uint32_t)want;
            const bool first = (s0 == 0);  // no threshold yet: dense tile instead of appending everything
            g.slab0 = s0; g.slab_rows = rows; g.dense = first ? c->d_dense : nullptr;
            HIP_TRY(launch_batch_gemm(g, e->metric, st), WAX_HIP_ERR_INTERNAL, "gemm kernel launch");
            TightenArgs t{};
            t.cand = c->d_cand; t.cand_cap = kBatchCandCap; t.cand_count = c->d_cand_count; t.kp = kp; t.nq = qn;
            t.tau = c->d_tau; t.overflow = c->d_overflow;
            t.dense = first ? c->d_dense : nullptr; t.dense_ld = kBatchFirstSlab; t.dense_rows = rows;
            t.dense_row0 = (uint32_t)e->row_base + s0;
            t.seg_count = c->d_seg_count; t.seg_base = kBatchSegBase; t.nq_pad = nq_pad;
            if (!batch_gemm_segments(g, e->metric, &t.nseg, &t.seg_slots)) { t.nseg = 0; t.seg_slots = 0; }
            HIP_TRY(launch_tighten(t, st), WAX_HIP_ERR_INTERNAL, "tighten kernel launch");
            s0 += rows;
        }
        RescoreArgs r{};
        r.store = e->d_store; r.queries = d_queries; r.q_norm = c->d_qnorm; r.cand = c->d_cand; r.exact = c->d_exact;
        r.n_rows = n; r.row_base = (uint32_t)e->row_base; r.dims = D; r.nq = qn; r.cand_cap = kBatchCandCap; r.kp = kp;
        HIP_TRY(launch_rescore(r, e->metric, st), WAX_HIP_ERR_INTERNAL, "rescore kernel launch");
        HIP_TRY(launch_finalize_batch(c->d_cand, kBatchCandCap, c->d_overflow, c->d_exact, kp, k_eff, c->d_eps, e->d_ids,
                                      (uint32_t)e->row_base, n, qn, d_out, out_stride, c->h_cert + cert_off, st),
                WAX_HIP_ERR_INTERNAL, "finalize kernel launch");
    }
    e->st_batch_queries += qn;
    e->st_searches += qn;
    e->st_rows += (uint64_t)qn * n;
    e->st_bytes += (uint64_t)n * D * 2ull;
    return WAX_HIP_OK;
}

// Can the MFMA pipelines answer (nq queries, k_eff) on this engine at all?
bool batch_mfma_applicable(wax_hip_engine* e, uint32_t dims, int k_eff, uint32_t nq, OnepassPlan* plan, bool* onepass) {
    if (e->batch_mode.load() == 0 || dims != e->dims || (dims % 64u) != 0 || e->count == 0 || k_eff <= 0) return false;
    *onepass = plan_onepass(e, (uint32_t)e->count, k_eff, nq, plan);
    return *onepass || k_eff <= kBatchMaxKSlab;
}

// The whole batched search on device-resident queries (shared lock held, mirror ready): enqueue every block of
// kBatchMaxQ queries back to back (no host round trip in between), one synchronisation, then the uncertified queries are
// re-run on the exact single-query path, their hits written over the same rows of d_out. d_out rows are out_stride wide.
// Enqueue a whole device-resident batch (blocks of <= kBatchMaxQ queries) on the workspace's stream, followed by the
// download of the certificate flags and exact norms. Nothing is synchronised: batch_finish_device_locked completes it.
int batch_submit_device_locked(wax_hip_engine* e, BatchCtx* c, const float* d_queries, uint32_t nq, int k_eff,
                               const OnepassPlan* plan, wax_hip_hit* d_out, uint32_t out_stride) {
    hipStream_t st = c->stream;
    const uint32_t D = e->dims;
    int rc = bctx_reserve(c, plan ? plan->seg_area : kBatchCandCap, plan ? (uint64_t)plan->kp : (uint64_t)FUSED_MAX_K,
                          plan ? plan->sample_tiles : 0, nq, plan == nullptr);
    if (rc != WAX_HIP_OK) return rc;
    for (uint32_t q0 = 0; q0 < nq; q0 += kBatchMaxQ) {
        const uint32_t qn = (nq - q0 < kBatchMaxQ) ? nq - q0 : kBatchMaxQ;
        rc = batch_enqueue(e, c, d_queries + (uint64_t)q0 * D, qn, k_eff, plan, d_out + (uint64_t)q0 * out_stride, out_stride, q0);
        if (rc != WAX_HIP_OK) { (void)hipStreamSynchronize(st); return rc; }
    }
    // certificate flags and exact norms are written by the kernels straight into pinned host memory (h_cert, h_qnorm)
    return WAX_HIP_OK;
}

// Exact answers for `nf` queries of a device-resident block (rows fq[i] of d_queries, norms h_norm_by_query[fq[i]]), hits
// into rows fq[i] of d_out, on the workspace's stream; synchronised on return. Groups of scan_multi_group() queries share
// one pass over the f32 store; what the multi-query kernel does not serve (k > 192, unspecialised dims, a single query)
// takes one fused / general scan per query as before.
int exact_scan_queries(wax_hip_engine* e, BatchCtx* c, const float* d_queries, const uint32_t* fq, const float* h_norm_by_query,
                       uint32_t nf, int k_eff, wax_hip_hit* d_out, uint32_t out_stride) {
    hipStream_t st = c->stream;
    const uint32_t D = e->dims;
    const uint32_t n = (uint32_t)e->count;
    int rc = WAX_HIP_OK;
    uint32_t done = 0;
    const uint32_t group = (e->batch_multi.load() != 0 && nf >= 2 && e->force_general.load() == 0) ? scan_multi_group(D, k_eff) : 0u;
    if (group >= 2) {
        const int grid = scan_multi_grid(n, D, (int)e->grid_blocks.load());
        rc = grow_dev(&c->d_mpart, &c->mpart_cap, (uint64_t)group * (uint64_t)grid * (uint64_t)k_eff, sizeof(int64_t),
                      "Failed to allocate exact-pass partials");
        if (rc != WAX_HIP_OK) return rc;
        for (uint32_t base = 0; base < nf && rc == WAX_HIP_OK; base += kBatchMaxQ) {
            const uint32_t m = nf - base < kBatchMaxQ ? nf - base : kBatchMaxQ;
            for (uint32_t i = 0; i < m; ++i) {
                c->h_qlist[i] = fq[base + i];
                c->h_fnorm[i] = h_norm_by_query[fq[base + i]];
            }
            HIP_TRY(hipMemcpyAsync(c->d_qlist, c->h_qlist, m * sizeof(uint32_t), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "exact-pass list upload");
            HIP_TRY(hipMemcpyAsync(c->d_fnorm, c->h_fnorm, m * sizeof(float), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "exact-pass norm upload");
            for (uint32_t g0 = 0; g0 < m; g0 += group) {
                ScanMultiArgs a{};
                a.store = e->d_store; a.queries = d_queries; a.qlist = c->d_qlist + g0; a.q_norm = c->d_fnorm + g0;
                a.partials = c->d_mpart; a.n_rows = n; a.row_base = (uint32_t)e->row_base; a.dims = D;
                a.nq = m - g0 < group ? m - g0 : group; a.k = k_eff;
                int used_grid = 0;
                HIP_TRY(launch_scan_multi(a, e->metric, (int)e->grid_blocks.load(), st, &used_grid), WAX_HIP_ERR_INTERNAL, "multi-query scan launch");
                HIP_TRY(launch_merge_keys_multi(c->d_mpart, (uint32_t)used_grid, k_eff, e->d_ids, a.row_base, n, d_out, out_stride,
                                                c->d_qlist + g0, a.nq, st), WAX_HIP_ERR_INTERNAL, "multi-query merge launch");
                e->st_multi_passes += 1;
                e->st_rows += e->count;
                e->st_bytes += e->count * (uint64_t)D * 4ull;
            }
            HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "exact pass failed on device");   // the pinned lists are reused
            done += m;
        }
        e->st_multi_queries += done;
        e->st_searches += done;
        return rc;
    }
    Slot* s = nullptr;
    for (; done < nf; ++done) {
        const uint32_t q = fq[done];
        if (!s) {
            rc = acquire_slot(e, &s, /*try_only=*/false, /*holding=*/true);
            if (rc != WAX_HIP_OK) return rc;
        }
        rc = enqueue_scan(e, d_queries + (uint64_t)q * D, h_norm_by_query[q], k_eff, (int)out_stride, s->d_partials, s,
                          d_out + (uint64_t)q * out_stride, st, nullptr, nullptr, /*chain=*/false);
        if (rc != WAX_HIP_OK) break;
    }
    if (s) {
        (void)hipStreamSynchronize(st);
        release_slot(e, s);
    }
    return rc;
}

// Wait for a submitted batch; queries whose certificate failed (ties, clustered data, an overflowed list) are answered
// by the exact path in place — never an approximation.
// out_rewritten: queries whose rows of d_out were written HERE (full retries + exact path), i.e. after whatever the caller
// enqueued behind the submit — the sharded handle re-sends a shard's part when it is not 0.
int batch_finish_device_locked(wax_hip_engine* e, BatchCtx* c, const float* d_queries, uint32_t nq, int k_eff,
                               wax_hip_hit* d_out, uint32_t out_stride, uint32_t* out_fallbacks, uint32_t* out_rewritten = nullptr) {
    uint32_t retried = 0;
    hipStream_t st = c->stream;
    const uint32_t D = e->dims;
    HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "batch search failed on device");
    if (c->gemm_timed) {
        c->gemm_timed = false;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_g0, c->ev_g1) == hipSuccess) {
            std::unique_lock<std::mutex> sg(e->st_mu);
            e->st_gemm_ms += ms; e->st_gemm_timed += 1; e->st_gemm_rows += c->gemm_rows; e->st_gemm_queries += c->gemm_queries;
        }
    }
    int rc = WAX_HIP_OK;
    {   // queries the device-side full retry certified (flag value 2); and the hint that decides whether the NEXT batches carry
        // the retry kernel: armed (for 16 batches) by any query the first finish left uncertified, counted down by clean batches
        uint32_t inl = 0, unc = 0;
        for (uint32_t q = 0; q < nq; ++q) { inl += c->h_cert[q] == 2u; unc += c->h_cert[q] != 1u; }
        e->st_batch_retries += inl;
        e->st_batch_inline_retries += inl;
        if (unc > 0) e->retry_hint.store(16);
        else {   // count a clean batch down, never below zero (concurrent collects race on this word)
            int cur = e->retry_hint.load();
            while (cur > 0 && !e->retry_hint.compare_exchange_weak(cur, cur - 1)) {}
        }
    }
    // Second rung of the ladder, without another pass over the store: an uncertified query's survivors — EVERY row the
    // filtering GEMM admitted (approx distance <= tau) — are still in its segments, so all of them are re-scored exactly
    // (rescore_kernel in survivor-area mode) and the best k selected (full_retry_select_kernel). With every survivor re-scored
    // the certificate only needs tau - eps > the exact k-th: it no longer matters how many rows sit inside the bf16 error band
    // of the k-th neighbour (dense neighbourhoods, runs of near-duplicates: what k' = 2k + 32 candidates cannot cover), only
    // that no segment overflowed — which is why the planner gives every segment room for 128 survivors. A query that fails
    // again (overflow, fewer than k survivors, a tie ACROSS the threshold) goes to the exact path below.
    if (c->last_finish_valid && nq <= kBatchMaxQ && e->batch_retry.load() != 0 && batch_finish_fused_dims(D)) {
        std::vector<uint32_t> failing;
        for (uint32_t q = 0; q < nq; ++q)
            if (!c->h_cert[q]) failing.push_back(q);
        const FinishArgs& lf = c->last_finish;
        const uint64_t area = lf.cand_cap;
        const uint64_t per_round = std::max<uint64_t>(1, std::min<uint64_t>(kBatchMaxQ, (64ull << 20) / (area * sizeof(int64_t))));
        for (size_t off = 0; off < failing.size() && rc == WAX_HIP_OK; off += per_round) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(per_round, failing.size() - off);
            rc = grow_dev(&c->d_rescue, &c->rescue_cap, (uint64_t)m * area, sizeof(int64_t), "Failed to allocate full-retry keys");
            if (rc != WAX_HIP_OK) { rc = WAX_HIP_OK; break; }   // no memory for the retry: the exact path still answers
            std::memcpy(c->h_qlist, failing.data() + off, m * sizeof(uint32_t));
            HIP_TRY(hipMemcpyAsync(c->d_qlist, c->h_qlist, m * sizeof(uint32_t), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "retry list upload");
            // step 1: the live survivors of every query of the round, packed; their counts come back through pinned memory
            CompactArgs ca{};
            ca.cand = lf.cand; ca.cand_cap = lf.cand_cap; ca.seg_count = lf.seg_count; ca.nseg = lf.nseg; ca.seg_slots = lf.seg_slots;
            ca.nq_pad = lf.nq_pad; ca.count_stride = lf.count_stride; ca.qlist = c->d_qlist; ca.n_slots = m;
            ca.dense = c->d_rescue; ca.dense_stride = (uint32_t)area; ca.live_out = c->h_live;
            HIP_TRY(launch_compact_survivors(ca, st), WAX_HIP_ERR_INTERNAL, "full-retry compaction launch");
            HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "full retry failed on device");
            uint32_t live_max = 0;
            for (uint32_t i = 0; i < m; ++i) live_max = std::max(live_max, c->h_live[i]);
            if (live_max == 0) continue;                      // nothing passed the filter: the exact path answers
            const uint32_t kp = (uint32_t)std::min<uint64_t>(area, ((uint64_t)live_max + 63u) & ~63ull);
            rc = grow_dev(&c->d_rescue_exact, &c->rescue_exact_cap, (uint64_t)m * kp, sizeof(int64_t), "Failed to allocate full-retry keys");
            if (rc != WAX_HIP_OK) { rc = WAX_HIP_OK; break; }
            // step 2: exact f32 distance of every survivor (scan_kernel's arithmetic)
            RescoreArgs r{};
            r.store = e->d_store; r.queries = d_queries; r.q_norm = c->d_qnorm; r.cand = c->d_rescue; r.exact = c->d_rescue_exact;
            r.n_rows = lf.n_rows; r.row_base = lf.row_base; r.dims = D; r.nq = m; r.cand_cap = (uint32_t)area; r.kp = (int)kp;
            r.qlist = c->d_qlist; r.by_slot = 1;
            HIP_TRY(launch_rescore(r, c->last_metric, st), WAX_HIP_ERR_INTERNAL, "full-retry rescore launch");
            // step 3: the k best + certificate
            FullRetryArgs fr{};
            fr.exact = c->d_rescue_exact; fr.area = kp; fr.qlist = c->d_qlist; fr.n_slots = m;
            fr.seg_count = lf.seg_count; fr.nseg = lf.nseg; fr.seg_slots = lf.seg_slots; fr.nq_pad = lf.nq_pad; fr.count_stride = lf.count_stride;
            fr.tau = lf.tau; fr.eps = lf.eps; fr.overflow = lf.overflow; fr.ids = lf.ids; fr.n_rows = lf.n_rows; fr.row_base = lf.row_base;
            fr.k = lf.k; fr.out = lf.out; fr.out_stride = lf.out_stride; fr.certified = lf.certified;
            HIP_TRY(launch_full_retry_select(fr, st), WAX_HIP_ERR_INTERNAL, "full-retry select launch");
            HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "full retry failed on device");   // the pinned list is reused; flags are read below
            e->st_batch_retries += m;
            retried += m;
        }
    }
    c->last_finish_valid = false;
    // Exact path for whatever is still uncertified. The queries share passes over the f32 store (up to 16 per pass,
    // scan_multi_kernel: scan_kernel's arithmetic, bit-identical distances) instead of taking one scan each.
    std::vector<uint32_t> failed;
    for (uint32_t q = 0; q < nq; ++q)
        if (!c->h_cert[q]) failed.push_back(q);
    const uint32_t fallbacks = (uint32_t)failed.size();
    if (fallbacks > 0) rc = exact_scan_queries(e, c, d_queries, failed.data(), c->h_qnorm, fallbacks, k_eff, d_out, out_stride);
    e->st_batch_fallbacks += fallbacks;
    if (out_fallbacks) *out_fallbacks = fallbacks;
    if (out_rewritten) *out_rewritten = retried + fallbacks;
    return rc;
}

int batch_search_device_locked(wax_hip_engine* e, BatchCtx* c, const float* d_queries, uint32_t nq, int k_eff,
                               const OnepassPlan* plan, wax_hip_hit* d_out, uint32_t out_stride, uint32_t* out_fallbacks) {
    int rc = batch_submit_device_locked(e, c, d_queries, nq, k_eff, plan, d_out, out_stride);
    if (rc != WAX_HIP_OK) return rc;
    return batch_finish_device_locked(e, c, d_queries, nq, k_eff, d_out, out_stride, out_fallbacks);
}

// "MV2V" encoding-2 segment: header + lengths, in MetalVectorEngine.deserialize's validation order and with its reasons
// (:716-808). One copy for the single-device engine and the sharded handle.
int validate_mv2v_segment(uint8_t metric, uint32_t engine_dims, const uint8_t* data, size_t len, uint64_t* out_n, uint64_t* out_vec_len,
                          uint64_t* out_id_len) {
    if (len < 36) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Metal segment too small: " + std::to_string(len) + " bytes");
    const uint8_t magic[4] = {0x4D, 0x56, 0x32, 0x56};
    if (std::memcmp(data, magic, 4) != 0) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Metal segment magic mismatch");
    uint16_t ver; std::memcpy(&ver, data + 4, 2);
    if (ver != 1) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Unsupported Metal segment version " + std::to_string(ver));
    if (data[6] != 2) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Unsupported Metal segment encoding " + std::to_string((int)data[6]));
    if (data[7] > 2 || data[7] != metric)
        return fail(WAX_HIP_ERR_BAD_SEGMENT, "Metric mismatch: expected " + std::to_string((int)metric) + ", got " + std::to_string((int)data[7]));
    uint32_t dims; std::memcpy(&dims, data + 8, 4);
    if (dims != engine_dims) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Dimension mismatch: expected " + std::to_string(engine_dims) + ", got " + std::to_string(dims));
    uint64_t n, vec_len; std::memcpy(&n, data + 12, 8); std::memcpy(&vec_len, data + 20, 8);
    for (int i = 0; i < 8; ++i)
        if (data[28 + i] != 0) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Metal segment reserved bytes must be zero");
    if (n > 0xffffffffull || vec_len != n * (uint64_t)dims * 4ull) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Vector data length mismatch");
    if ((uint64_t)len < 36 + vec_len + 8) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Metal segment missing frameId length");
    uint64_t id_len; std::memcpy(&id_len, data + 36 + vec_len, 8);
    if (id_len != n * 8ull) return fail(WAX_HIP_ERR_BAD_SEGMENT, "FrameId data length mismatch");
    if ((uint64_t)len < 36 + vec_len + 8 + id_len) return fail(WAX_HIP_ERR_BAD_SEGMENT, "Metal segment truncated frameId data");
    *out_n = n; *out_vec_len = vec_len; *out_id_len = id_len;
    return WAX_HIP_OK;
}

bool device_is_gfx950(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
    return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

// ===========================================================================
// C ABI
extern "C" {

int wax_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int wax_hip_available(void) {
    const int n = wax_hip_device_count();
    for (int d = 0; d < n; ++d)
        if (device_is_gfx950(d)) return 1;
    return 0;
}

uint32_t wax_hip_abi_version(void) { return WAX_HIP_ABI_VERSION; }

const char* wax_hip_last_error(void) { return g_last_error.c_str(); }

int wax_hip_engine_create(uint8_t metric, uint32_t dims, int device_id, wax_hip_engine** out) {
    if (!out) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (dims == 0) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0");  // :154-156
    if (dims > WAX_HIP_MAX_DIMENSIONS)                                                     // :157-162
        return fail(WAX_HIP_ERR_CAPACITY, "capacity exceeded: limit " + std::to_string(WAX_HIP_MAX_DIMENSIONS) +
                                              ", requested " + std::to_string(dims));
    if (metric > WAX_HIP_METRIC_L2) return fail(WAX_HIP_ERR_METRIC_UNSUPPORTED, "unsupported metric " + std::to_string((int)metric));
    const int n = wax_hip_device_count();
    if (n <= 0) return fail(WAX_HIP_ERR_NO_DEVICE, "HIP device not available");          // :167-169
    int dev = device_id;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= n) return fail(WAX_HIP_ERR_NO_DEVICE, "HIP device " + std::to_string(dev) + " not available");
    if (!device_is_gfx950(dev)) return fail(WAX_HIP_ERR_NO_DEVICE, "HIP device " + std::to_string(dev) + " is not gfx950 (MI355X)");

    DeviceGuard g(dev);
    wax_hip_engine* e = new wax_hip_engine();
    e->device = dev;
    e->metric = metric;
    e->dims = dims;
    if (const char* v = std::getenv("WAX_HIP_BATCH_REGA")) {  // default of the "batch_rega" tunable (A/B runs of the whole test suite)
        const long m = std::strtol(v, nullptr, 10);
        if (m == 0 || m == 1 || m == 5) e->batch_rega = m;
    }
    int rc = resize_store(e, WAX_HIP_INITIAL_RESERVE);  // :225-229
    if (rc == WAX_HIP_OK) {
        hipError_t err = hipMalloc(&e->d_sink, MAX_GRID_BLOCKS * sizeof(float));
        if (err != hipSuccess) rc = fail(WAX_HIP_ERR_ALLOC, std::string("Failed to allocate sink: ") + hipGetErrorString(err));
    }
    if (rc == WAX_HIP_OK) {
        // Timing stays ENABLED on purpose: the runtime gives a timing event its own completion signal, so a stream
        // waiting on it resumes when the scan ends. A hipEventDisableTiming event was observed to resolve only with
        // the next command of the recording stream (the merge kernel): +13 us of idle HBM per query.
        hipError_t err = hipEventCreateWithFlags(&e->scan_done, hipEventReleaseToDevice);
        if (err != hipSuccess) rc = fail(WAX_HIP_ERR_ALLOC, std::string("Failed to create event: ") + hipGetErrorString(err));
    }
    for (int i = 0; i < kMaxStreams && rc == WAX_HIP_OK; ++i) {
        hipError_t err = hipStreamCreateWithFlags(&e->streams[i], hipStreamNonBlocking);
        if (err != hipSuccess) rc = fail(WAX_HIP_ERR_ALLOC, std::string("Failed to create stream: ") + hipGetErrorString(err));
    }
    if (rc == WAX_HIP_OK) {
        Slot* s = nullptr;
        rc = alloc_slot(e, &s);  // one pooled slot up front, like transientBufferPool's seed entry (:266-273)
        if (rc == WAX_HIP_OK) {
            s->index = 0;
            s->stream = e->streams[0];
            e->all_slots.push_back(s);
            e->free_slots.push_back(s);
        }
    }
    if (rc != WAX_HIP_OK) {
        wax_hip_engine_destroy(e);
        return rc;
    }
    *out = e;
    return WAX_HIP_OK;
}

void wax_hip_engine_destroy(wax_hip_engine* e) {
    if (!e) return;
    if (e->sh) { sh_destroy(e); delete e; return; }
    DeviceGuard g(e->device);
    (void)hipDeviceSynchronize();
    for (Slot* s : e->all_slots) free_slot(s);
    for (int i = 0; i < kMaxStreams; ++i)
        if (e->streams[i]) (void)hipStreamDestroy(e->streams[i]);
    if (e->scan_done) (void)hipEventDestroy(e->scan_done);
    for (hipEvent_t ev : e->tev) if (ev) (void)hipEventDestroy(ev);
    if (e->h_pend) (void)hipHostFree(e->h_pend);
    for (int i = 0; i < kShardRing; ++i) {
        (void)hipFree(e->ring_d_query[i]);
        (void)hipHostFree(e->ring_h_query[i]);
        (void)hipFree(e->ring_d_partials[i]);
        if (e->ring_ev0[i]) (void)hipEventDestroy(e->ring_ev0[i]);
        if (e->ring_ev1[i]) (void)hipEventDestroy(e->ring_ev1[i]);
        if (e->ring_done[i]) (void)hipEventDestroy(e->ring_done[i]);
    }
    {
        BatchMirror& b = e->batch;
        (void)hipFree(b.d_cb); (void)hipFree(b.d_vn2); (void)hipFree(b.d_maxnorm); (void)hipFree(b.d_dirty);
        if (b.ev_ready) (void)hipEventDestroy(b.ev_ready);
        for (BatchCtx* c : e->bctx_all) free_bctx(c);
        for (FilterWork* f : e->filter_all) free_filter_work(f);
        (void)hipFree(e->idhash.d_table);
    }
    (void)hipFree(e->d_store);
    (void)hipFree(e->d_ids);
    (void)hipFree(e->d_sink);
    (void)hipFree(e->d_bounce);
    delete e;
}

uint32_t wax_hip_dimensions(const wax_hip_engine* e) { return e ? e->dims : 0; }
uint64_t wax_hip_count(const wax_hip_engine* e) { return e ? (e->sh ? sh_total_count(e) : e->count) : 0; }
uint8_t wax_hip_metric_of(const wax_hip_engine* e) { return e ? e->metric : 0; }
int wax_hip_device_of(const wax_hip_engine* e) { return e ? e->device : -1; }

// ---- mutation -------------------------------------------------------------

int wax_hip_reserve(wax_hip_engine* e, uint64_t rows) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (e->sh) return sh_reserve(e, rows);
    REFUSE_IF_HOLDING(e);
    DeviceGuard g(e->device);
    WriteGuard w(e->lock);
    sync_shard_work(e);
    int rc = reserve_rows(e, rows);
    if (rc == WAX_HIP_OK) e->idmap.reserve(rows);
    return rc;
}

int wax_hip_add_batch(wax_hip_engine* e, const uint64_t* frame_ids, const float* rows, uint64_t n, uint32_t dims) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (e->sh) return sh_add_batch(e, frame_ids, rows, n, dims);
    if (n == 0) return WAX_HIP_OK;  // :360
    if (!frame_ids || !rows) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "addBatch: null input");
    if (dims != e->dims) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, dims));  // :367-370
    REFUSE_IF_HOLDING(e);
    DeviceGuard g(e->device);
    WriteGuard w(e->lock);
    sync_shard_work(e);
    mirror_note_append(e);
    e->batch.mirror_wanted = 0;
    const size_t row_bytes = (size_t)e->dims * sizeof(float);
    if (n == 1) {
        // add(frameId:vector:) (:330-357): no device call at all on the common paths — the row goes to the staging
        // area and reaches HBM with the next reader (one copy for all rows added since)
        const int64_t existing = e->idmap.find(frame_ids[0]);
        const uint64_t pend = e->pend_rows.load(std::memory_order_relaxed);
        if (existing >= 0) {
            if ((uint64_t)existing >= e->count - pend) {     // still staged: overwrite in place
                std::memcpy(e->h_pend + ((uint64_t)existing - (e->count - pend)) * e->dims, rows, row_bytes);
            } else {
                HIP_TRY(hipMemcpy(e->d_store + (uint64_t)existing * e->dims, rows, row_bytes, hipMemcpyHostToDevice),
                        WAX_HIP_ERR_INTERNAL, "vector upload");
            }
            mirror_note_upsert(e, (uint64_t)existing);
            return WAX_HIP_OK;
        }
        int rc1 = reserve_rows(e, e->count + 1);             // :341 (flushes before it reallocates)
        if (rc1 != WAX_HIP_OK) return rc1;
        if (!e->h_pend) {
            uint64_t cap = (8ull << 20) / row_bytes;
            if (cap < 1) cap = 1;
            if (cap > 65536) cap = 65536;
            HIP_TRY(hipHostMalloc(&e->h_pend, (size_t)cap * row_bytes, hipHostMallocDefault), WAX_HIP_ERR_ALLOC,
                    "Failed to allocate append staging buffer");
            e->pend_cap = cap;
        }
        if (e->pend_rows.load(std::memory_order_relaxed) == e->pend_cap) {
            rc1 = flush_pending(e);
            if (rc1 != WAX_HIP_OK) return rc1;
        }
        const uint64_t slot = e->pend_rows.load(std::memory_order_relaxed);
        std::memcpy(e->h_pend + slot * e->dims, rows, row_bytes);
        e->idmap.put(frame_ids[0], (uint32_t)e->count);
        e->ids.push_back(frame_ids[0]);
        e->count += 1;
        e->pend_rows.store(slot + 1, std::memory_order_release);
        return WAX_HIP_OK;
    }
    int rc = flush_pending(e);               // batch rows go straight to HBM; keep the staged ones in front of them
    if (rc != WAX_HIP_OK) return rc;
    rc = reserve_rows(e, e->count + n);      // :379-380 (upper bound: every id new)
    if (rc != WAX_HIP_OK) return rc;
    // Sequential upsert semantics of :384-398; consecutive appends are flushed as one H2D copy.
    uint64_t run_start = 0, run_len = 0;  // pending append run: input rows [run_start, run_start+run_len)
    auto flush = [&]() -> int {
        if (run_len == 0) return WAX_HIP_OK;
        const uint64_t first_row = e->count - run_len;
        HIP_TRY(hipMemcpy(e->d_store + first_row * e->dims, rows + run_start * e->dims, (size_t)run_len * row_bytes,
                          hipMemcpyHostToDevice), WAX_HIP_ERR_INTERNAL, "vector upload");
        HIP_TRY(hipMemcpy(e->d_ids + first_row, e->ids.data() + first_row, (size_t)run_len * sizeof(uint64_t),
                          hipMemcpyHostToDevice), WAX_HIP_ERR_INTERNAL, "frame id upload");
        run_len = 0;
        return WAX_HIP_OK;
    };
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t existing = e->idmap.find(frame_ids[i]);
        if (existing >= 0) {
            if ((rc = flush()) != WAX_HIP_OK) return rc;
            HIP_TRY(hipMemcpy(e->d_store + (uint64_t)existing * e->dims, rows + i * e->dims, row_bytes,
                              hipMemcpyHostToDevice), WAX_HIP_ERR_INTERNAL, "vector upload");
            mirror_note_upsert(e, (uint64_t)existing);
        } else {
            if (run_len == 0) run_start = i;
            e->idmap.put(frame_ids[i], (uint32_t)e->count);
            e->ids.push_back(frame_ids[i]);
            e->count += 1;
            run_len += 1;
        }
    }
    return flush();
}

int wax_hip_add(wax_hip_engine* e, uint64_t frame_id, const float* vector, uint32_t dims) {
    return wax_hip_add_batch(e, &frame_id, vector, 1, dims);  // :330-357 is the n == 1 case of :359-402
}

int wax_hip_add_batch_device(wax_hip_engine* e, const uint64_t* frame_ids, const float* d_rows, uint64_t n, uint32_t dims) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (e->sh) return sh_add_batch_device(e, frame_ids, d_rows, n, dims);
    if (n == 0) return WAX_HIP_OK;
    if (!frame_ids || !d_rows) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "addBatch: null input");
    if (dims != e->dims) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, dims));
    REFUSE_IF_HOLDING(e);
    DeviceGuard g(e->device);
    WriteGuard w(e->lock);
    sync_shard_work(e);
    mirror_note_append(e);
    e->batch.mirror_wanted = 0;
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) return frc; }
    for (uint64_t i = 0; i < n; ++i)
        if (e->idmap.find(frame_ids[i]) >= 0)
            return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "add_batch_device: frame id " + std::to_string(frame_ids[i]) + " already present (append-only path)");
    int rc = reserve_rows(e, e->count + n);
    if (rc != WAX_HIP_OK) return rc;
    const uint64_t first_row = e->count;
    // Failure-atomic: duplicates inside the batch are detected on a scratch map, the rows and ids reach HBM next, and the
    // host bookkeeping (ids, id map, count) is committed only after both copies succeeded — a failed copy leaves the
    // engine exactly as it was (the bytes written past `count` are invisible to every reader).
    {
        IdMap seen;
        seen.reserve((size_t)n);
        for (uint64_t i = 0; i < n; ++i) {
            if (seen.find(frame_ids[i]) >= 0)
                return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "add_batch_device: duplicate frame id " + std::to_string(frame_ids[i]) + " inside batch");
            seen.put(frame_ids[i], 0);
        }
    }
    HIP_TRY(hipMemcpy(e->d_store + first_row * e->dims, d_rows, (size_t)n * e->dims * sizeof(float), hipMemcpyDeviceToDevice),
            WAX_HIP_ERR_INTERNAL, "vector copy");
    HIP_TRY(hipMemcpy(e->d_ids + first_row, frame_ids, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice),
            WAX_HIP_ERR_INTERNAL, "frame id upload");
    e->ids.reserve(e->ids.size() + n);
    e->idmap.reserve((size_t)(first_row + n));
    for (uint64_t i = 0; i < n; ++i) {
        e->idmap.put(frame_ids[i], (uint32_t)(first_row + i));
        e->ids.push_back(frame_ids[i]);
    }
    e->count += n;
    return WAX_HIP_OK;
}

int wax_hip_apply_put_embeddings(wax_hip_engine* e, const uint8_t* payloads, uint64_t len, uint64_t* out_applied) {
    if (out_applied) *out_applied = 0;
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (len == 0) return WAX_HIP_OK;
    if (!payloads) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "applyPutEmbeddings: null input");
    constexpr uint64_t kHead = 1 + 8 + 4;  // opcode, frameId, dimension (WALEntryCodec.swift:39-42)
    std::vector<uint64_t> ids;
    std::vector<uint64_t> offs;  // byte offset of each record's f32 payload
    uint64_t pos = 0;
    while (pos < len) {
        if (payloads[pos] != 0x04) {
            char b[64];
            snprintf(b, sizeof b, "unknown opcode 0x%02X at byte %llu", payloads[pos], (unsigned long long)pos);
            return fail(WAX_HIP_ERR_BAD_SEGMENT, payloads[pos] >= 0x01 && payloads[pos] <= 0x03
                                                     ? std::string("not a putEmbedding entry: ") + b : std::string(b));
        }
        if (len - pos < kHead) return fail(WAX_HIP_ERR_BAD_SEGMENT, "truncated putEmbedding header");
        uint64_t id;
        uint32_t dim;
        memcpy(&id, payloads + pos + 1, 8);
        memcpy(&dim, payloads + pos + 9, 4);
        if (dim > WAX_HIP_MAX_DIMENSIONS) return fail(WAX_HIP_ERR_BAD_SEGMENT, "embedding dimension exceeds limit");  // :117-119
        if (dim != e->dims) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, dim));
        const uint64_t bytes = (uint64_t)dim * 4;
        if (len - pos - kHead < bytes) return fail(WAX_HIP_ERR_BAD_SEGMENT, "truncated putEmbedding vector");
        ids.push_back(id);
        offs.push_back(pos + kHead);
        pos += kHead + bytes;
    }
    std::vector<float> rows((size_t)ids.size() * e->dims);
    for (size_t i = 0; i < ids.size(); ++i) memcpy(rows.data() + i * e->dims, payloads + offs[i], (size_t)e->dims * 4);
    const int rc = wax_hip_add_batch(e, ids.data(), rows.data(), ids.size(), e->dims);
    if (rc == WAX_HIP_OK && out_applied) *out_applied = ids.size();
    return rc;
}

int wax_hip_remove(wax_hip_engine* e, uint64_t frame_id) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (e->sh) return sh_remove(e, frame_id);
    REFUSE_IF_HOLDING(e);
    DeviceGuard g(e->device);
    WriteGuard w(e->lock);
    sync_shard_work(e);
    if (e->count == 0) return WAX_HIP_OK;                 // :425
    const int64_t idx = e->idmap.find(frame_id);
    if (idx < 0) return WAX_HIP_OK;                       // :426
    e->batch.mirror_wanted = 0;
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) return frc; }   // the shift below works on device rows
    const uint64_t after = e->count - 1 - (uint64_t)idx;  // :431
    if (after > 0) {
        if (!e->d_bounce)
            HIP_TRY(hipMalloc(&e->d_bounce, kBounceBytes), WAX_HIP_ERR_ALLOC, "Failed to allocate bounce buffer");
        const uint64_t rb = (uint64_t)e->dims * sizeof(float);
        HIP_TRY(device_shift_down(e->d_store, (uint64_t)idx * rb, (uint64_t)(idx + 1) * rb, after * rb, e->d_bounce,
                                  kBounceBytes, nullptr), WAX_HIP_ERR_INTERNAL, "row shift");
        HIP_TRY(device_shift_down(e->d_ids, (uint64_t)idx * 8, (uint64_t)(idx + 1) * 8, after * 8, e->d_bounce,
                                  kBounceBytes, nullptr), WAX_HIP_ERR_INTERNAL, "frame id shift");
        HIP_TRY(hipStreamSynchronize(nullptr), WAX_HIP_ERR_INTERNAL, "row shift sync");
    }
    { const int mrc = mirror_note_remove(e, (uint64_t)idx); if (mrc != WAX_HIP_OK) return mrc; }   // the mirror's tail follows the store's
    e->ids.erase(e->ids.begin() + idx);                   // :440
    e->idmap.erase_row(frame_id, (uint32_t)idx);
    e->count -= 1;                                        // :441
    return WAX_HIP_OK;
}

// ---- search ---------------------------------------------------------------

// try_only: never block waiting for a scratch slot (returns kSlotBusy) — used by callers that already hold
// tickets, which must collect one instead of waiting (two such callers would starve each other).
static int submit_impl(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint64_t* out_ticket,
                       bool try_only);

static bool chain_scans(wax_hip_engine* e, int tk_mode) {
    const int64_t sc = e->scan_chain.load();
    return sc > 0 || (sc < 0 && tk_mode != 0);
}

int wax_hip_search_submit(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint64_t* out_ticket) {
    if (e && e->sh) return sh_submit(e, query, dims, top_k, out_ticket);
    return submit_impl(e, query, dims, top_k, out_ticket, false);
}

static int submit_impl(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint64_t* out_ticket,
                       bool try_only) {
    if (!e || !out_ticket) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine/ticket is null");
    DeviceGuard g(e->device);
    e->lock.lock_shared(holding(e) > 0);             // withReadLock (:447)
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) { e->lock.unlock_shared(); return frc; } }   // staged single-frame appends reach HBM here
    bool others_in_flight = false;                     // uncollected single-query tickets of this engine, whoever holds them
    { std::unique_lock<std::mutex> tg(e->slot_mu); others_in_flight = !e->tickets.empty(); }
    Slot* s = nullptr;
    int rc = WAX_HIP_OK;
    do {
        if (e->count == 0) {                               // :448 — an empty ticket, no GPU work
            rc = acquire_slot(e, &s, try_only, holding(e) > 0);
            if (rc != WAX_HIP_OK) break;
            s->k_eff = 0; s->timed = false;
            break;
        }
        if (dims != e->dims || !query) {                   // validate (:449, 830-833)
            rc = fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, query ? dims : 0));
            break;
        }
        if (e->row_base + e->count > 0x100000000ull) {
            rc = fail(WAX_HIP_ERR_CAPACITY, "row_base + count exceeds UInt32 row indices");
            break;
        }
        const int limit = clamp_topk(top_k);               // :450
        const int k_eff = (uint64_t)limit < e->count ? limit : (int)e->count;  // :451
        rc = acquire_slot(e, &s, try_only, holding(e) > 0);
        if (rc != WAX_HIP_OK) break;
        s->k_eff = k_eff;
        const int tk_mode = (int)e->time_kernels.load();   // read once per submit
        s->timed = tk_mode != 0;
        const float qn = query_norm(query, dims);
        hipError_t err = hipSuccess;
        const bool overlap = scan_overlaps_merge(e, others_in_flight);
        const bool qargs = scan_uses_query_args(e, k_eff, true, overlap);
        if (!qargs) {
            std::memcpy(s->h_query, query, (size_t)dims * sizeof(float));  // :467-468
            err = hipMemcpyAsync(s->d_query, s->h_query, (size_t)dims * sizeof(float), hipMemcpyHostToDevice, s->stream);
            if (err != hipSuccess) { rc = fail(WAX_HIP_ERR_INTERNAL, std::string("query upload: ") + hipGetErrorString(err)); break; }
        } else {
            e->st_query_args++;      // the query rides in the launch packet: no copy on the stream
        }
        // The last kernel of the chain writes the k hits straight into the slot's pinned host buffer
        // (device-visible, 16*k bytes over PCIe): no D2H copy launch; visibility at ev_done.
        // "done_flag" (default 1): a scan that merges in the kernel (small grids) publishes a completion word in pinned memory behind
        // its hits and collect polls that word — no event behind the kernel. Not while kernels are timed (the timing events must have
        // completed when collect reads them).
        const bool want_flag = e->done_flag.load() != 0 && !s->timed && s->coherent;
        s->flag_wait = false;
        if (want_flag) s->done_seq += 1;
        bool flagged = false;
        rc = enqueue_scan(e, qargs ? nullptr : s->d_query, qn, k_eff, k_eff, s->d_partials, s, s->h_hits, s->stream,
                          s->timed ? s->ev0 : nullptr, s->timed ? s->ev1 : nullptr, /*chain=*/chain_scans(e, tk_mode), &s->t_start, &s->t_end, query,
                          want_flag ? s->h_done : nullptr, s->done_seq, &flagged, overlap, tk_mode);
        if (rc != WAX_HIP_OK) break;
        if (flagged) {
            s->flag_wait = true;
            e->st_flag_waits++;
        } else {
            err = hipEventRecord(s->ev_done, s->stream);
            if (err != hipSuccess) { rc = fail(WAX_HIP_ERR_INTERNAL, std::string("event record: ") + hipGetErrorString(err)); break; }
        }
    } while (0);
    if (rc != WAX_HIP_OK) {
        if (s) { (void)hipStreamSynchronize(s->stream); release_slot(e, s); }
        e->lock.unlock_shared();
        return rc;
    }
    {
        std::unique_lock<std::mutex> tg(e->slot_mu);
        const uint64_t t = e->next_ticket++;
        e->tickets[t] = s;
        *out_ticket = t;
    }
    note_submit(e, s);
    return WAX_HIP_OK;  // shared lock stays held until collect
}

// Shared tail of collect: either converts to (ids, scores) or hands back the raw hits (padded to kcap).
// `capacity`: entries the (ids, scores) arrays hold; `hits_cap`: entries of out_hits (every one is written).
static int collect_impl(wax_hip_engine* e, uint64_t ticket, uint64_t* out_ids, float* out_scores, uint32_t capacity,
                        uint32_t* out_count, wax_hip_hit* out_hits, uint32_t hits_cap);

int wax_hip_search_collect(wax_hip_engine* e, uint64_t ticket, uint64_t* out_ids, float* out_scores, uint32_t out_capacity,
                           uint32_t* out_count) {
    if (e && e->sh) return sh_collect(e, ticket, out_ids, out_scores, out_capacity, out_count);
    return collect_impl(e, ticket, out_ids, out_scores, out_capacity, out_count, nullptr, 0);
}

static int collect_impl(wax_hip_engine* e, uint64_t ticket, uint64_t* out_ids, float* out_scores, uint32_t capacity,
                        uint32_t* out_count, wax_hip_hit* out_hits, uint32_t hits_cap) {
    if (!e || !out_count) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine/out_count is null");
    DeviceGuard g(e->device);
    Slot* s = nullptr;
    {
        std::unique_lock<std::mutex> tg(e->slot_mu);
        auto it = e->tickets.find(ticket);
        if (it == e->tickets.end()) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "unknown ticket " + std::to_string(ticket));
        s = it->second;
        e->tickets.erase(it);
    }
    int rc = WAX_HIP_OK;
    *out_count = 0;
    if (s->k_eff == 0 && out_hits)
        for (uint32_t i = 0; i < hits_cap; ++i) out_hits[i] = wax_hip_hit{KEY_PAD, ID_PAD};
    if (s->k_eff > 0) {
        hipError_t err = hipSuccess;
        if (s->flag_wait) {
            // poll the completion word the kernel's last workgroup publishes behind the hits (system-scope release). The stream is
            // queried now and then so that a kernel that died without publishing fails loudly instead of spinning for ever.
            // Spin for a bounded time (the scans this path serves take 10 - 300 us), then give the core away between polls: many
            // collecting threads behind a busy stream must not each burn a core.
            const uint64_t want = s->done_seq;
            const auto t_spin = std::chrono::steady_clock::now();
            bool polite = false;
            for (uint32_t spins = 1;; ++spins) {
                if (__atomic_load_n(s->h_done, __ATOMIC_ACQUIRE) == want) break;
                if (!polite) {
                    cpu_relax();
                    if ((spins & 0x3ffu) == 0u &&
                        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_spin).count() > 400)
                        polite = true;
                } else {
                    std::this_thread::yield();
                    if ((spins & 0xffu) == 0u) std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
                if ((spins & 0x3fffu) == 0u || (polite && (spins & 0x3fu) == 0u)) {
                    const hipError_t qs = hipStreamQuery(s->stream);
                    if (qs == hipErrorNotReady) continue;
                    if (__atomic_load_n(s->h_done, __ATOMIC_ACQUIRE) == want) break;
                    err = qs == hipSuccess ? hipErrorUnknown : qs;   // the stream drained (or failed) and the word never came
                    break;
                }
            }
        } else {
            err = hipEventSynchronize(s->ev_done);   // commandBuffer completion (:577-582); later queries on the stream keep running
        }
        if (err != hipSuccess) {
            rc = fail(WAX_HIP_ERR_INTERNAL, std::string("search failed on device: ") + hipGetErrorString(err));
            // a scan that died part-way leaves its fused-merge ticket half counted; no later scan on this slot could ever be
            // "last arriver" again and callers would read stale hits with no error. Re-arm it (best effort: after a device
            // fault the runtime usually refuses this too, and every later call on the slot then fails loudly as well).
            std::string keep = g_last_error;
            (void)hipStreamSynchronize(s->stream);
            (void)hipMemsetAsync(partials_ticket(s->d_partials), 0, 128, s->stream);
            (void)hipStreamSynchronize(s->stream);
            (void)hipGetLastError();
            g_last_error = keep;
        } else {
            if (s->timed) {
                float ms = 0.f;
                if (s->t_start && s->t_end && hipEventElapsedTime(&ms, s->t_start, s->t_end) == hipSuccess) {
                    std::unique_lock<std::mutex> sg(e->st_mu);
                    e->st_last_ms = ms; e->st_total_ms += ms; e->st_timed += 1;
                }
            }
            if (out_hits) {
                uint32_t m = 0;
                for (uint32_t i = 0; i < hits_cap; ++i) {
                    out_hits[i] = (i < (uint32_t)s->k_eff) ? s->h_hits[i] : wax_hip_hit{KEY_PAD, ID_PAD};
                    if (out_hits[i].key != KEY_PAD) ++m;
                }
                *out_count = m;
            } else if ((!out_ids || !out_scores) && capacity > 0) {
                rc = fail(WAX_HIP_ERR_INVALID_ARGUMENT, "output arrays are null");
            } else {
                rc = hits_to_results(e->metric, s->h_hits, (uint32_t)s->k_eff, out_ids, out_scores, capacity, out_count);
            }
        }
    }
    note_collect(e, s);
    release_slot(e, s);
    e->lock.unlock_shared();
    return rc;
}

uint32_t wax_hip_result_capacity(int32_t top_k) { return (uint32_t)clamp_topk(top_k); }

int wax_hip_search(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint64_t* out_ids,
                   float* out_scores, uint32_t out_capacity, uint32_t* out_count) {
    uint64_t t = 0;
    int rc = wax_hip_search_submit(e, query, dims, top_k, &t);
    if (rc != WAX_HIP_OK) return rc;
    return wax_hip_search_collect(e, t, out_ids, out_scores, out_capacity, out_count);
}

// nq queries -> nq rows of `stride` hits (ascending key, every row KEY_PAD padded to `stride`). Chooses the MFMA path
// when the batch is a genuine GEMM, otherwise pipelines single-query scans over the scratch-slot pool. The row
// count is read under the lock by whichever path runs; the output layout depends on `stride` alone, so a
// concurrent add can never make the library write past the caller's arrays.
static int search_batch_hits_impl(wax_hip_engine* e, const float* queries, uint32_t nq, uint32_t dims, int32_t top_k,
                                  wax_hip_hit* out_hits, uint32_t stride, uint32_t* out_counts) {
    for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
    for (size_t i = 0; i < (size_t)nq * stride; ++i) out_hits[i] = wax_hip_hit{KEY_PAD, ID_PAD};
    if (stride == 0) return WAX_HIP_OK;
    const uint64_t cnt = e->count;                       // heuristics only; every path re-reads it under the lock
    const uint64_t limit = (uint64_t)clamp_topk(top_k);
    const uint64_t kguess = limit < cnt ? limit : cnt;
    // Enough queries for the scan to be a dense GEMM: bf16 MFMA path with exact re-score; queries
    // whose exactness certificate fails are re-run on the exact single-query path inside it.
    bool use_mfma = e->batch_mode.load() != 0 && (int64_t)nq >= e->batch_min.load() && dims == e->dims &&
                    (dims % 64u) == 0 && cnt > 0 && kguess <= (uint64_t)kBatchMaxK && kguess > 0;
    if (use_mfma && nq < 16) {
        // A small batch costs nq scans of the f32 store on the loop path but ONE pass over the bf16 mirror (half the
        // bytes) plus the fixed pipeline cost on the MFMA path, whatever nq is (measured, profiles/r01/bi_*: 100 K rows
        // 0.15 ms, 1 M rows 0.30 ms, 10 M rows 1.78 ms for any nq <= 16; loop path 0.05 / 0.25 / 2.22 ms PER query).
        // A stale mirror adds its rebuild (read f32, write bf16).
        const double elems = (double)cnt * (double)dims;
        const double t_loop = (double)nq * (elems * 4.0 / 6.9e12 + 25e-6);
        const double t_pass = 135e-6 + elems * 2.0 / 4.7e12;
        const double t_rebuild = mirror_rows_to_convert(e) * (double)dims * 6.0 / 5.0e12;
        use_mfma = t_pass + t_rebuild < t_loop;
        // ... but a steady stream of small batches amortises it: the third one in a row that a valid mirror would have
        // made cheaper pays for the rebuild
        if (!use_mfma && t_pass < t_loop && e->batch.mirror_wanted.fetch_add(1) >= 2) use_mfma = true;
    }
    if (use_mfma) {
        int brc = WAX_HIP_OK;
        bool ran = false;
        {
            DeviceGuard g(e->device);
            e->lock.lock_shared(holding(e) > 0);
            struct Unlock { RWLock& l; ~Unlock() { l.unlock_shared(); } } unlock{e->lock};
            { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) return frc; }
            if (e->row_base + e->count > 0x100000000ull)
                return fail(WAX_HIP_ERR_CAPACITY, "row_base + count exceeds UInt32 row indices");
            uint64_t k64 = limit < e->count ? limit : e->count;   // the row count this batch is answered on
            if (k64 > stride) k64 = stride;                       // a smaller result array: the best `stride` of them
            OnepassPlan plan{};
            bool onepass = false;
            if (batch_mfma_applicable(e, dims, (int)k64, nq, &plan, &onepass)) {
                const int k_eff = (int)k64;
                BatchCtx* c = nullptr;
                brc = acquire_bctx(e, &c);
                if (brc != WAX_HIP_OK) return brc;
                brc = ensure_mirror(e, c->stream);
                if (brc == WAX_HIP_OK) brc = bctx_reserve_host(e, c, nq, (uint64_t)nq * k_eff);
                if (brc == WAX_HIP_OK) {
                    hipError_t err = hipMemcpyAsync(c->d_q, queries, (size_t)nq * dims * sizeof(float), hipMemcpyHostToDevice, c->stream);
                    if (err != hipSuccess) brc = fail(WAX_HIP_ERR_INTERNAL, std::string("query upload: ") + hipGetErrorString(err));
                }
                if (brc == WAX_HIP_OK)
                    brc = batch_search_device_locked(e, c, c->d_q, nq, k_eff, onepass ? &plan : nullptr, c->d_hits, (uint32_t)k_eff, nullptr);
                if (brc == WAX_HIP_OK) {
                    hipError_t err = hipMemcpyAsync(c->h_hits, c->d_hits, (size_t)nq * k_eff * sizeof(wax_hip_hit), hipMemcpyDeviceToHost, c->stream);
                    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
                    if (err != hipSuccess) brc = fail(WAX_HIP_ERR_INTERNAL, std::string("hits download: ") + hipGetErrorString(err));
                }
                if (brc == WAX_HIP_OK) {
                    for (uint32_t q = 0; q < nq; ++q) {
                        std::memcpy(out_hits + (size_t)q * stride, c->h_hits + (size_t)q * k_eff, (size_t)k_eff * sizeof(wax_hip_hit));
                        uint32_t m = 0;
                        for (int i = 0; i < k_eff; ++i) m += c->h_hits[(size_t)q * k_eff + i].key != KEY_PAD;
                        out_counts[q] = m;
                    }
                    ran = true;
                } else {
                    (void)hipStreamSynchronize(c->stream);
                }
                release_bctx(e, c);
            }
        }
        if (brc != WAX_HIP_OK) return brc;
        if (ran) return WAX_HIP_OK;
    }
    // Loop path: pipelined single-query scans over the scratch-slot pool.
    std::vector<uint32_t> todo(nq);
    for (uint32_t q = 0; q < nq; ++q) todo[q] = q;
    const uint32_t depth = (uint32_t)(e->max_slots > 1 ? e->max_slots : 1);
    std::vector<uint64_t> tk(todo.size(), 0);
    size_t submitted = 0, collected = 0;
    int rc = WAX_HIP_OK;
    while (collected < todo.size()) {
        while (submitted < todo.size() && submitted - collected < depth) {
            // holding tickets already => never wait for a slot (another batch may be doing the same): collect instead
            rc = submit_impl(e, queries + (uint64_t)todo[submitted] * dims, dims, top_k, &tk[submitted],
                             /*try_only=*/submitted > collected);
            if (rc == kSlotBusy) { rc = WAX_HIP_OK; break; }
            if (rc != WAX_HIP_OK) break;
            ++submitted;
        }
        if (rc != WAX_HIP_OK) break;
        const uint32_t q = todo[collected];
        rc = collect_impl(e, tk[collected], nullptr, nullptr, 0, &out_counts[q], out_hits + (uint64_t)q * stride, stride);
        ++collected;
        if (rc != WAX_HIP_OK) break;
    }
    if (rc != WAX_HIP_OK) {  // drain whatever is still in flight so the shared lock is released
        std::string keep = g_last_error;
        uint32_t dummy = 0;
        std::vector<wax_hip_hit> tmp(stride ? stride : 1);
        for (size_t i = collected; i < submitted; ++i)
            (void)collect_impl(e, tk[i], nullptr, nullptr, 0, &dummy, tmp.data(), stride);
        g_last_error = keep;
    }
    return rc;
}

int wax_hip_search_batch_hits(wax_hip_engine* e, const float* queries, uint32_t nq, uint32_t dims, int32_t top_k,
                              wax_hip_hit* out_hits, uint32_t out_stride, uint32_t* out_counts) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (nq == 0) return WAX_HIP_OK;
    if (!queries || !out_counts || (!out_hits && out_stride)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null input");
    if (e->sh) return sh_search_batch_hits(e, queries, nq, dims, top_k, out_hits, out_stride, out_counts);
    return search_batch_hits_impl(e, queries, nq, dims, top_k, out_hits, out_stride, out_counts);
}

int wax_hip_search_batch(wax_hip_engine* e, const float* queries, uint32_t nq, uint32_t dims, int32_t top_k,
                         uint64_t* out_ids, float* out_scores, uint32_t out_stride, uint32_t* out_counts) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (nq == 0) return WAX_HIP_OK;
    if (!queries || !out_counts) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null input");
    if ((!out_ids || !out_scores) && out_stride) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "output arrays are null");
    const uint32_t limit = (uint32_t)clamp_topk(top_k);
    const uint32_t w = limit < out_stride ? limit : out_stride;          // hits per query worth fetching
    std::vector<wax_hip_hit> hits((size_t)nq * (w ? w : 1));
    int rc = e->sh ? sh_search_batch_hits(e, queries, nq, dims, top_k, hits.data(), w, out_counts)
                   : search_batch_hits_impl(e, queries, nq, dims, top_k, hits.data(), w, out_counts);
    if (rc != WAX_HIP_OK) return rc;
    for (uint32_t q = 0; q < nq; ++q)
        hits_to_results(e->metric, hits.data() + (size_t)q * w, w, out_ids + (uint64_t)q * out_stride,
                        out_scores + (uint64_t)q * out_stride, out_stride, &out_counts[q]);
    return WAX_HIP_OK;
}

// Device-resident form: queries already in HBM (row-major nq x dims f32 on the engine's device), hits left in HBM
// ([nq][out_stride], rows padded). Blocking; `stream` is the stream whose earlier work produced d_queries (the
// library's own stream waits for it), and on return every result is complete. The only PCIe traffic is nq
// certificate flags + norms (8 bytes per query) — or the query block itself when the batch has to take the loop path.
// Shared body of the blocking and the pipelined device-resident batch search. ticket == nullptr: blocking.
// Would wax_hip_search_batch_submit_device on this (single-device) engine finish device work before it returns? True for
// everything but the asynchronous MFMA pipeline with a ready mirror: the loop path (dims not a multiple of 64, k beyond the
// MFMA limits, batch_mode = 0, small nq under the cost model) ends with a stream synchronise, and a mirror (re)build is
// synchronous too. The sharded handle asks before it decides whether ONE thread may drive every shard's submit
// (sharded.inc: sh_batch_submit). Conservative for nq < 16 (the cost model has a side effect; a worker hop costs microseconds).
bool batch_submit_blocks(wax_hip_engine* e, uint32_t dims, int32_t top_k, uint32_t nq, uint32_t out_stride) {
    if (nq == 0 || e->count == 0) return false;
    const uint64_t limit = (uint64_t)clamp_topk(top_k);
    uint64_t k64 = limit < e->count ? limit : e->count;
    if (k64 > out_stride) k64 = out_stride;
    if (k64 == 0) return false;
    OnepassPlan plan{};
    bool onepass = false;
    if ((int64_t)nq < e->batch_min.load() || nq < 16) return true;
    if (!batch_mfma_applicable(e, dims, (int)k64, nq, &plan, &onepass)) return true;
    if (e->pend_rows.load() != 0) return true;               // staged appends are flushed (and the mirror rebuilt) inside submit
    // an incremental conversion (appended / upserted rows) is a small asynchronous launch; (re)allocation and a whole-store conversion are not
    return e->batch.mirror_cap < e->capacity || mirror_rows_to_convert(e) > 65536.0;
}

static int batch_device_impl(wax_hip_engine* e, const float* d_queries, uint32_t nq, uint32_t dims, int32_t top_k,
                             wax_hip_hit* d_out_hits, uint32_t out_stride, void* stream, uint64_t* ticket, const char* what) {
    if (ticket) *ticket = 0;
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    (void)what;
    if (e->sh) return sh_batch_device(e, d_queries, nq, dims, top_k, d_out_hits, out_stride, stream, ticket);   // queries / hits on the handle's first device
    if (nq > 0 && (!d_queries || !d_out_hits || out_stride == 0)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null input");
    if (nq > 0 && dims != e->dims) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, dims));
    DeviceGuard g(e->device);
    const bool holder = holding(e) > 0;
    e->lock.lock_shared(holder);
    bool keep_lock = false;   // a pending ticket keeps the shared lock until its collect
    struct Unlock { RWLock& l; bool& keep; ~Unlock() { if (!keep) l.unlock_shared(); } } unlock{e->lock, keep_lock};
    auto done_ticket = [&]() {   // answered (or nothing to do) at submit time: a ticket whose collect is a no-op
        if (!ticket) return;
        std::unique_lock<std::mutex> tg(e->bticket_mu);
        *ticket = e->next_bticket++;
        e->btickets[*ticket] = wax_hip_engine::BatchTicket{};
    };
    if (nq == 0) { done_ticket(); return WAX_HIP_OK; }
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) return frc; }
    if (e->row_base + e->count > 0x100000000ull) return fail(WAX_HIP_ERR_CAPACITY, "row_base + count exceeds UInt32 row indices");
    const uint64_t limit = (uint64_t)clamp_topk(top_k);
    uint64_t k64 = limit < e->count ? limit : e->count;
    if (k64 > out_stride) k64 = out_stride;
    const int k_eff = (int)k64;
    BatchCtx* c = nullptr;
    int rc = acquire_bctx(e, &c, /*try_only=*/ticket != nullptr && holder);
    if (rc != WAX_HIP_OK) return rc;
    bool keep_ctx = false;
    struct Release { wax_hip_engine* e; BatchCtx* c; bool& keep; ~Release() { if (!keep) release_bctx(e, c); } } release{e, c, keep_ctx};
    hipStream_t st = c->stream;
    // inputs are produced on the caller's stream
    HIP_TRY(hipEventRecord(c->ev_in, static_cast<hipStream_t>(stream)), WAX_HIP_ERR_INTERNAL, "input event record");
    HIP_TRY(hipStreamWaitEvent(st, c->ev_in, 0), WAX_HIP_ERR_INTERNAL, "input event wait");
    if (k_eff == 0) {   // empty engine: all padding
        std::vector<wax_hip_hit> pad((size_t)nq * out_stride, wax_hip_hit{KEY_PAD, ID_PAD});
        HIP_TRY(hipMemcpyAsync(d_out_hits, pad.data(), pad.size() * sizeof(wax_hip_hit), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "pad upload");
        HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "pad upload");
        done_ticket();
        return WAX_HIP_OK;
    }
    OnepassPlan plan{};
    bool onepass = false;
    bool use_mfma = (int64_t)nq >= e->batch_min.load() && batch_mfma_applicable(e, dims, k_eff, nq, &plan, &onepass);
    if (use_mfma && nq < 16) {   // same cost model as the host-pointer form
        const double elems = (double)e->count * (double)dims;
        const double t_loop = (double)nq * (elems * 4.0 / 6.9e12 + 25e-6);
        const double t_pass = 135e-6 + elems * 2.0 / 4.7e12;
        const double t_rebuild = mirror_rows_to_convert(e) * (double)dims * 6.0 / 5.0e12;
        use_mfma = t_pass + t_rebuild < t_loop;
        if (!use_mfma && t_pass < t_loop && e->batch.mirror_wanted.fetch_add(1) >= 2) use_mfma = true;
    }
    if (use_mfma) {
        rc = ensure_mirror(e, st);
        if (rc != WAX_HIP_OK) return rc;
        if (!ticket) return batch_search_device_locked(e, c, d_queries, nq, k_eff, onepass ? &plan : nullptr, d_out_hits, out_stride, nullptr);
        rc = batch_submit_device_locked(e, c, d_queries, nq, k_eff, onepass ? &plan : nullptr, d_out_hits, out_stride);
        if (rc != WAX_HIP_OK) return rc;
        wax_hip_engine::BatchTicket t;
        t.c = c; t.d_queries = d_queries; t.d_out = d_out_hits; t.nq = nq; t.out_stride = out_stride; t.k_eff = k_eff;
        note_submit_id(e, &t.owner);
        {
            std::unique_lock<std::mutex> tg(e->bticket_mu);
            *ticket = e->next_bticket++;
            e->btickets[*ticket] = t;
        }
        keep_ctx = true;
        keep_lock = true;
        return WAX_HIP_OK;
    }
    // loop path (batches the MFMA pipelines do not take): exact scans on the workspace's stream — groups of queries
    // share one pass over the f32 store (exact_scan_queries); the norms come from the host
    std::vector<float> hq((size_t)nq * dims);
    HIP_TRY(hipMemcpyAsync(hq.data(), d_queries, hq.size() * sizeof(float), hipMemcpyDeviceToHost, st), WAX_HIP_ERR_INTERNAL, "query download");
    HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "query download");
    std::vector<uint32_t> all(nq);
    std::vector<float> norms(nq);
    for (uint32_t q = 0; q < nq; ++q) {
        all[q] = q;
        norms[q] = query_norm(hq.data() + (size_t)q * dims, dims);
    }
    rc = exact_scan_queries(e, c, d_queries, all.data(), norms.data(), nq, k_eff, d_out_hits, out_stride);
    if (rc == WAX_HIP_OK) done_ticket();
    return rc;
}

int wax_hip_search_batch_hits_device(wax_hip_engine* e, const float* d_queries, uint32_t nq, uint32_t dims, int32_t top_k,
                                     wax_hip_hit* d_out_hits, uint32_t out_stride, void* stream) {
    return batch_device_impl(e, d_queries, nq, dims, top_k, d_out_hits, out_stride, stream, nullptr, "wax_hip_search_batch_hits_device");
}

int wax_hip_search_batch_submit_device(wax_hip_engine* e, const float* d_queries, uint32_t nq, uint32_t dims, int32_t top_k,
                                       wax_hip_hit* d_out_hits, uint32_t out_stride, void* stream, uint64_t* out_ticket) {
    if (!out_ticket) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "ticket is null");
    return batch_device_impl(e, d_queries, nq, dims, top_k, d_out_hits, out_stride, stream, out_ticket, "wax_hip_search_batch_submit_device");
}

// out_rewritten (internal; see batch_finish_device_locked): rows of the batch's output written during this call
int batch_collect_device_impl(wax_hip_engine* e, uint64_t ticket, uint32_t* out_fallbacks, uint32_t* out_rewritten) {
    if (out_fallbacks) *out_fallbacks = 0;
    if (out_rewritten) *out_rewritten = 0;
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (e->sh) return sh_batch_collect_device(e, ticket, out_fallbacks);
    wax_hip_engine::BatchTicket t;
    {
        std::unique_lock<std::mutex> tg(e->bticket_mu);
        auto it = e->btickets.find(ticket);
        if (it == e->btickets.end()) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "unknown batch ticket " + std::to_string(ticket));
        t = it->second;
        e->btickets.erase(it);
    }
    if (!t.c) return WAX_HIP_OK;   // answered at submit time
    DeviceGuard g(e->device);
    const int rc = batch_finish_device_locked(e, t.c, t.d_queries, t.nq, t.k_eff, t.d_out, t.out_stride, out_fallbacks, out_rewritten);
    if (rc != WAX_HIP_OK) (void)hipStreamSynchronize(t.c->stream);
    release_bctx(e, t.c);
    note_collect_id(e, t.owner);
    e->lock.unlock_shared();
    return rc;
}

int wax_hip_search_batch_collect_device(wax_hip_engine* e, uint64_t ticket, uint32_t* out_fallbacks) {
    return batch_collect_device_impl(e, ticket, out_fallbacks, nullptr);
}

// ---- sharded search -------------------------------------------------------

int wax_hip_set_row_base(wax_hip_engine* e, uint64_t row_base) {
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (row_base > 0xffffffffull) return fail(WAX_HIP_ERR_CAPACITY, "row_base exceeds UInt32 row indices");
    SHARDED_UNSUPPORTED(e, "wax_hip_set_row_base");
    REFUSE_IF_HOLDING(e);
    WriteGuard w(e->lock);
    e->row_base = row_base;
    return WAX_HIP_OK;
}

// q_norm < 0: compute it here; the sharded handle computes ||q|| once per query and hands it to every shard.
static int search_shard_device_impl(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, wax_hip_hit* d_out_hits,
                                    void* stream, float q_norm);

int wax_hip_search_shard_device(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k,
                                wax_hip_hit* d_out_hits, void* stream) {
    return search_shard_device_impl(e, query, dims, top_k, d_out_hits, stream, -1.0f);
}

static int search_shard_device_impl(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, wax_hip_hit* d_out_hits,
                                    void* stream, float q_norm) {
    if (!e || !d_out_hits) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine/output is null");
    SHARDED_UNSUPPORTED(e, "wax_hip_search_shard_device");
    if (dims != e->dims || !query) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, query ? dims : 0));
    const int kpad = clamp_topk(top_k);
    if (kpad > FUSED_MAX_K) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "top_k too large for the device-resident shard path (max 192)");
    DeviceGuard g(e->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    e->lock.lock_shared(holding(e) > 0);
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) { e->lock.unlock_shared(); return frc; } }
    int rc = WAX_HIP_OK;
    do {
        if (e->row_base + e->count > 0x100000000ull) { rc = fail(WAX_HIP_ERR_CAPACITY, "row_base + count exceeds UInt32 row indices"); break; }
        const uint32_t r = e->ring_next.fetch_add(1) % kShardRing;
        // One user of a ring entry at a time; the entry's previous use (kShardRing calls ago) must have finished on
        // ITS stream before the pinned query is overwritten and the partials are reused.
        std::unique_lock<std::mutex> rg(e->ring_mu[r]);
        if (e->ring_busy[r]) {
            hipError_t werr = hipEventSynchronize(e->ring_done[r]);
            e->ring_busy[r] = false;
            if (werr != hipSuccess) { rc = fail(WAX_HIP_ERR_INTERNAL, std::string("shard scratch wait: ") + hipGetErrorString(werr)); break; }
        }
        if (!e->ring_d_query[r]) {
            {
                hipError_t err = hipMalloc(&e->ring_d_query[r], (size_t)e->dims * sizeof(float));
                if (err == hipSuccess) err = hipHostMalloc(&e->ring_h_query[r], (size_t)e->dims * sizeof(float), hipHostMallocDefault);
                if (err == hipSuccess) err = hipMalloc(&e->ring_d_partials[r], kPartialsBytes);
                if (err == hipSuccess) err = hipMemset(partials_ticket(e->ring_d_partials[r]), 0, 128);
                if (err == hipSuccess) err = hipStreamSynchronize(nullptr);   // ordered before any scan on the shard's own stream
                if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ring_ev0[r], hipEventReleaseToDevice);
                if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ring_ev1[r], hipEventReleaseToDevice);
                if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ring_done[r], hipEventDisableTiming);
                if (err != hipSuccess) { rc = fail(WAX_HIP_ERR_ALLOC, std::string("Failed to allocate shard scratch: ") + hipGetErrorString(err)); break; }
            }
        }
        if (e->count == 0) {  // all padding
            std::vector<wax_hip_hit> pad((size_t)kpad, wax_hip_hit{KEY_PAD, ID_PAD});
            hipError_t err = hipMemcpyAsync(d_out_hits, pad.data(), pad.size() * sizeof(wax_hip_hit), hipMemcpyHostToDevice, st);
            if (err == hipSuccess) err = hipStreamSynchronize(st);
            if (err != hipSuccess) rc = fail(WAX_HIP_ERR_INTERNAL, std::string("pad upload: ") + hipGetErrorString(err));
            break;
        }
        const int k_eff = (uint64_t)kpad < e->count ? kpad : (int)e->count;
        const float qn = q_norm >= 0.0f ? q_norm : query_norm(query, dims);
        // the previous call's scan still running (its ring entry in use or not yet drained) = a stream of scans: see "merge_overlap_mb"
        bool prev_in_flight = false;
        if (e->merge_overlap_mb.load() > 0 && e->count * (uint64_t)e->dims * sizeof(float) >= (uint64_t)e->merge_overlap_mb.load() << 20) {
            const uint32_t pr = (r + kShardRing - 1) % kShardRing;
            std::unique_lock<std::mutex> pg(e->ring_mu[pr], std::try_to_lock);
            prev_in_flight = !pg.owns_lock() || (e->ring_busy[pr] && hipEventQuery(e->ring_done[pr]) == hipErrorNotReady);
            (void)hipGetLastError();
        }
        const bool overlap = scan_overlaps_merge(e, prev_in_flight);
        const bool qargs = scan_uses_query_args(e, k_eff, false, overlap);
        if (!qargs) {
            std::memcpy(e->ring_h_query[r], query, (size_t)dims * sizeof(float));
            hipError_t err = hipMemcpyAsync(e->ring_d_query[r], e->ring_h_query[r], (size_t)dims * sizeof(float), hipMemcpyHostToDevice, st);
            if (err != hipSuccess) { rc = fail(WAX_HIP_ERR_INTERNAL, std::string("query upload: ") + hipGetErrorString(err)); break; }
        } else {
            e->st_query_args++;
        }
        harvest_ring_event(e, (int)r);  // the entry's previous use has finished (waited for above)
        const int tk_mode = (int)e->time_kernels.load();   // read once per submit
        const bool timed = tk_mode != 0;
        // chained (when kernels are timed, or "scan_chain" = 1): scans issued on different caller streams never overlap
        // each other, while the merge kernel and whatever the caller enqueues next (RCCL all-gather, merge, download)
        // do overlap the following scan.
        rc = enqueue_scan(e, qargs ? nullptr : e->ring_d_query[r], qn, k_eff, kpad, e->ring_d_partials[r], nullptr, d_out_hits, st,
                          timed ? e->ring_ev0[r] : nullptr, timed ? e->ring_ev1[r] : nullptr, /*chain=*/chain_scans(e, tk_mode), &e->ring_t0[r], &e->ring_t1[r], query,
                          nullptr, 0, nullptr, overlap, tk_mode);
        if (rc == WAX_HIP_OK && timed) {
            std::unique_lock<std::mutex> sg(e->st_mu);
            e->ring_ev_pending[r] = true;
        }
        // whatever was enqueued (even a partial chain after an error) must drain before the entry is reused
        if (hipEventRecord(e->ring_done[r], st) == hipSuccess) e->ring_busy[r] = true;
        else (void)hipStreamSynchronize(st);
    } while (0);
    e->lock.unlock_shared();
    return rc;
}

int wax_hip_merge_hits_device(const wax_hip_hit* d_in, uint32_t n, uint32_t k, wax_hip_hit* d_out, void* stream) {
    if (!d_in || !d_out) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null device buffer");
    if (k < 1 || k > (uint32_t)FUSED_MAX_K || n > 16384u) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "merge_hits: k must be 1..192 and n <= 16384");
    HIP_TRY(launch_merge_hits(d_in, n, (int)k, d_out, static_cast<hipStream_t>(stream)), WAX_HIP_ERR_INTERNAL, "merge kernel launch");
    return WAX_HIP_OK;
}

int wax_hip_merge_batch_hits_device(const wax_hip_hit* d_in, uint32_t n_shards, uint32_t nq, uint32_t k_in, uint32_t k,
                                    wax_hip_hit* d_out, void* stream) {
    if (!d_in || !d_out) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null device buffer");
    if (nq == 0) return WAX_HIP_OK;
    if (k < 1 || k > (uint32_t)FUSED_MAX_K || n_shards == 0 || k_in == 0 || (uint64_t)n_shards * k_in > 16384u)
        return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "merge_batch_hits: k must be 1..192 and n_shards * k_in <= 16384");
    HIP_TRY(launch_merge_batch_hits(d_in, n_shards, nq, k_in, (int)k, d_out, static_cast<hipStream_t>(stream)),
            WAX_HIP_ERR_INTERNAL, "merge kernel launch");
    return WAX_HIP_OK;
}

int wax_hip_hits_to_results(uint8_t metric, const wax_hip_hit* hits, uint32_t n, uint64_t* out_ids, float* out_scores,
                            uint32_t* out_count) {
    if (!hits || !out_ids || !out_scores || !out_count) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    return hits_to_results(metric, hits, n, out_ids, out_scores, n, out_count);
}

// ---- rank fusion --------------------------------------------------------------

int wax_hip_rrf_fuse_batch_device(const wax_hip_rrf_lane* lanes, uint32_t n_lanes, uint32_t nq, int32_t k,
                                  uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_best_rank,
                                  uint32_t* d_out_sources, uint32_t out_stride, uint32_t* d_out_counts, void* stream) {
    if (nq == 0) return WAX_HIP_OK;
    if (!d_out_ids || !d_out_scores || out_stride == 0) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null output");
    if (n_lanes > WAX_HIP_RRF_MAX_LANES || (n_lanes && !lanes)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "at most 8 lanes");
    if (k > (1 << 30)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "k exceeds 2^30");
    uint64_t total = 0;
    for (uint32_t l = 0; l < n_lanes; ++l) {
        if (lanes[l].stride && !lanes[l].d_ids) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "lane without ids");
        if (lanes[l].weight > 0.0f) total += lanes[l].stride;   // a lane with weight <= 0 is skipped (HybridSearch.swift:31): it takes no table space
    }
    if (total > WAX_HIP_RRF_MAX_ENTRIES)
        return fail(WAX_HIP_ERR_CAPACITY, "capacity exceeded: limit " + std::to_string(WAX_HIP_RRF_MAX_ENTRIES) + ", requested " + std::to_string(total));
    HIP_TRY(launch_rrf_fuse(lanes, n_lanes, nq, k, d_out_ids, d_out_scores, d_out_best_rank, d_out_sources, out_stride, d_out_counts,
                            static_cast<hipStream_t>(stream)), WAX_HIP_ERR_INTERNAL, "fusion kernel launch");
    return WAX_HIP_OK;
}

int wax_hip_rrf_fuse(const float* weights, const uint64_t* const* lists, const uint32_t* list_counts, uint32_t n_lists,
                     int32_t k, int device_id, uint64_t* out_ids, float* out_scores, uint32_t* out_best_rank,
                     uint32_t* out_sources, uint32_t out_capacity, uint32_t* out_count) {
    if (out_count) *out_count = 0;
    if (!out_count || ((!out_ids || !out_scores) && out_capacity)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n_lists > WAX_HIP_RRF_MAX_LANES || (n_lists && (!weights || !lists || !list_counts))) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "at most 8 lists");
    uint64_t total = 0, counted = 0;
    for (uint32_t l = 0; l < n_lists; ++l) {
        if (list_counts[l] && !lists[l]) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "list without ids");
        total += list_counts[l];
        if (weights[l] > 0.0f) counted += list_counts[l];       // skipped lanes (weight <= 0, HybridSearch.swift:31) take no table space
    }
    if (counted > WAX_HIP_RRF_MAX_ENTRIES)
        return fail(WAX_HIP_ERR_CAPACITY, "capacity exceeded: limit " + std::to_string(WAX_HIP_RRF_MAX_ENTRIES) + ", requested " + std::to_string(counted));
    if (counted == 0 || out_capacity == 0) return WAX_HIP_OK;
    if (wax_hip_device_count() <= 0) return fail(WAX_HIP_ERR_NO_DEVICE, "HIP device not available");
    int dev = device_id;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    DeviceGuard g(dev);
    const uint32_t stride_out = (uint32_t)(counted < out_capacity ? counted : out_capacity);
    // one allocation: [ids of every list][out ids][out scores][out best rank][out sources][out count]
    const size_t in_bytes = (size_t)total * 8, out_bytes = (size_t)stride_out * (8 + 4 + 4 + 4) + 8;
    unsigned char* d = nullptr;
    HIP_TRY(hipMalloc(&d, in_bytes + out_bytes), WAX_HIP_ERR_ALLOC, "Failed to allocate fusion buffers");
    struct Free { void* p; ~Free() { (void)hipFree(p); } } guard{d};
    wax_hip_rrf_lane lanes[WAX_HIP_RRF_MAX_LANES] = {};
    size_t off = 0;
    for (uint32_t l = 0; l < n_lists; ++l) {
        if (list_counts[l])
            HIP_TRY(hipMemcpy(d + off, lists[l], (size_t)list_counts[l] * 8, hipMemcpyHostToDevice), WAX_HIP_ERR_INTERNAL, "list upload");
        lanes[l].d_ids = reinterpret_cast<const uint64_t*>(d + off); lanes[l].d_counts = nullptr;
        lanes[l].stride = list_counts[l]; lanes[l].pitch = 1; lanes[l].weight = weights[l];
        off += (size_t)list_counts[l] * 8;
    }
    uint64_t* d_ids = reinterpret_cast<uint64_t*>(d + in_bytes);
    float* d_scores = reinterpret_cast<float*>(d_ids + stride_out);
    uint32_t* d_rank = reinterpret_cast<uint32_t*>(d_scores + stride_out);
    uint32_t* d_src = d_rank + stride_out;
    uint32_t* d_cnt = d_src + stride_out;
    int rc = wax_hip_rrf_fuse_batch_device(lanes, n_lists, 1, k, d_ids, d_scores, d_rank, d_src, stride_out, d_cnt, nullptr);
    if (rc != WAX_HIP_OK) return rc;
    uint32_t m = 0;
    HIP_TRY(hipMemcpy(&m, d_cnt, sizeof(m), hipMemcpyDeviceToHost), WAX_HIP_ERR_INTERNAL, "fusion failed on device");
    if (m > stride_out) m = stride_out;
    HIP_TRY(hipMemcpy(out_ids, d_ids, (size_t)m * 8, hipMemcpyDeviceToHost), WAX_HIP_ERR_INTERNAL, "result download");
    HIP_TRY(hipMemcpy(out_scores, d_scores, (size_t)m * 4, hipMemcpyDeviceToHost), WAX_HIP_ERR_INTERNAL, "result download");
    if (out_best_rank) HIP_TRY(hipMemcpy(out_best_rank, d_rank, (size_t)m * 4, hipMemcpyDeviceToHost), WAX_HIP_ERR_INTERNAL, "result download");
    if (out_sources) HIP_TRY(hipMemcpy(out_sources, d_src, (size_t)m * 4, hipMemcpyDeviceToHost), WAX_HIP_ERR_INTERNAL, "result download");
    *out_count = m;
    return WAX_HIP_OK;
}

// ---- filtered search --------------------------------------------------------

int wax_hip_search_filtered(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, int has_allow,
                            const uint64_t* allow_frame_ids, uint64_t n_allow, int has_min_score, float min_score,
                            uint64_t* out_ids, float* out_scores, uint32_t out_capacity, uint32_t* out_count) {
    if (out_count) *out_count = 0;
    if (!e) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is null");
    if (!query || !out_count || ((!out_ids || !out_scores) && out_capacity)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (has_allow && n_allow > 0 && !allow_frame_ids) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "allow-list is null");
    if (e->sh) return sh_search_filtered(e, query, dims, top_k, has_allow, allow_frame_ids, n_allow, has_min_score, min_score, out_ids,
                                         out_scores, out_capacity, out_count);
    if (dims != e->dims) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, dims));
    const int kpad = clamp_topk(top_k);
    uint32_t n = 0;
    if (!has_allow) {
        // no allow-list: the ordinary scan, then the score cut
        int rc = wax_hip_search(e, query, dims, top_k, out_ids, out_scores, out_capacity, &n);
        if (rc != WAX_HIP_OK) return rc;
    } else {
        DeviceGuard g(e->device);
        e->lock.lock_shared(holding(e) > 0);
        struct Unlock { RWLock& l; ~Unlock() { l.unlock_shared(); } } unlock{e->lock};
        { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) return frc; }
        FilterWork* fp = nullptr;
        { const int arc = acquire_filter_work(e, &fp); if (arc != WAX_HIP_OK) return arc; }
        struct Release { wax_hip_engine* e; FilterWork* f; ~Release() { (void)hipStreamSynchronize(f->stream); release_filter_work(e, f); } } release{e, fp};
        FilterWork& f = *fp;
        hipStream_t st = f.stream;
        // allowed frame ids -> local rows, ascending and unique (row order is the tie-break order of every path)
        uint64_t m = 0;
        const int64_t dev_min = e->filter_device_min.load();
        if (e->count == 0 || n_allow == 0) {
            m = 0;
        } else if (dev_min >= 0 && n_allow >= (uint64_t)dev_min) {
            // long lists: the probes are cache misses (~60 ns each on the host, 5.65 ms for 1M ids); on the device the
            // same probes are a few tens of microseconds against the id -> row table in HBM (filter.hip), and the
            // bitmap they mark hands the rows back ascending and unique
            { const int hrc = ensure_idhash(e, st); if (hrc != WAX_HIP_OK) return hrc; }
            const uint64_t n_words = (e->count + 31) / 32, n_blocks = filter_bitmap_blocks((uint32_t)e->count);
            int grc = grow_dev(&f.d_allow, &f.allow_cap, n_allow, sizeof(uint64_t), "Failed to allocate allow-list");
            if (grc == WAX_HIP_OK) grc = grow_dev(&f.d_bitmap, &f.bitmap_words, n_words, sizeof(uint32_t), "Failed to allocate row bitmap");
            if (grc == WAX_HIP_OK) grc = grow_dev(&f.d_block_sum, &f.block_cap, n_blocks, sizeof(uint32_t), "Failed to allocate bitmap offsets");
            const uint64_t m_max = n_allow < e->count ? n_allow : e->count;
            if (grc == WAX_HIP_OK && f.cap < m_max) {
                uint64_t cap = 1024;
                while (cap < m_max) cap *= 2;
                uint64_t c1 = f.cap, c2 = f.cap, c3 = f.cap;
                grc = grow_dev(&f.d_rows, &c1, cap, sizeof(uint32_t), "Failed to allocate allowed-row list");
                if (grc == WAX_HIP_OK) grc = grow_dev(&f.d_ids, &c2, cap, sizeof(uint64_t), "Failed to allocate allowed-id list");
                if (grc == WAX_HIP_OK) grc = grow_dev(&f.d_dist, &c3, cap, sizeof(float), "Failed to allocate allowed-row distances");
                f.cap = grc == WAX_HIP_OK ? cap : 0;
            }
            if (grc != WAX_HIP_OK) return grc;
            HIP_TRY(hipMemcpyAsync(f.d_allow, allow_frame_ids, (size_t)n_allow * sizeof(uint64_t), hipMemcpyHostToDevice, st),
                    WAX_HIP_ERR_INTERNAL, "allow-list upload");
            HIP_TRY(launch_allow_probe(f.d_allow, n_allow, e->d_ids, (uint32_t)e->count, e->idhash.d_table, e->idhash.slots, f.d_bitmap,
                                       f.d_block_sum, f.d_total, st), WAX_HIP_ERR_INTERNAL, "allow-list probe launch");
            HIP_TRY(launch_allow_emit(f.d_bitmap, (uint32_t)e->count, f.d_block_sum, e->d_ids, f.d_rows, f.d_ids, st),
                    WAX_HIP_ERR_INTERNAL, "allow-list compaction launch");
            HIP_TRY(hipMemcpyAsync(f.h_total, f.d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, st), WAX_HIP_ERR_INTERNAL, "row count download");
            HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "allow-list probe failed on device");
            m = *f.h_total;
            e->st_filter_device++;
        } else {
            std::vector<uint32_t> rows;
            rows.reserve((size_t)n_allow);
            for (uint64_t i = 0; i < n_allow; ++i) {
                const int64_t r = e->idmap.find(allow_frame_ids[i]);
                if (r >= 0) rows.push_back((uint32_t)r);
            }
            std::sort(rows.begin(), rows.end());
            rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
            m = rows.size();
            if (m) {
                std::vector<uint64_t> ids((size_t)m);
                for (uint64_t i = 0; i < m; ++i) ids[i] = e->ids[rows[i]];
                if (f.cap < m) {
                    uint64_t cap = 1024;
                    while (cap < m) cap *= 2;
                    uint64_t c1 = f.cap, c2 = f.cap, c3 = f.cap;
                    int grc = grow_dev(&f.d_rows, &c1, cap, sizeof(uint32_t), "Failed to allocate allowed-row list");
                    if (grc == WAX_HIP_OK) grc = grow_dev(&f.d_ids, &c2, cap, sizeof(uint64_t), "Failed to allocate allowed-id list");
                    if (grc == WAX_HIP_OK) grc = grow_dev(&f.d_dist, &c3, cap, sizeof(float), "Failed to allocate allowed-row distances");
                    f.cap = grc == WAX_HIP_OK ? cap : 0;
                    if (grc != WAX_HIP_OK) return grc;
                }
                // pageable sources: the runtime stages them before returning, so the vectors may die at the end of this block
                HIP_TRY(hipMemcpyAsync(f.d_rows, rows.data(), (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "row list upload");
                HIP_TRY(hipMemcpyAsync(f.d_ids, ids.data(), (size_t)m * sizeof(uint64_t), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "id list upload");
                HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "row list upload");
            }
        }
        if (m == 0) { *out_count = 0; return WAX_HIP_OK; }
        const int k_eff = (uint64_t)kpad < m ? kpad : (int)m;
        const float qn = query_norm(query, dims);
        HIP_TRY(hipMemcpyAsync(f.d_query, query, (size_t)dims * sizeof(float), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "query upload");
        HIP_TRY(hipMemcpyAsync(f.d_qnorm, &qn, sizeof(float), hipMemcpyHostToDevice, st), WAX_HIP_ERR_INTERNAL, "query norm upload");
        RescoreArgs r{};   // exact f32 distances with scan_kernel's lane mapping and summation order
        r.store = e->d_store; r.queries = f.d_query; r.q_norm = f.d_qnorm; r.rows = f.d_rows; r.dist_out = f.d_dist;
        r.n_rows = (uint32_t)e->count; r.row_base = 0; r.dims = dims; r.nq = 1; r.cand_cap = 0; r.kp = (int)m;
        HIP_TRY(launch_rescore(r, e->metric, st), WAX_HIP_ERR_INTERNAL, "distance kernel launch");
        // keys of the compact list carry the POSITION in it; positions ascend with rows, so ties order as everywhere else
        HIP_TRY(launch_select_general(f.d_dist, (uint32_t)m, 0u, k_eff, k_eff, f.d_ids, f.sw, f.d_hits, st),
                WAX_HIP_ERR_INTERNAL, "select kernel launch");
        HIP_TRY(hipMemcpyAsync(f.h_hits, f.d_hits, (size_t)k_eff * sizeof(wax_hip_hit), hipMemcpyDeviceToHost, st),
                WAX_HIP_ERR_INTERNAL, "hits download");
        HIP_TRY(hipStreamSynchronize(st), WAX_HIP_ERR_INTERNAL, "filtered search failed on device");
        int rc = hits_to_results(e->metric, f.h_hits, (uint32_t)k_eff, out_ids, out_scores, out_capacity, &n);
        if (rc != WAX_HIP_OK) return rc;
        e->st_searches++;
        e->st_rows += m;
        e->st_bytes += m * (uint64_t)e->dims * 4ull;
    }
    if (has_min_score) {  // `score < minScore` drops a candidate (UnifiedSearch.swift:1248); results are best-first
        uint32_t keep = 0;
        for (uint32_t i = 0; i < n; ++i)
            if (!(out_scores[i] < min_score)) { out_ids[keep] = out_ids[i]; out_scores[keep] = out_scores[i]; ++keep; }
        n = keep;
    }
    *out_count = n;
    return WAX_HIP_OK;
}

// ---- persistence ----------------------------------------------------------

void wax_hip_free(void* p) { std::free(p); }

int wax_hip_serialize(wax_hip_engine* e, uint8_t** out_bytes, size_t* out_len) {
    if (!e || !out_bytes || !out_len) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (e->sh) return sh_serialize(e, out_bytes, out_len);
    DeviceGuard g(e->device);
    e->lock.lock_shared(holding(e) > 0);  // withReadLock (:683)
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) { e->lock.unlock_shared(); return frc; } }
    const uint64_t n = e->count;
    const uint64_t vec_bytes = n * (uint64_t)e->dims * 4ull;  // :697
    const uint64_t id_bytes = n * 8ull;                       // :707
    const size_t total = 36 + (size_t)vec_bytes + 8 + (size_t)id_bytes;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(total));
    if (!buf) { e->lock.unlock_shared(); return fail(WAX_HIP_ERR_ALLOC, "Failed to allocate serialize buffer"); }
    uint8_t* p = buf;
    const uint8_t magic[4] = {0x4D, 0x56, 0x32, 0x56};        // "MV2V" :686
    std::memcpy(p, magic, 4); p += 4;
    const uint16_t ver = 1; std::memcpy(p, &ver, 2); p += 2;  // :687
    *p++ = 2;                                                 // encoding 2 (flat) :689
    *p++ = e->metric;                                         // VecSimilarity raw :690
    std::memcpy(p, &e->dims, 4); p += 4;                      // :691
    std::memcpy(p, &n, 8); p += 8;                            // :693
    std::memcpy(p, &vec_bytes, 8); p += 8;                    // :698
    std::memset(p, 0, 8); p += 8;                             // reserved :700
    int rc = WAX_HIP_OK;
    if (vec_bytes) {
        hipError_t err = hipMemcpy(p, e->d_store, (size_t)vec_bytes, hipMemcpyDeviceToHost);  // :703-705
        if (err != hipSuccess) rc = fail(WAX_HIP_ERR_INTERNAL, std::string("vector download: ") + hipGetErrorString(err));
    }
    p += vec_bytes;
    std::memcpy(p, &id_bytes, 8); p += 8;                     // :708
    if (id_bytes) std::memcpy(p, e->ids.data(), (size_t)id_bytes);  // :710
    e->lock.unlock_shared();
    if (rc != WAX_HIP_OK) { std::free(buf); return rc; }
    *out_bytes = buf;
    *out_len = total;
    return WAX_HIP_OK;
}

int wax_hip_deserialize(wax_hip_engine* e, const uint8_t* data, size_t len) {
    if (!e || (!data && len)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (e->sh) return sh_deserialize(e, data, len);
    uint64_t n = 0, vec_len = 0, id_len = 0;
    { const int vrc = validate_mv2v_segment(e->metric, e->dims, data, len, &n, &vec_len, &id_len); if (vrc != WAX_HIP_OK) return vrc; }
    REFUSE_IF_HOLDING(e);
    DeviceGuard g(e->device);
    WriteGuard w(e->lock);  // withWriteLock (:717)
    sync_shard_work(e);
    mirror_note_replaced(e);
    e->batch.mirror_wanted = 0;
    e->pend_rows.store(0, std::memory_order_release);   // the store is replaced wholesale: staged appends are dropped with it
    // :790-792 — capacity only grows
    int rc = resize_store(e, n > e->capacity ? n : e->capacity);
    if (rc != WAX_HIP_OK) return rc;
    e->count = n;
    e->ids.resize((size_t)n);
    if (n) {
        std::memcpy(e->ids.data(), data + 36 + vec_len + 8, (size_t)id_len);  // :809-811
        HIP_TRY(hipMemcpy(e->d_store, data + 36, (size_t)vec_len, hipMemcpyHostToDevice), WAX_HIP_ERR_INTERNAL, "vector upload");  // :794-799
        HIP_TRY(hipMemcpy(e->d_ids, e->ids.data(), (size_t)id_len, hipMemcpyHostToDevice), WAX_HIP_ERR_INTERNAL, "frame id upload");
    }
    e->idmap.clear();
    e->idmap.reserve((size_t)n);
    for (uint64_t i = 0; i < n; ++i)
        if (e->idmap.find(e->ids[i]) < 0) e->idmap.put(e->ids[i], (uint32_t)i);  // firstIndex(of:) semantics: first row wins
    return WAX_HIP_OK;
}

// ---- observability / tuning -----------------------------------------------

int wax_hip_stats(wax_hip_engine* e, wax_hip_stats_t* out) {
    if (!e || !out) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (e->sh) return sh_stats(e, out);
    out->searches = e->st_searches.load();
    out->rows_scanned = e->st_rows.load();
    out->bytes_scanned = e->st_bytes.load();
    out->transient_allocations = e->st_alloc.load();
    out->reuse_count = e->st_reuse.load();
    out->reserved_rows = e->capacity;
    {
        DeviceGuard g(e->device);
        for (int r = 0; r < kShardRing; ++r) harvest_ring_event(e, r);
    }
    std::unique_lock<std::mutex> sg(e->st_mu);
    out->last_scan_kernel_ms = e->st_last_ms;
    out->scan_kernel_ms_total = e->st_total_ms;
    out->scan_kernels_timed = e->st_timed;
    out->batch_gemm_ms_total = e->st_gemm_ms;
    out->batch_gemms_timed = e->st_gemm_timed;
    out->batch_gemm_rows = e->st_gemm_rows;
    out->batch_gemm_queries = e->st_gemm_queries;
    return WAX_HIP_OK;
}

int wax_hip_set_tuning(wax_hip_engine* e, const char* key, int64_t value) {
    if (!e || !key) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const std::string k(key);
    if (e->sh) return sh_set_tuning(e, k, value);
    if (k == "grid_blocks") e->grid_blocks = value;
    else if (k == "variant") e->variant = value;
    else if (k == "time_kernels") e->time_kernels = value;
    else if (k == "force_general") e->force_general = value;
    else if (k == "stream_nt") e->stream_nt = value;
    else if (k == "fuse_merge") e->fuse_merge = value != 0;
    else if (k == "done_flag") e->done_flag = value != 0;
    else if (k == "merge_kway") e->merge_kway = value != 0;
    else if (k == "merge_overlap_mb") e->merge_overlap_mb = value < 0 ? 0 : value;
    else if (k == "batch_qfrag") e->batch_qfrag = value != 0;
    else if (k == "scan_plain_mb") e->scan_plain_mb = value;
    else if (k == "query_args") { if (value < 0 || value > 2) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "query_args must be 0, 1 or 2"); e->query_args = value; }
    else if (k == "batch_min") e->batch_min = value;
    else if (k == "batch_mode") e->batch_mode = value;
    else if (k == "batch_rega") { if (value != 0 && value != 1 && value != 5) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_rega must be 0, 1 or 5"); e->batch_rega = value; }
    else if (k == "batch_debug") { if (value & ~(int64_t)(4096 | 16384 | 65536)) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_debug: bits 4096, 16384, 65536 only"); e->batch_debug = value; }
    else if (k == "batch_prof_ptr") e->batch_prof_ptr = value;
    else if (k == "batch_onepass") e->batch_onepass = value;
    else if (k == "batch_onepass_tiles") { if (value < 1024) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_onepass_tiles must be >= 1024"); e->batch_onepass_tiles = value; }
    else if (k == "batch_kp_fused") e->batch_kp_fused = value != 0;
    else if (k == "batch_survivors") { if (value < 2 || value > 64) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_survivors must be 2..64"); e->batch_survivors = value; }
    else if (k == "batch_eps_measured") e->batch_eps_measured = value != 0;
    else if (k == "retry_hint") {   // > 0: the next `value` clean batches still carry the device-side retry kernel (set by collect; tests force it)
        if (value < 0 || value > 1024) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "retry_hint must be 0..1024");
        e->retry_hint = (int)value;
    }
    else if (k == "batch_retry") { if (value < 0 || value > 2) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_retry must be 0, 1 or 2"); e->batch_retry = value; }
    else if (k == "batch_multi") e->batch_multi = value != 0;
    else if (k == "scan_chain") { if (value < -1 || value > 1) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "scan_chain must be -1 (auto), 0 or 1"); e->scan_chain = value; }
    else if (k == "share_timing") e->share_timing = value != 0;   // 0: every chained scan records its own start event (one more packet between scans)
    else if (k == "filter_device_min") { if (value < -1) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "filter_device_min must be >= -1"); e->filter_device_min = value; }
    else if (k == "batch_sample_div") { if (value < 4 || value > 4096) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_sample_div must be 4..4096"); e->batch_sample_div = value; }
    else if (k == "batch_workspaces") {
        if (value < 1 || value > 16) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_workspaces must be 1..16");
        std::unique_lock<std::mutex> g(e->bctx_mu);
        e->bctx_max = (int)value;
    }
    else if (k == "batch_first") { if (value < 128 || value > kBatchFirstSlab || value % 128) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_first must be a multiple of 128 in 128..2048"); e->batch_first = value; }
    else if (k == "batch_growth") { if (value < 1 || value > 64) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_growth must be 1..64"); e->batch_growth = value; }
    else if (k == "batch_slab_mb") { if (value < 1 || value > 4096) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "batch_slab_mb must be 1..4096"); e->batch_slab_mb = value; }
    else if (k == "streams") {
        if (value < 1 || value > kMaxStreams) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "streams must be 1..4");
        std::unique_lock<std::mutex> g(e->slot_mu);
        e->n_streams = (int)value;
    } else if (k == "slots") {
        if (value < 1 || value > 64) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "slots must be 1..64");
        std::unique_lock<std::mutex> g(e->slot_mu);
        e->max_slots = (int)value;
    } else if (k == "reset_stats") {
        e->st_searches = 0; e->st_rows = 0; e->st_bytes = 0;
        {
            DeviceGuard g(e->device);
            for (int r = 0; r < kShardRing; ++r) harvest_ring_event(e, r);
        }
        std::unique_lock<std::mutex> sg(e->st_mu);
        e->st_last_ms = 0; e->st_total_ms = 0; e->st_timed = 0;
        e->st_gemm_ms = 0; e->st_gemm_timed = 0; e->st_gemm_rows = 0; e->st_gemm_queries = 0;
    } else return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "unknown tuning key '" + k + "'");
    return WAX_HIP_OK;
}

int64_t wax_hip_get_tuning(wax_hip_engine* e, const char* key) {
    if (!e || !key) return -1;
    const std::string k(key);
    if (e->sh) return sh_get_tuning(e, k);
    if (k == "grid_blocks") return e->grid_blocks.load();
    if (k == "variant") return e->variant.load();
    if (k == "time_kernels") return e->time_kernels.load();
    if (k == "force_general") return e->force_general.load();
    if (k == "stream_nt") return e->stream_nt.load();
    if (k == "fuse_merge") return e->fuse_merge.load();
    if (k == "query_args") return e->query_args.load();
    if (k == "done_flag") return e->done_flag.load();
    if (k == "merge_kway") return e->merge_kway.load();
    if (k == "merge_overlap_mb") return e->merge_overlap_mb.load();
    if (k == "batch_qfrag") return e->batch_qfrag.load();
    if (k == "overlap_scans") return (int64_t)e->st_overlap_scans.load();
    if (k == "scan_plain_mb") return e->scan_plain_mb.load();
    if (k == "done_flag_waits") return (int64_t)e->st_flag_waits.load();
    if (k == "query_args_scans") return (int64_t)e->st_query_args.load();
    if (k == "merged_scans") return (int64_t)e->st_merged_scans.load();
    if (k == "batch_inline_retries") return (int64_t)e->st_batch_inline_retries.load();
    if (k == "retry_hint") return e->retry_hint.load();
    if (k == "batch_eps_measured") return e->batch_eps_measured.load();
    if (k == "batch_max_row_err_e9" || k == "batch_max_norm_e6") {   // the mirror's measured bounds (device words; a blocking read)
        unsigned int bits[2] = {0u, 0u};
        if (e->batch.d_maxnorm) {
            DeviceGuard g(e->device);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(bits, e->batch.d_maxnorm, sizeof(bits), hipMemcpyDeviceToHost);
        }
        float f[2];
        std::memcpy(f, bits, sizeof(f));
        return k == "batch_max_norm_e6" ? (int64_t)((double)f[0] * 1e6) : (int64_t)((double)f[1] * 1e9);
    }
    if (k == "mirror_rows_converted") return (int64_t)e->batch.rows_converted.load();   // bf16 mirror: rows converted so far / conversions enqueued
    if (k == "mirror_conversions") return (int64_t)e->batch.conversions.load();
    if (k == "idhash_rows_inserted") return (int64_t)e->st_idhash_rows.load();
    if (k == "batch_min") return e->batch_min.load();
    if (k == "batch_mode") return e->batch_mode.load();
    if (k == "batch_rega") return e->batch_rega.load();
    if (k == "batch_slab_mb") return e->batch_slab_mb.load();
    if (k == "batch_growth") return e->batch_growth.load();
    if (k == "batch_first") return e->batch_first.load();
    if (k == "batch_onepass") return e->batch_onepass.load();
    if (k == "batch_onepass_tiles") return e->batch_onepass_tiles.load();
    if (k == "batch_kp_fused") return e->batch_kp_fused.load();
    if (k == "batch_survivors") return e->batch_survivors.load();
    if (k == "batch_sample_div") return e->batch_sample_div.load();
    if (k == "batch_workspaces") return e->bctx_max;
    if (k == "batch_max_k") return kBatchMaxK;
    if (k == "onepass_queries") return (int64_t)e->st_onepass_queries.load();
    if (k == "scan_chain") return e->scan_chain.load();
    if (k == "share_timing") return e->share_timing.load();
    if (k == "batch_retry") return e->batch_retry.load();
    if (k == "batch_retries") return (int64_t)e->st_batch_retries.load();
    if (k == "batch_multi") return e->batch_multi.load();
    if (k == "batch_multi_passes") return (int64_t)e->st_multi_passes.load();
    if (k == "batch_multi_queries") return (int64_t)e->st_multi_queries.load();
    if (k == "batch_multi_group") return (int64_t)scan_multi_group(e->dims, 10);        // queries per shared exact pass, k <= 60 (64-slot lists)
    if (k == "batch_multi_group_big") return (int64_t)scan_multi_group(e->dims, 192);   // ... k <= 192 (256-slot lists)
    if (k == "filter_device_min") return e->filter_device_min.load();
    if (k == "filter_device_searches") return (int64_t)e->st_filter_device.load();
    if (k == "batch_queries") return (int64_t)e->st_batch_queries.load();
    if (k == "batch_fallbacks") return (int64_t)e->st_batch_fallbacks.load();
    if (k == "slots") return e->max_slots;
    if (k == "streams") return e->n_streams;
    if (k == "variant_count") return scan_variant_count(e->dims);
    if (k == "scan_grid") return scan_grid_for((uint32_t)e->count, e->dims, (int)e->variant.load(), (int)e->grid_blocks.load());
    if (k == "fused_max_k") return FUSED_MAX_K;
    if (k == "store_ptr") return (int64_t)(uintptr_t)e->d_store;   // diagnosis: where the slab sits (tools/bimodal_probe.py)
    return -1;
}

int wax_hip_time_scan_kernel(wax_hip_engine* e, const float* query, uint32_t dims, int32_t top_k, uint32_t iters,
                             double* out_avg_ms) {
    if (!e || !out_avg_ms || !query) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    SHARDED_UNSUPPORTED(e, "wax_hip_time_scan_kernel");
    if (dims != e->dims) return fail(WAX_HIP_ERR_DIM_MISMATCH, dim_mismatch_msg(e->dims, dims));
    if (iters == 0) iters = 1;
    DeviceGuard g(e->device);
    e->lock.lock_shared(holding(e) > 0);
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) { e->lock.unlock_shared(); return frc; } }
    Slot* s = nullptr;
    int rc = WAX_HIP_OK;
    do {
        if (e->count == 0) { rc = fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is empty"); break; }
        const int limit = clamp_topk(top_k);
        const int k_eff = (uint64_t)limit < e->count ? limit : (int)e->count;
        if (k_eff > FUSED_MAX_K) { rc = fail(WAX_HIP_ERR_INVALID_ARGUMENT, "time_scan_kernel: top_k must be <= 192"); break; }
        rc = acquire_slot(e, &s);
        if (rc != WAX_HIP_OK) break;
        std::memcpy(s->h_query, query, (size_t)dims * sizeof(float));
        ScanArgs a{};
        a.store = e->d_store; a.query = s->d_query; a.partials = s->d_partials; a.dist_out = nullptr;
        a.n_rows = (uint32_t)e->count; a.row_base = (uint32_t)e->row_base; a.dims = e->dims; a.k = k_eff;
        a.q_norm = query_norm(query, dims);
        const int cap = k_eff <= 64 ? 128 : 256;
        hipError_t err = hipMemcpyAsync(s->d_query, s->h_query, (size_t)dims * sizeof(float), hipMemcpyHostToDevice, s->stream);
        int grid = 0;
        for (int wu = 0; wu < 2 && err == hipSuccess; ++wu)
            err = launch_scan(a, e->metric, (int)e->variant.load(), cap, false, (int)e->grid_blocks.load(), s->stream, &grid);
        if (err == hipSuccess) err = hipEventRecord(s->ev0, s->stream);
        for (uint32_t i = 0; i < iters && err == hipSuccess; ++i)
            err = launch_scan(a, e->metric, (int)e->variant.load(), cap, false, (int)e->grid_blocks.load(), s->stream, &grid);
        if (err == hipSuccess) err = hipEventRecord(s->ev1, s->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, s->ev0, s->ev1);
        if (err != hipSuccess) { rc = fail(WAX_HIP_ERR_INTERNAL, std::string("time_scan_kernel: ") + hipGetErrorString(err)); break; }
        *out_avg_ms = (double)ms / (double)iters;
    } while (0);
    if (s) release_slot(e, s);
    e->lock.unlock_shared();
    return rc;
}

int wax_hip_time_stream_read(wax_hip_engine* e, uint32_t iters, double* out_avg_ms) {
    if (!e || !out_avg_ms) return fail(WAX_HIP_ERR_INVALID_ARGUMENT, "null argument");
    SHARDED_UNSUPPORTED(e, "wax_hip_time_stream_read");
    if (iters == 0) iters = 1;
    DeviceGuard g(e->device);
    e->lock.lock_shared(holding(e) > 0);
    { const int frc = flush_pending(e); if (frc != WAX_HIP_OK) { e->lock.unlock_shared(); return frc; } }
    Slot* s = nullptr;
    int rc = WAX_HIP_OK;
    do {
        const uint64_t bytes = (e->count * (uint64_t)e->dims * 4ull) & ~15ull;
        if (bytes == 0) { rc = fail(WAX_HIP_ERR_INVALID_ARGUMENT, "engine is empty"); break; }
        rc = acquire_slot(e, &s);
        if (rc != WAX_HIP_OK) break;
        int grid = (int)e->grid_blocks.load();
        if (grid <= 0) grid = 2048;
        if (grid > MAX_GRID_BLOCKS) grid = MAX_GRID_BLOCKS;
        const int nt = (int)e->stream_nt.load();
        hipError_t err = hipSuccess;
        for (int wu = 0; wu < 2 && err == hipSuccess; ++wu) err = launch_stream_read(e->d_store, bytes, nt, grid, e->d_sink, s->stream);
        if (err == hipSuccess) err = hipEventRecord(s->ev0, s->stream);
        for (uint32_t i = 0; i < iters && err == hipSuccess; ++i) err = launch_stream_read(e->d_store, bytes, nt, grid, e->d_sink, s->stream);
        if (err == hipSuccess) err = hipEventRecord(s->ev1, s->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, s->ev0, s->ev1);
        if (err != hipSuccess) { rc = fail(WAX_HIP_ERR_INTERNAL, std::string("time_stream_read: ") + hipGetErrorString(err)); break; }
        *out_avg_ms = (double)ms / (double)iters;
    } while (0);
    if (s) release_slot(e, s);
    e->lock.unlock_shared();
    return rc;
}

}  // extern "C"

#include "sharded.inc"
