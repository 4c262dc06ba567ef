// filter.hip — device side of the allow-list pre-filter of wax_hip_search_filtered (SURVEY.md §8f-4;
// FrameFilter.frameIds, UnifiedSearch.swift:1241-1258): frame ids -> rows of the store without a host probe per id.
//
//   idhash_build_kernel    id -> row open-addressing table in HBM (u32 slots holding row+1, 0 = empty; linear probing).
//                          Rebuilt lazily after a mutation, like the bf16 mirror. 4 bytes per slot, load factor <= 0.5.
//   idhash_probe_kernel    one thread per allowed id: probe, mark the row in a bitmap (atomicOr). The bitmap hands the
//                          rows back ascending and unique — the order every selection path breaks ties in.
//   bitmap_count / bitmap_scan / bitmap_emit   bitmap -> compact ascending row list + the rows' frame ids.
//
// All HBM-latency work: at a 1M-id allow-list the probes are ~2M random 64-byte sector reads, a few tens of µs on the
// device against 5.65 ms of cache-missing host probes (profiles/r01).
#include "kernels.h"

namespace wax {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__global__ __launch_bounds__(256) void idhash_build_kernel(const uint64_t* __restrict__ ids, uint32_t row0, uint32_t n, uint32_t* table,
                                                           uint32_t mask) {
    const uint32_t row = row0 + blockIdx.x * 256u + threadIdx.x;     // rows [row0, n): the appended ones, or all of them (row0 = 0)
    if (row >= n) return;
    uint32_t h = (uint32_t)mix64(ids[row]) & mask;
    while (atomicCAS(&table[h], 0u, row + 1u) != 0u) h = (h + 1u) & mask;
}

__global__ __launch_bounds__(256) void idhash_probe_kernel(const uint64_t* __restrict__ allow, uint64_t n_allow,
                                                           const uint64_t* __restrict__ ids, const uint32_t* __restrict__ table,
                                                           uint32_t mask, uint32_t* bitmap) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_allow) return;
    const uint64_t id = allow[i];
    uint32_t h = (uint32_t)mix64(id) & mask;
    // A store loaded from a segment may hold one frame id in several rows (deserialize keeps them as separate rows and the
    // host id map resolves the id to the FIRST of them, MetalVectorEngine.swift:809-811 / firstIndex(of:)). All rows of
    // one id sit in the same probe chain, in whatever order the build's CAS races left them: walk the chain to its end
    // and mark the LOWEST matching row, so the device path picks the row the host path picks, on every run.
    uint32_t best = 0xffffffffu;
    for (;;) {
        const uint32_t s = table[h];
        if (s == 0u) break;   // end of the chain (an allowed id without a vector is simply absent, UnifiedSearch.swift:1243)
        if (ids[s - 1u] == id && s - 1u < best) best = s - 1u;
        h = (h + 1u) & mask;
    }
    if (best != 0xffffffffu) atomicOr(&bitmap[best >> 5], 1u << (best & 31u));
}

constexpr int kWordsPerThread = 4;
constexpr int kWordsPerBlock = 256 * kWordsPerThread;

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wave_sum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sum[w] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < w) base += wave_sum[j];
        all += wave_sum[j];
    }
    *total = all;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint32_t* __restrict__ bitmap, uint32_t n_words,
                                                           uint32_t* block_sum) {
    const uint32_t w0 = blockIdx.x * kWordsPerBlock + threadIdx.x * kWordsPerThread;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < kWordsPerThread; ++j)
        if (w0 + j < n_words) c += __popc(bitmap[w0 + j]);
    uint32_t total;
    (void)block_exclusive_scan_256(c, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

// One workgroup: exclusive scan of block_sum in place, grand total to *total.
__global__ __launch_bounds__(256) void bitmap_scan_kernel(uint32_t* block_sum, uint32_t n_blocks, uint32_t* total) {
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256u) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sum[i] : 0u;
        uint32_t chunk;
        const uint32_t ex = block_exclusive_scan_256(v, &chunk);
        if (i < n_blocks) block_sum[i] = carry + ex;
        carry += chunk;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void bitmap_emit_kernel(const uint32_t* __restrict__ bitmap, uint32_t n_words,
                                                          const uint32_t* __restrict__ block_off, const uint64_t* __restrict__ ids,
                                                          uint32_t* rows_out, uint64_t* ids_out) {
    const uint32_t w0 = blockIdx.x * kWordsPerBlock + threadIdx.x * kWordsPerThread;
    uint32_t words[kWordsPerThread];
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < kWordsPerThread; ++j) {
        words[j] = w0 + j < n_words ? bitmap[w0 + j] : 0u;
        c += __popc(words[j]);
    }
    uint32_t total;
    uint32_t pos = block_off[blockIdx.x] + block_exclusive_scan_256(c, &total);
#pragma unroll
    for (int j = 0; j < kWordsPerThread; ++j) {
        uint32_t w = words[j];
        while (w) {
            const uint32_t row = (w0 + j) * 32u + (uint32_t)__builtin_ctz(w);
            rows_out[pos] = row;
            ids_out[pos] = ids[row];
            ++pos;
            w &= w - 1u;
        }
    }
}

hipError_t launch_idhash_build(const uint64_t* ids, uint32_t row0, uint32_t n, uint32_t* table, uint64_t slots, hipStream_t st) {
    hipError_t err = hipSuccess;
    if (row0 == 0) err = hipMemsetAsync(table, 0, (size_t)slots * sizeof(uint32_t), st);   // from scratch; row0 > 0 inserts into the live table
    if (err != hipSuccess || n <= row0) return err;
    hipLaunchKernelGGL(idhash_build_kernel, dim3((n - row0 + 255u) / 256u), dim3(256), 0, st, ids, row0, n, table, (uint32_t)(slots - 1));
    return hipGetLastError();
}

uint32_t filter_bitmap_blocks(uint32_t n_rows) {
    const uint32_t n_words = (n_rows + 31u) / 32u;
    return (n_words + (uint32_t)kWordsPerBlock - 1u) / (uint32_t)kWordsPerBlock;
}

hipError_t launch_allow_probe(const uint64_t* d_allow, uint64_t n_allow, const uint64_t* ids, uint32_t n_rows,
                              const uint32_t* table, uint64_t slots, uint32_t* bitmap, uint32_t* block_sum, uint32_t* total,
                              hipStream_t st) {
    const uint32_t n_words = (n_rows + 31u) / 32u;
    const uint32_t n_blocks = filter_bitmap_blocks(n_rows);
    hipError_t err = hipMemsetAsync(bitmap, 0, (size_t)n_words * sizeof(uint32_t), st);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(idhash_probe_kernel, dim3((unsigned)((n_allow + 255u) / 256u)), dim3(256), 0, st, d_allow, n_allow, ids,
                       table, (uint32_t)(slots - 1), bitmap);
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(n_blocks), dim3(256), 0, st, bitmap, n_words, block_sum);
    hipLaunchKernelGGL(bitmap_scan_kernel, dim3(1), dim3(256), 0, st, block_sum, n_blocks, total);
    return hipGetLastError();
}

hipError_t launch_allow_emit(const uint32_t* bitmap, uint32_t n_rows, const uint32_t* block_off, const uint64_t* ids,
                             uint32_t* rows_out, uint64_t* ids_out, hipStream_t st) {
    const uint32_t n_words = (n_rows + 31u) / 32u;
    hipLaunchKernelGGL(bitmap_emit_kernel, dim3(filter_bitmap_blocks(n_rows)), dim3(256), 0, st, bitmap, n_words, block_off, ids,
                       rows_out, ids_out);
    return hipGetLastError();
}

}  // namespace wax
