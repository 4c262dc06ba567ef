// ref_metal.cpp — runs the reference's OWN Metal compute shaders on the CPU.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref): nothing under wax_amd/ may use it. It is built only where the reference checkout
// exists (this container), by oracle/ref_metal/Makefile, into oracle/_ref/libwaxref_metal.so, and never reads /root/reference at
// run time. No reference SOURCE is copied into this repository: the two shader files are #included from where they lie
//   /root/reference/Sources/WaxVectorSearch/Shaders/CosineDistance.metal   (a3: cosineDistanceKernelSIMD4 / SIMD8, :152-328)
//   /root/reference/Sources/WaxVectorSearch/Shaders/TopKReduction.metal    (a4: topKReduceDistances / topKReduceEntries, :103-167)
// and compiled as C++14 against oracle/ref_metal/msl/metal_stdlib (a stand-in for Apple's header). What this file adds is the part
// of a GPU that a CPU lacks — a grid of threadgroups whose threads meet at threadgroup_barrier() — and a restatement of the HOST
// code that encodes the dispatches (MetalVectorEngine.swift:494-585: threadgroup sizes, threadgroup memory, the reduction loop),
// which is Swift and cannot be compiled here.
//
// Threads of a threadgroup are ucontext fibers run round-robin on one OS thread: a fiber runs until it returns or reaches a
// barrier; when every live fiber of the group waits at the barrier, all are released. A thread that returned early (the
// shaders' `if (vectorIndex >= vectorCount) return;` ahead of the barrier) counts as arrived — what Apple GPUs do in practice.
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include <metal_stdlib>

#include "CosineDistance.metal"
#include "TopKReduction.metal"

#undef kernel
#undef device
#undef constant
#undef threadgroup

namespace {

enum { RUNNABLE = 0, AT_BARRIER = 1, DONE = 2 };
struct Fiber {
    ucontext_t ctx;
    int state;
    uint32_t tid;
};
ucontext_t g_sched;
Fiber* g_cur = nullptr;
const std::function<void(uint32_t)>* g_body = nullptr;
constexpr size_t kStack = 64 * 1024;

void fiber_entry() {
    Fiber* f = g_cur;
    (*g_body)(f->tid);
    f->state = DONE;   // uc_link returns to the scheduler
}

// one threadgroup of `tg_size` threads: body(tid) is the kernel call for that thread
void run_threadgroup(uint32_t tg_size, const std::function<void(uint32_t)>& body) {
    static std::vector<Fiber> fibers;
    static std::vector<char> stacks;
    if (fibers.size() < tg_size) {
        fibers.resize(tg_size);
        stacks.resize((size_t)tg_size * kStack);
    }
    g_body = &body;
    for (uint32_t t = 0; t < tg_size; ++t) {
        Fiber& f = fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)t * kStack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &g_sched;
        f.state = RUNNABLE;
        f.tid = t;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    for (;;) {
        for (uint32_t t = 0; t < tg_size; ++t) {
            if (fibers[t].state != RUNNABLE) continue;
            g_cur = &fibers[t];
            swapcontext(&g_sched, &fibers[t].ctx);   // back here at a barrier or at the end of the thread
        }
        bool waiting = false;
        for (uint32_t t = 0; t < tg_size; ++t)
            if (fibers[t].state == AT_BARRIER) { fibers[t].state = RUNNABLE; waiting = true; }
        if (!waiting) break;
    }
    g_body = nullptr;
    g_cur = nullptr;
}

}  // namespace

void metal::threadgroup_barrier(metal::mem_flags) {
    Fiber* f = g_cur;
    f->state = AT_BARRIER;
    swapcontext(&f->ctx, &g_sched);
}

#define WAXREF_API extern "C" __attribute__((visibility("default")))

// MetalVectorEngine.swift:20 (maxThreadsPerThreadgroup = 256), :499-507 (one thread per row, ceil(n / 256) threadgroups),
// :493-494 (threadgroup memory = dimensions * 4 bytes). simd8 != 0: cosineDistanceKernelSIMD8 (the engine's choice for
// D >= 384, :185, :496), else cosineDistanceKernelSIMD4. Returns 0.
WAXREF_API int waxref_metal_cosine_distances(int simd8, const float* vectors, const float* query, uint32_t n, uint32_t dims, float* out) {
    const uint32_t tg = 256;
    std::vector<metal::float4> shared((size_t)dims / 4 + 2);
    const uint32_t groups = (n + tg - 1) / tg;
    for (uint32_t g = 0; g < groups; ++g) {
        // threadgroup memory is undefined at the start of a threadgroup: poison it, so that a read of something the kernel
        // never stored shows up as NaN instead of silently reusing the previous threadgroup's (identical) query
        for (auto& s4 : shared) s4 = metal::float4(std::nanf(""));
        run_threadgroup(tg, [&](uint32_t tid) {
            const uint32_t gid = g * tg + tid;
            if (simd8) cosineDistanceKernelSIMD8(vectors, query, out, n, dims, shared.data(), gid, tid, tg);
            else cosineDistanceKernelSIMD4(vectors, query, out, n, dims, shared.data(), gid, tid, tg);
        });
    }
    return 0;
}

// The GPU top-k of MetalVectorEngine.swift:517-585: topKReduceDistances over ceil(n / T) threadgroups of T = 256 threads
// (reductionThreadgroupSize, :848-855: the largest power of two <= min(maxTotalThreadsPerThreadgroup, 256)), each leaving k
// entries; then topKReduceEntries over the concatenated entries until no more than k remain. The caller guarantees the
// engine's precondition (:455): n >= 1000 and k <= 256. out_dist / out_idx: k entries, ascending by distance; padding entries
// are (+inf, 0xFFFFFFFF). out_passes: dispatches encoded (the "2-6 launches" of SURVEY.md a4). Returns 0, -1 if k > 256, or
// -2 when the host loop `while currentCount > topKCount` (:548) makes no progress: ceil(count / 256) * k == count has fixed
// points above k for every k > 128 (k = 200: 800 -> 4 groups -> 800; k = 256: never shrinks at all), where the reference would
// allocate merge buffers forever. Its callers pass candidateLimit = 30 (UnifiedSearch.swift:1195-1200), far below that.
WAXREF_API int waxref_metal_topk(const float* distances, uint32_t n, uint32_t k, float* out_dist, uint32_t* out_idx, uint32_t* out_passes) {
    const uint32_t T = 256;
    if (k == 0 || k > T || n == 0) return -1;
    std::vector<TopKEntry> shared(T);
    uint32_t groups = (n + T - 1) / T;
    std::vector<TopKEntry> cur((size_t)groups * k), next;
    for (uint32_t g = 0; g < groups; ++g)
        run_threadgroup(T, [&](uint32_t tid) { topKReduceDistances(distances, n, k, cur.data(), shared.data(), tid, g, T); });
    uint32_t passes = 1;
    uint32_t count = groups * k;
    while (count > k) {
        const uint32_t ng = (count + T - 1) / T;
        if (ng * k >= count) return -2;   // the reference's loop would not terminate
        next.assign((size_t)ng * k, TopKEntry{INFINITY, 0xFFFFFFFFu});
        for (uint32_t g = 0; g < ng; ++g)
            run_threadgroup(T, [&](uint32_t tid) { topKReduceEntries(cur.data(), count, k, next.data(), shared.data(), tid, g, T); });
        cur.swap(next);
        count = ng * k;
        ++passes;
    }
    for (uint32_t i = 0; i < k; ++i) {
        out_dist[i] = cur[i].distance;
        out_idx[i] = cur[i].index;
    }
    if (out_passes) *out_passes = passes;
    return 0;
}
