/* Blocking host-pointer batched calls (wax_hip_search_batch_hits, 256 queries) through the C ABI, no Python in the process:
 * per-call latency, the slowest calls and their indices — does the 37-43 ms call that tools/blocking_batch_timeline.py --host
 * sees at a fixed call index belong to the library (or to HIP under it) or to the Python / torch process around it?
 *   gcc -O2 -Iinclude tools/batch_stall_probe.c -o /tmp/batch_stall_probe -Lwax_amd/lib -lwaxhip -Wl,-rpath,$PWD/wax_amd/lib -lm
 *   /tmp/batch_stall_probe [rows=1000000] [dims=384] [nq=256] [calls=600] [batch_onepass_tiles] [top_k=10] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "wax_hip.h"

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
static unsigned long long lcg;
static float rnd(void) {   /* sum of four uniforms, centred: close enough to a Gaussian for a corpus */
    float s = 0.f;
    for (int i = 0; i < 4; ++i) { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; s += (float)(lcg >> 40) / 16777216.0f; }
    return s - 2.0f;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, dims = argc > 2 ? atoi(argv[2]) : 384, nq = argc > 3 ? atoi(argv[3]) : 256;
    const int calls = argc > 4 ? atoi(argv[4]) : 600, k = argc > 6 ? atoi(argv[6]) : 10;
    if (!wax_hip_available()) { printf("{\"error\": \"no gfx950 device\"}\n"); return 0; }
    lcg = 99;
    wax_hip_engine* e = NULL;
    if (wax_hip_engine_create(WAX_HIP_METRIC_COSINE, (uint32_t)dims, -1, &e)) { printf("{\"error\": \"%s\"}\n", wax_hip_last_error()); return 1; }
    const int chunk = 50000;
    float* rows = malloc((size_t)chunk * dims * sizeof(float));
    uint64_t* ids = malloc((size_t)chunk * sizeof(uint64_t));
    for (int r0 = 0; r0 < n; r0 += chunk) {
        const int m = n - r0 < chunk ? n - r0 : chunk;
        for (int i = 0; i < m; ++i) {
            ids[i] = (uint64_t)(r0 + i);
            double nrm = 0.0;
            for (int d = 0; d < dims; ++d) { const float v = rnd(); rows[(size_t)i * dims + d] = v; nrm += (double)v * v; }
            for (int d = 0; d < dims; ++d) rows[(size_t)i * dims + d] /= (float)sqrt(nrm);
        }
        if (wax_hip_add_batch(e, ids, rows, (uint64_t)m, (uint32_t)dims)) { printf("{\"error\": \"%s\"}\n", wax_hip_last_error()); return 1; }
    }
    if (argc > 5 && wax_hip_set_tuning(e, "batch_onepass_tiles", atoi(argv[5]))) { printf("{\"error\": \"%s\"}\n", wax_hip_last_error()); return 1; }
    float* q = malloc((size_t)nq * dims * sizeof(float));
    for (size_t i = 0; i < (size_t)nq * dims; ++i) q[i] = rnd();
    wax_hip_hit* hits = malloc((size_t)nq * k * sizeof(wax_hip_hit));
    uint32_t* counts = malloc((size_t)nq * sizeof(uint32_t));
    double* lat = malloc((size_t)calls * sizeof(double));
    for (int i = 0; i < 5; ++i)
        if (wax_hip_search_batch_hits(e, q, (uint32_t)nq, (uint32_t)dims, k, hits, k, counts)) { printf("{\"error\": \"%s\"}\n", wax_hip_last_error()); return 1; }
    for (int i = 0; i < calls; ++i) {
        const double t0 = now_us();
        if (wax_hip_search_batch_hits(e, q, (uint32_t)nq, (uint32_t)dims, k, hits, k, counts)) { printf("{\"error\": \"%s\"}\n", wax_hip_last_error()); return 1; }
        lat[i] = now_us() - t0;
    }
    int w[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) {
        int best = -1;
        for (int i = 0; i < calls; ++i) {
            int used = 0;
            for (int t = 0; t < j; ++t) used |= w[t] == i;
            if (!used && (best < 0 || lat[i] > lat[best])) best = i;
        }
        w[j] = best;
    }
    double sum = 0.0;
    for (int i = 0; i < calls; ++i) sum += lat[i];
    printf("{\"tool\": \"batch_stall_probe\", \"rows\": %d, \"dims\": %d, \"nq\": %d, \"calls\": %d, \"top_k\": %d, \"onepass_tiles\": %d, \"onepass_queries\": %lld, \"mean_us\": %.1f, \"slowest\": [[%d, %.0f], [%d, %.0f], [%d, %.0f], [%d, %.0f]], ",
           n, dims, nq, calls, k, (int)wax_hip_get_tuning(e, "batch_onepass_tiles"), (long long)wax_hip_get_tuning(e, "onepass_queries"), sum / calls, w[0], lat[w[0]], w[1], lat[w[1]], w[2], lat[w[2]], w[3], lat[w[3]]);
    qsort(lat, (size_t)calls, sizeof(double), cmp);
    printf("\"median_us\": %.1f, \"p99_us\": %.1f, \"mixed_hits\": %u}\n", lat[calls / 2], lat[(int)(calls * 0.99)], counts[0]);
    wax_hip_engine_destroy(e);
    return 0;
}
