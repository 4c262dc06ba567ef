/* Single-query latency through the C ABI, one blocking wax_hip_search at a time — no Python, no ctypes between calls.
 * The reference harness shape (Tests/WaxIntegrationTests/MetalVectorEngineBenchmark.swift:65-128: 10 000 x 384, top-24,
 * vector[d] = ((i + d) % 256) / 255, 10 warm searches) and the bench's s10k shape (unit Gaussian rows, top-10), for each
 * value of the "query_args" tunable. Prints one JSON line per (corpus, query_args) with the mean / median / p99 latency.
 *   gcc -O2 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lwax_amd/lib -lwaxhip -Wl,-rpath,$PWD/wax_amd/lib -lm */
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
static double gauss(unsigned long long* s) {   /* Box-Muller over a 64-bit LCG: any fixed corpus will do */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    const double u1 = ((*s >> 11) + 1.0) / 9007199254740993.0;
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    const double u2 = (*s >> 11) / 9007199254740992.0;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10000, dims = argc > 2 ? atoi(argv[2]) : 384, reps = argc > 3 ? atoi(argv[3]) : 2000;
    if (!wax_hip_available()) { printf("{\"error\": \"no gfx950 device\"}\n"); return 0; }
    float* rows = malloc((size_t)n * dims * sizeof(float));
    uint64_t* ids = malloc((size_t)n * sizeof(uint64_t));
    float* q = malloc((size_t)dims * sizeof(float));
    double* lat = malloc((size_t)reps * sizeof(double));
    for (int corpus = 0; corpus < 2; ++corpus) {
        const int topk = corpus == 0 ? 24 : (argc > 4 ? atoi(argv[4]) : 10);   /* argv[4]: top_k of the unit-gaussian part (<= 64) */
        unsigned long long seed = 12345;
        for (int i = 0; i < n; ++i) {
            ids[i] = (uint64_t)i;
            double nrm = 0.0;
            for (int d = 0; d < dims; ++d) {
                const float v = corpus == 0 ? (float)((i + d) % 256) / 255.0f : (float)gauss(&seed);
                rows[(size_t)i * dims + d] = v;
                nrm += (double)v * v;
            }
            if (corpus == 1) for (int d = 0; d < dims; ++d) rows[(size_t)i * dims + d] /= (float)sqrt(nrm);
        }
        for (int d = 0; d < dims; ++d) q[d] = corpus == 0 ? (float)rand() / (float)RAND_MAX : rows[(size_t)5 * dims + d];
        wax_hip_engine* e = NULL;
        if (wax_hip_engine_create(WAX_HIP_METRIC_COSINE, (uint32_t)dims, -1, &e) || wax_hip_add_batch(e, ids, rows, (uint64_t)n, (uint32_t)dims)) {
            printf("{\"error\": \"%s\"}\n", wax_hip_last_error());
            return 1;
        }
        uint64_t out_ids[64], first[64];
        float out_scores[64];
        uint32_t got = 0;
        /* modes 0..2 = "query_args" 0 / 1 / 2 on the automatic grid; 3 = query_args 1 with at most 80 scan workgroups (fewer, fatter
         * workgroups: fewer partial lists for the fused final merge, more rows per wave); 4 = query_args 1, "done_flag" 0;
         * 5 = query_args 1, "merge_kway" 0 (the last workgroup streams the partial lists through its wave lists instead of merging their heads) */
        /* 6 = query_args 1 with ordinary (cacheable) row loads whatever the store size ("scan_plain_mb" = 1 << 20) */
        /* 7 = query_args 1 with non-temporal row loads whatever the store size ("scan_plain_mb" = 0) */
        for (int mode = 0; mode <= 7; ++mode) {
            wax_hip_set_tuning(e, "scan_plain_mb", mode == 6 ? (1 << 20) : mode == 7 ? 0 : -1);
            wax_hip_set_tuning(e, "query_args", mode <= 2 ? mode : 1);
            wax_hip_set_tuning(e, "grid_blocks", mode == 3 ? (getenv("WAX_LAT_GRID") ? atoi(getenv("WAX_LAT_GRID")) : 80) : 0);   /* WAX_LAT_GRID: mode 3's grid cap */
            wax_hip_set_tuning(e, "merge_kway", mode == 5 ? 0 : 1);
            wax_hip_set_tuning(e, "done_flag", mode == 4 ? 0 : 1);   /* mode 4: query_args 1 with an event behind the kernel instead of the completion word */
            for (int i = 0; i < 50; ++i) wax_hip_search(e, q, (uint32_t)dims, topk, out_ids, out_scores, 64, &got);
            if (mode == 0) for (uint32_t i = 0; i < got; ++i) first[i] = out_ids[i];
            int same = 1;
            for (uint32_t i = 0; i < got; ++i) same = same && first[i] == out_ids[i];
            const int64_t before = wax_hip_get_tuning(e, "query_args_scans");
            const double t0 = now_us();
            for (int i = 0; i < reps; ++i) {
                const double a = now_us();
                if (wax_hip_search(e, q, (uint32_t)dims, topk, out_ids, out_scores, 64, &got)) { printf("{\"error\": \"%s\"}\n", wax_hip_last_error()); return 1; }
                lat[i] = now_us() - a;
            }
            const double mean = (now_us() - t0) / reps;
            qsort(lat, (size_t)reps, sizeof(double), cmp);
            printf("{\"tool\": \"latency_c\", \"corpus\": \"%s\", \"rows\": %d, \"dims\": %d, \"top_k\": %d, \"mode\": %d, \"scan_grid\": %lld, "
                   "\"scans_with_query_in_kernel_args\": %lld, \"reps\": %d, \"mean_us\": %.2f, \"median_us\": %.2f, \"p99_us\": %.2f, "
                   "\"min_us\": %.2f, \"hits\": %u, \"same_ids_as_query_args_0\": %s}\n",
                   corpus == 0 ? "reference harness ((i + d) % 256) / 255" : "unit gaussian", n, dims, topk, mode, (long long)wax_hip_get_tuning(e, "scan_grid"),
                   (long long)(wax_hip_get_tuning(e, "query_args_scans") - before), reps, mean, lat[reps / 2], lat[(int)(reps * 0.99)],
                   lat[0], got, same ? "true" : "false");
        }
        wax_hip_engine_destroy(e);
    }
    return 0;
}
