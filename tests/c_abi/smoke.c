/* Plain-C consumer of include/wax_hip.h: proves the boundary needs nothing but a C compiler and the
 * shared library (no torch, no C++, no HIP headers). With a GPU it runs the reference's 2-d upsert case
 * (VectorSearchEngineTests.swift:36-47); without one it checks that the library fails loudly. */
#include <stdio.h>
#include <string.h>

#include "wax_hip.h"

int main(void) {
    if (wax_hip_abi_version() != WAX_HIP_ABI_VERSION) { printf("abi mismatch\n"); return 2; }
    wax_hip_engine* e = NULL;
    int rc = wax_hip_engine_create(WAX_HIP_METRIC_COSINE, 2, -1, &e);
    if (!wax_hip_available()) {
        if (rc != WAX_HIP_ERR_NO_DEVICE || e != NULL || strlen(wax_hip_last_error()) == 0) { printf("expected a loud NO_DEVICE\n"); return 3; }
        printf("no-device: %s\n", wax_hip_last_error());
        return 0;
    }
    if (rc != WAX_HIP_OK) { printf("create failed: %s\n", wax_hip_last_error()); return 4; }
    const float a[2] = {1.f, 0.f}, b[2] = {0.f, 1.f}, c[2] = {0.7f, 0.7f}, bad[3] = {1.f, 2.f, 3.f};
    const uint64_t id20 = 20;
    if (wax_hip_add(e, 10, a, 2) || wax_hip_add(e, 20, b, 2) || wax_hip_add_batch(e, &id20, c, 1, 2)) { printf("add failed: %s\n", wax_hip_last_error()); return 5; }
    uint64_t ids[2]; float scores[2]; uint32_t n = 0;
    if (wax_hip_search(e, c, 2, 1, ids, scores, 2, &n) || n != 1 || ids[0] != 20) { printf("search wrong: n=%u id=%llu %s\n", n, (unsigned long long)ids[0], wax_hip_last_error()); return 6; }
    /* the capacity bounds what is written: top_k 2 into a 1-entry array returns the single best */
    ids[1] = 777;
    if (wax_hip_search(e, c, 2, 2, ids, scores, 1, &n) || n != 1 || ids[0] != 20 || ids[1] != 777) { printf("capacity not honoured\n"); return 10; }
    if (wax_hip_result_capacity(0) != 1 || wax_hip_result_capacity(50000) != WAX_HIP_MAX_RESULTS) return 11;
    if (wax_hip_search(e, bad, 3, 1, ids, scores, 2, &n) != WAX_HIP_ERR_DIM_MISMATCH ||
        strcmp(wax_hip_last_error(), "vector dimension mismatch: expected 2, got 3") != 0) { printf("bad error: %s\n", wax_hip_last_error()); return 7; }
    uint8_t* blob = NULL; size_t len = 0;
    if (wax_hip_serialize(e, &blob, &len) || len != 36 + 2 * 2 * 4 + 8 + 2 * 8 || memcmp(blob, "MV2V", 4) != 0) { printf("serialize wrong\n"); return 8; }
    wax_hip_free(blob);
    if (wax_hip_count(e) != 2 || wax_hip_dimensions(e) != 2) return 9;
    wax_hip_engine_destroy(e);
    printf("c-abi ok: id %llu score %.6f\n", (unsigned long long)ids[0], scores[0]);
    return 0;
}
