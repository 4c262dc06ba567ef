// wax_hip.hpp — header-only C++17 convenience wrapper over the C ABI in wax_hip.h.
// Mirrors the reference's VectorSearchEngine protocol (VectorSearchEngine.swift:10-18) for C++ hosts.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "wax_hip.h"

namespace wax_hip {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

inline void check(int rc) {
    if (rc != WAX_HIP_OK) throw Error(rc, wax_hip_last_error());
}

class VectorEngine {
  public:
    static bool isAvailable() { return wax_hip_available() != 0; }
    VectorEngine(wax_hip_metric metric, uint32_t dimensions, int device = -1) {
        check(wax_hip_engine_create((uint8_t)metric, dimensions, device, &h_));
    }
    ~VectorEngine() { wax_hip_engine_destroy(h_); }
    VectorEngine(const VectorEngine&) = delete;
    VectorEngine& operator=(const VectorEngine&) = delete;

    uint32_t dimensions() const { return wax_hip_dimensions(h_); }
    uint64_t count() const { return wax_hip_count(h_); }

    std::vector<std::pair<uint64_t, float>> search(const std::vector<float>& vector, int topK) {
        // sized by topK alone (never by a row count read outside the engine's lock) and passed as the capacity
        const uint32_t cap = wax_hip_result_capacity(topK);
        std::vector<uint64_t> ids(cap);
        std::vector<float> scores(cap);
        uint32_t got = 0;
        check(wax_hip_search(h_, vector.data(), (uint32_t)vector.size(), topK, ids.data(), scores.data(), cap, &got));
        std::vector<std::pair<uint64_t, float>> out(got);
        for (uint32_t i = 0; i < got; ++i) out[i] = {ids[i], scores[i]};
        return out;
    }
    void add(uint64_t frameId, const std::vector<float>& v) { check(wax_hip_add(h_, frameId, v.data(), (uint32_t)v.size())); }
    /// Allow-list / minScore filtered search (UnifiedSearch.swift:1241-1258): best topK among the allowed frames.
    std::vector<std::pair<uint64_t, float>> searchFiltered(const std::vector<float>& q, int topK,
                                                           const std::vector<uint64_t>* allow, const float* minScore) {
        const uint32_t cap = wax_hip_result_capacity(topK);
        std::vector<uint64_t> ids(cap);
        std::vector<float> scores(cap);
        uint32_t n = 0;
        check(wax_hip_search_filtered(h_, q.data(), (uint32_t)q.size(), topK, allow ? 1 : 0,
                                      allow && !allow->empty() ? allow->data() : nullptr, allow ? allow->size() : 0,
                                      minScore ? 1 : 0, minScore ? *minScore : 0.0f, ids.data(), scores.data(), cap, &n));
        std::vector<std::pair<uint64_t, float>> out(n);
        for (uint32_t i = 0; i < n; ++i) out[i] = {ids[i], scores[i]};
        return out;
    }
    /// Pending-embedding replay: WAL putEmbedding payloads back to back (UnifiedSearchEngineCache.swift:252-283).
    uint64_t applyPutEmbeddings(const uint8_t* payloads, uint64_t len) {
        uint64_t applied = 0;
        check(wax_hip_apply_put_embeddings(h_, payloads, len, &applied));
        return applied;
    }
    void addBatch(const std::vector<uint64_t>& frameIds, const std::vector<float>& rowsRowMajor) {
        if (frameIds.empty()) return;
        check(wax_hip_add_batch(h_, frameIds.data(), rowsRowMajor.data(), frameIds.size(),
                                (uint32_t)(rowsRowMajor.size() / frameIds.size())));
    }
    void remove(uint64_t frameId) { check(wax_hip_remove(h_, frameId)); }
    std::vector<uint8_t> serialize() {
        uint8_t* p = nullptr;
        size_t len = 0;
        check(wax_hip_serialize(h_, &p, &len));
        std::vector<uint8_t> out(p, p + len);
        wax_hip_free(p);
        return out;
    }
    void deserialize(const std::vector<uint8_t>& bytes) { check(wax_hip_deserialize(h_, bytes.data(), bytes.size())); }
    wax_hip_engine* raw() { return h_; }

  private:
    wax_hip_engine* h_ = nullptr;
};

}  // namespace wax_hip
