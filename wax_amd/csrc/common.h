// common.h — shared host/device definitions for libwaxhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/wax_hip.h"

namespace wax {

constexpr int64_t KEY_PAD = INT64_MAX;          // "no candidate" (TopKReduction.metal:112 pads with (+inf, 0xFFFFFFFF))
constexpr uint64_t ID_PAD = UINT64_MAX;
constexpr int WAVE = 64;                         // CDNA wavefront
constexpr int SCAN_WAVES = 4;                    // waves per scan workgroup
constexpr int SCAN_THREADS = SCAN_WAVES * WAVE;
constexpr int MERGE_WAVES = 16;
constexpr int MERGE_THREADS = MERGE_WAVES * WAVE;
constexpr int FUSED_MAX_K = 192;                 // largest k served by the fused scan+select kernel
constexpr int MAX_GRID_BLOCKS = 8192;

// Monotone map f32 -> i32: signed integer order == float order (-0 < +0, NaN
// canonicalised by callers). An involution: applying it twice returns the bits.
__host__ __device__ inline int32_t order_bits(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t b = __float_as_int(d);
#else
    int32_t b;
    memcpy(&b, &d, 4);
#endif
    return b ^ ((b >> 31) & 0x7fffffff);
}

__host__ __device__ inline float unorder_bits(int32_t o) {
    int32_t b = o ^ ((o >> 31) & 0x7fffffff);
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(b);
#else
    float d;
    memcpy(&d, &b, 4);
    return d;
#endif
}

// key = ordered(distance) : row. Signed ascending order == (distance asc, row asc).
__host__ __device__ inline int64_t make_key(float dist, uint32_t row) {
    return (int64_t)(((uint64_t)(uint32_t)order_bits(dist) << 32) | (uint64_t)row);
}
__host__ __device__ inline float key_distance(int64_t key) { return unorder_bits((int32_t)(key >> 32)); }
__host__ __device__ inline uint32_t key_row(int64_t key) { return (uint32_t)((uint64_t)key & 0xffffffffull); }

}  // namespace wax
