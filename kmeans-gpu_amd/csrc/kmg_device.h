// kmg_device.h -- device-side helpers shared by the gfx950 kernel translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmg_kernels.h"
#include "kmg_math.h"

namespace kmg {

__device__ __forceinline__ void px_to_lab(const float *s_lut, uint32_t px, float &L, float &a, float &b)
{
    linear100_to_lab(s_lut[px & 255u], s_lut[(px >> 8) & 255u], s_lut[(px >> 16) & 255u], L, a, b);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// streaming (non-temporal) variants: the pixel and label streams are touched once per pass and
// should not evict the small gather tables from L2
__device__ __forceinline__ void load4_stream(const uint32_t *rgba, uint64_t i0, uint64_t n, bool aligned,
                                             uint32_t px[4])
{
    if (aligned && i0 + 4 <= n) {
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rgba + i0));
        px[0] = v.x; px[1] = v.y; px[2] = v.z; px[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) px[j] = (i0 + j < n) ? rgba[i0 + j] : 0u;
    }
}

__device__ __forceinline__ void store4_stream(uint32_t *out, uint64_t i0, uint64_t n, bool aligned,
                                              const uint32_t v[4])
{
    if (aligned && i0 + 4 <= n) {
        u32x4 q = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(out + i0));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) out[i0 + j] = v[j];
    }
}

// 4 consecutive pixels of this thread; `full16` = all four in range and the address 16-B aligned.
__device__ __forceinline__ void load4(const uint32_t *rgba, uint64_t i0, uint64_t n, bool aligned,
                                      uint32_t px[4])
{
    if (aligned && i0 + 4 <= n) {
        uint4 v = *reinterpret_cast<const uint4 *>(rgba + i0);
        px[0] = v.x; px[1] = v.y; px[2] = v.z; px[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) px[j] = (i0 + j < n) ? rgba[i0 + j] : 0u;
    }
}

__device__ __forceinline__ void store4(uint32_t *out, uint64_t i0, uint64_t n, bool aligned,
                                       const uint32_t v[4])
{
    if (aligned && i0 + 4 <= n) {
        *reinterpret_cast<uint4 *>(out + i0) = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) out[i0 + j] = v[j];
    }
}


// Centroid table -> LDS as (L, a, b, C) float4, padded to a multiple of 4 with entries no pixel can
// be closest to (key ~ 1e36, above every sentinel).
__device__ __forceinline__ void stage_centroids(float4 *s_cent, const Centroid *__restrict__ cent,
                                                uint32_t k, uint32_t kpad)
{
    for (uint32_t i = threadIdx.x; i < kpad; i += kBlock) {
        float4 v;
        if (i < k) {
            const Centroid c = cent[i];
            v = make_float4(c.L, c.a, c.b, c.C);
        } else {
            v = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);   // key ~ 1e36: never below any sentinel
        }
        s_cent[i] = v;
    }
}


}  // namespace kmg
