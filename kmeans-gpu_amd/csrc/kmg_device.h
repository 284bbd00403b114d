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

// choose_centroid.wgsl:180-206 `pick` for all clusters by ONE workgroup of `block` threads (s_count: a word of LDS):
// centroid <- sum / count where count > 0, *n_converged = number of clusters that moved less than `convergence`
// (literal CIE94 of the new against the previous centroid); an empty cluster keeps its centroid and counts as not converged.
__device__ __forceinline__ void update_centroids(const int64_t *acc, uint32_t k, float convergence, Centroid *cent,
                                                 uint32_t *n_converged, uint32_t *s_count, uint32_t block)
{
    if (threadIdx.x == 0) *s_count = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t c = threadIdx.x; c < k; c += block) {
        const long long count = acc[4ull * c + 3];
        if (count > 0) {                                         // :185
            float nw[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double mean = ((double)acc[4ull * c + j] / (double)count) * (1.0 / 1048576.0);
                nw[j] = (float)mean;                             // :186
            }
            const Centroid prev = cent[c];
            Centroid nc;
            nc.L = nw[0]; nc.a = nw[1]; nc.b = nw[2]; nc.C = chroma(nw[1], nw[2]);
            cent[c] = nc;
            // :191 distance_cie94(new, previous) < settings.convergence
            if (cie94(nw[0], nw[1], nw[2], prev.L, prev.a, prev.b) < convergence) mine += 1;
        }                                                        // :192-194 empty: unchanged, 0
    }
    if (mine) atomicAdd(s_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) *n_converged = *s_count;               // :196-202
}

// ---- Lab -> sRGB8 on the device (meld output pass: lab_to_rgb.wgsl per pixel) ------------------------------------------------
// rgba8unorm store of a float in [0, 1]
__device__ __forceinline__ uint32_t unorm8(float v)
{
    v = v > 0.0f ? v : 0.0f;
    v = v > 1.0f ? 1.0f : v;
    return (uint32_t)rintf(v * 255.0f);
}

__device__ __forceinline__ float srgb_encode_dev(float c)
{
    // lab_to_rgb.wgsl:21-35; pow(c, 1/2.4): the shared-source routine of kmg_math.h (same bytes as host and oracle)
    return c > 0.0031308f ? 1.055f * pow_inv_2p4(c) - 0.055f : 12.92f * c;
}

__device__ __forceinline__ float lab_finv_dev(float t)
{
    const float t3 = t * t * t;                                  // lab_to_rgb.wgsl:45-59
    return t3 > 0.008856f ? t3 : KMG_DIV(t - 16.0f / 116.0f, 7.787f);
}

// The byte of a linear channel value c: unorm8(srgb_encode_dev(c)) is a monotone step function of the FLOAT c, so it equals the
// number of thresholds T[1..255] (T[b] = the smallest float whose byte is >= b) that do not exceed c -- eight steps of a
// binary search in a 1 KiB table instead of ~120 binary64 operations of pow_inv_2p4.  The table is made on the device, by
// that very function (k_encode_thresholds), so the bytes are the same by construction.
__device__ __forceinline__ uint32_t encode_byte(const float *s_thr, float c)
{
    uint32_t b = 0;
#pragma unroll
    for (uint32_t step = 128u; step > 0u; step >>= 1)
        b += c >= s_thr[b + step] ? step : 0u;
    return b;
}

template <bool TABLE>
__device__ __forceinline__ uint32_t lab_to_rgba8_dev(float L, float a, float b, const float *s_thr = nullptr)
{
    float y = KMG_DIV(L + 16.0f, 116.0f);
    float x = KMG_DIV(a, 500.0f) + y;
    float z = y - KMG_DIV(b, 200.0f);
    const float X = lab_finv_dev(x) * 95.0489f, Y = lab_finv_dev(y) * 100.0f, Z = lab_finv_dev(z) * 108.8840f;
    x = KMG_DIV(X, 100.0f); y = KMG_DIV(Y, 100.0f); z = KMG_DIV(Z, 100.0f);
    const float r = fmaf(-0.4985314f, z, fmaf(-1.5371385f, y, 3.2404542f * x));
    const float g = fmaf(0.0415560f, z, fmaf(1.8760108f, y, -0.9692660f * x));
    const float bl = fmaf(1.0572252f, z, fmaf(-0.2040259f, y, 0.0556434f * x));
    if (TABLE) return encode_byte(s_thr, r) | (encode_byte(s_thr, g) << 8) | (encode_byte(s_thr, bl) << 16) | 0xFF000000u;
    return unorm8(srgb_encode_dev(r)) | (unorm8(srgb_encode_dev(g)) << 8) | (unorm8(srgb_encode_dev(bl)) << 16) |
           0xFF000000u;
}

}  // namespace kmg
