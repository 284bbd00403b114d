// kmg_color.h -- host-side, O(k) colour conversions of libkmeans_hip: the work the reference
// does on the CPU with the `palette` crate 0.7.3 (core/src/structures.rs:523-553,581-617,
// core/src/lib.rs:276-284) plus the per-centroid (not per-pixel) halves of the output pass
// (lab_to_rgb.wgsl for the k palette entries, the dither threshold of mix_colors.wgsl:53-67).
// Compile with -ffp-contract=off.
#pragma once

#include <stdint.h>
#include <math.h>
#include <algorithm>
#include <vector>

#include "kmg_math.h"

namespace kmg {

// rgb_to_lab.wgsl:16-33 for the 256 possible byte values, times 100 (:31-33).
inline void build_srgb_lut100(float lut[256])
{
    for (int v = 0; v < 256; ++v) {
        float c = (float)v / 255.0f;
        float lin = c > 0.04045f ? (float)pow((double)((c + 0.055f) / 1.055f), 2.4) : c / 12.92f;
        lut[v] = lin * 100.0f;
    }
}

inline uint8_t to_unorm8(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)rintf(v * 255.0f);
}

// lab_to_rgb.wgsl:11-81 for one colour.
inline void shader_lab_to_rgba8(const float lab[3], uint8_t out[4])
{
    auto finv = [](float t) {
        float t3 = t * t * t;
        return t3 > 0.008856f ? t3 : (t - 16.0f / 116.0f) / 7.787f;
    };
    auto encode = [](float c) {
        return c > 0.0031308f ? 1.055f * pow_inv_2p4(c) - 0.055f : 12.92f * c;   // kmg_math.h: shared with the device
    };
    float y = (lab[0] + 16.0f) / 116.0f;
    float x = lab[1] / 500.0f + y;
    float z = y - lab[2] / 200.0f;
    float X = finv(x) * 95.0489f, Y = finv(y) * 100.0f, Z = finv(z) * 108.8840f;
    x = X / 100.0f; y = Y / 100.0f; z = Z / 100.0f;
    float r = fmaf(-0.4985314f, z, fmaf(-1.5371385f, y, 3.2404542f * x));
    float g = fmaf(0.0415560f, z, fmaf(1.8760108f, y, -0.9692660f * x));
    float b = fmaf(1.0572252f, z, fmaf(-0.2040259f, y, 0.0556434f * x));
    out[0] = to_unorm8(encode(r));
    out[1] = to_unorm8(encode(g));
    out[2] = to_unorm8(encode(b));
    out[3] = 255;
}

// palette 0.7.3: Srgb<u8> -> Srgb<f32> -> LinSrgb -> Xyz(D65) -> Lab  (structures.rs:533-536)
inline void crate_srgb8_to_lab(const uint8_t rgb[3], float lab[3])
{
    float lin[3];
    for (int i = 0; i < 3; ++i) {
        float c = (float)rgb[i] / 255.0f;
        lin[i] = c <= 0.04045f ? c / 12.92f : (float)pow((double)((c + 0.055f) / 1.055f), 2.4);
    }
    float X = 0.4124564f * lin[0] + 0.3575761f * lin[1] + 0.1804375f * lin[2];
    float Y = 0.2126729f * lin[0] + 0.7151522f * lin[1] + 0.0721750f * lin[2];
    float Z = 0.0193339f * lin[0] + 0.1191920f * lin[1] + 0.9503041f * lin[2];
    const float eps = 216.0f / 24389.0f, kappa = 24389.0f / 27.0f;
    float t[3] = {X / 0.95047f, Y / 1.0f, Z / 1.08883f}, f[3];
    for (int i = 0; i < 3; ++i)
        f[i] = t[i] > eps ? (float)cbrt((double)t[i]) : (kappa * t[i] + 16.0f) / 116.0f;
    lab[0] = 116.0f * f[1] - 16.0f;
    lab[1] = 500.0f * (f[0] - f[1]);
    lab[2] = 200.0f * (f[1] - f[2]);
}

// palette 0.7.3: Lab -> Xyz(D65) -> LinSrgb -> Srgb<f32> -> Srgb<u8>  (structures.rs:601-607)
inline void crate_lab_to_srgb8(const float lab[3], uint8_t rgb[3])
{
    const float eps = 6.0f / 29.0f, kappa = 108.0f / 841.0f, delta = 4.0f / 29.0f;
    float y = (lab[0] + 16.0f) / 116.0f;
    float x = y + lab[1] / 500.0f;
    float z = y - lab[2] / 200.0f;
    float t[3] = {x, y, z}, v[3];
    for (int i = 0; i < 3; ++i) v[i] = t[i] > eps ? t[i] * t[i] * t[i] : (t[i] - delta) * kappa;
    float X = v[0] * 0.95047f, Y = v[1], Z = v[2] * 1.08883f;
    float lin[3];
    lin[0] = 3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z;
    lin[1] = -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z;
    lin[2] = 0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z;
    for (int i = 0; i < 3; ++i) {
        float l = lin[i];
        float s = l <= 0.0031308f ? 12.92f * l : 1.055f * (float)pow((double)l, 1.0 / 2.4) - 0.055f;
        rgb[i] = to_unorm8(s);
    }
}

// mix_colors.wgsl:53-67 (needs k >= 2): distance of an approximate farthest pair / sqrt(k).
inline float dither_threshold(const float *c4, uint32_t k)
{
    const float *A = c4, *B = c4 + 4;
    float dAB = cie94(A[0], A[1], A[2], B[0], B[1], B[2]);
    for (uint32_t i = 2; i < k; ++i) {
        const float *ci = c4 + 4 * i;
        float dA = cie94(ci[0], ci[1], ci[2], A[0], A[1], A[2]);
        float dB = cie94(ci[0], ci[1], ci[2], B[0], B[1], B[2]);
        if (dA > dB && dA > dAB) { dAB = dA; B = ci; }
        else if (dB > dAB)       { dAB = dB; A = ci; }
    }
    return dAB / sqrtf((float)k);
}

}  // namespace kmg
