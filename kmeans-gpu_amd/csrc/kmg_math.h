// kmg_math.h -- scalar colour arithmetic shared by the gfx950 kernels and the (O(k)) host
// control code of libkmeans_hip.  Every function states the IEEE-754 binary32 operation order;
// translation units including this header MUST be compiled with -ffp-contract=off so that the
// only fused operations are the explicit fmaf()/fma() calls.
//
// References (relative to the reference repository):
//   core/shaders/converters/rgb_to_lab.wgsl, core/shaders/functions/delta_e.wgsl
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMG_HD __host__ __device__ __forceinline__
#else
#define KMG_HD inline
#endif

#define KMG_FIX_SCALE 1048576.0f   // 2^20, see include/kmeans_hip.h KMG_FIX_SHIFT

namespace kmg {

KMG_HD float bits_to_float(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
KMG_HD uint32_t float_to_bits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

// Correctly rounded binary32 cube root for x in [1e-3, 2] (exhaustively verified against
// libm on that range, tests/test_host_math.py; the Lab conversion only feeds it
// t in (0.008856, 1.01]).  rgb_to_lab.wgsl:45,50,55 `pow(t, 1.0/3.0)`.
//   1. r ~ x^(-1/3): host -- bit-level seed, three Newton steps r <- r*(4/3 - (x/3) r^3) in f32;
//      device -- v_log_f32 / v_exp_f32 (rel. error ~1e-6: the two binary64 steps below square it twice)
//   2. y0 = x r^2, g = r^2/3 ~ 1/(3 y^2)
//   3. two Newton steps y <- y + (x - y^3) g in f64, round once to f32
// Both seeds end in the correctly rounded result, hence in the same float: tools/lab_rate.hip compares the two
// on the device over every binary32 in [1e-3, 2] (0 differences), tests/test_gpu_parity.py over every colour.
KMG_HD float cbrt_cr(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * -0.33333334f);
#else
    uint32_t ir = 0x54A21D2Au - float_to_bits(x) / 3u;
    float r = bits_to_float(ir);
    float xt = x * 0.33333334f;
    for (int i = 0; i < 3; ++i) {
        float r3 = r * r * r;
        r = r * fmaf(-xt, r3, 1.3333334f);
    }
#endif
    float rr = r * r;
    double yd = (double)(x * rr);
    double gd = (double)(rr * 0.33333334f);
    double xd = (double)x;
    for (int i = 0; i < 2; ++i) {
        double res = fma(-yd * yd, yd, xd);
        yd = fma(res, gd, yd);
    }
    return (float)yd;
}

// x / c for one of the three white-point constants.  Device: q0 = x RN(1/c), one residual correction
// q = q0 + (x - q0 c) RN(1/c) -- 3 operations instead of the ~10 of an IEEE division (~47 issue cycles on
// gfx950), and the same float: tools/lab_rate.hip compares it with x / c over every binary32 in [2^-20, 128)
// for each constant (0 differences; X, Y, Z are 0 or in [5e-4, 109]).
template <int WHICH>
KMG_HD float div_white(float x)
{
    constexpr float c = WHICH == 0 ? 95.0489f : (WHICH == 1 ? 100.0f : 108.8840f);
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr float rc = 1.0f / c;
    const float q0 = x * rc;
    return fmaf(fmaf(-q0, c, x), rc, q0);
#else
    return x / c;
#endif
}

// x / c for the constants of lab_to_rgb.wgsl (116, 500, 200, 100, 7.787) on the device: the same three operations.  For which
// x they give the IEEE quotient is not argued but counted: kmg_debug_division_check (tests/test_gpu_parity.py) tries every
// binary32 x for every constant; outside [kDivLo, kDivHi] in magnitude (where the residual or the quotient leaves the normal
// range) the division itself is used.
constexpr float kDivLo = 1.0e-30f, kDivHi = 1.0e30f;
KMG_HD float div_const(float x, float c, float rc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float ax = __builtin_fabsf(x);
    if (ax >= kDivLo && ax <= kDivHi) {
        const float q0 = x * rc;
        return fmaf(fmaf(-q0, c, x), rc, q0);
    }
#else
    (void)rc;
#endif
    return x / c;
}
#define KMG_DIV(x, c) ::kmg::div_const((x), (c), 1.0f / (c))

// lab_to_rgb.wgsl:21-35 `pow(c, 1.0 / 2.4)` for c in (0.0031308, 1): ONE definition shared by the device kernels
// (meld output pass), the host palette code and -- restated operation by operation -- the oracle, so that their
// bytes agree (two different libm / ocml `pow` implementations disagree on ~1e-5 of the channels).
// The shader's exponent is the binary32 value y = f32(1 / 2.4) = 5/12 - d, d = 9.93e-9:
//   c^y = c^(5/12) * exp(-d ln c),   c^(1/12) = sqrt(sqrt(cbrt(c))),   ln c = 24 atanh((s - 1) / (s + 1)), s = c^(1/12)
// in binary64 with IEEE +, -, *, /, sqrt only (compile with -ffp-contract=off): relative error ~1e-15 before the
// single rounding to binary32.  c >= 1 gives 1 (the caller clamps to 255 anyway).
KMG_HD float pow_inv_2p4(float c)
{
    if (!(c < 1.0f)) return 1.0f;
    const double x = (double)c;
    double y = (double)cbrt_cr(c < 1.0e-3f ? 1.0e-3f : c);        // binary32 cube root as the seed (c > 0.0031308 here)
    for (int i = 0; i < 2; ++i) y = y - (((y * y) * y) - x) / ((3.0 * y) * y);
    const double s = sqrt(sqrt(y));                               // c^(1/12)
    const double s2 = s * s, s4 = s2 * s2, p = s4 * s;            // c^(5/12)
    const double z = (s - 1.0) / (s + 1.0), z2 = z * z;
    const double series = 1.0 + z2 * (1.0 / 3.0 + z2 * (1.0 / 5.0 + z2 * (1.0 / 7.0 + z2 * (1.0 / 9.0 + z2 * (1.0 / 11.0 + z2 * (1.0 / 13.0 + z2 * (1.0 / 15.0)))))));
    const double ln_c = 24.0 * (z * series);
    const double d = 5.0 / 12.0 - (double)(1.0f / 2.4f);
    const double t = -(d * ln_c);
    return (float)(p * (1.0 + t + (0.5 * t) * t));
}

// rgb_to_lab.wgsl:44-58
KMG_HD float lab_f(float t)
{
    return t > 0.008856f ? cbrt_cr(t) : fmaf(7.787f, t, 16.0f / 116.0f);
}

// rgb_to_lab.wgsl:11-64.  r,g,b are the (sRGB-decoded * 100) table values of the three bytes.
KMG_HD void linear100_to_lab(float r, float g, float b, float &L, float &A, float &B)
{
    float X = fmaf(0.1804375f, b, fmaf(0.3575761f, g, 0.4124564f * r));
    float Y = fmaf(0.0721750f, b, fmaf(0.7151522f, g, 0.2126729f * r));
    float Z = fmaf(0.9503041f, b, fmaf(0.1191920f, g, 0.0193339f * r));
    float fx = lab_f(div_white<0>(X));
    float fy = lab_f(div_white<1>(Y));
    float fz = lab_f(div_white<2>(Z));
    L = fmaf(116.0f, fy, -16.0f);
    A = 500.0f * (fx - fy);
    B = 200.0f * (fy - fz);
}

// delta_e.wgsl:8-9
KMG_HD float chroma(float a, float b) { return sqrtf(a * a + b * b); }

// delta_e.wgsl:1-22, literal form (no fused operations).  Asymmetric in its arguments.
KMG_HD float cie94(float L1, float a1, float b1, float L2, float a2, float b2)
{
    float dL = L1 - L2, da = a1 - a2, db = b1 - b2;
    float C1 = chroma(a1, b1);
    float C2 = chroma(a2, b2);
    float dC = C1 - C2;
    float dH = sqrtf(fmaxf((da * da) + (db * db) - (dC * dC), 0.0f));
    float SC = 1.0f + 0.045f * C1;
    float SH = 1.0f + 0.015f * C1;
    float tL = dL / 1.0f, tC = dC / SC, tH = dH / SH;
    return sqrtf(tL * tL + tC * tC + tH * tH);
}

// Per-pixel terms of the arg-min key (hoisted out of the centroid loop).
struct PixelTerms { float L, a, b, C, wC, wH; };

// (C = chroma(a, b) supplied by the caller, e.g. from the per-colour Lab table)
KMG_HD PixelTerms pixel_terms_c(float L, float a, float b, float C)
{
    PixelTerms p;
    p.L = L; p.a = a; p.b = b;
    p.C = C;
    float SC = 1.0f + 0.045f * p.C;
    float SH = 1.0f + 0.015f * p.C;
    float iSC = 1.0f / SC, iSH = 1.0f / SH;
    p.wC = iSC * iSC;
    p.wH = iSH * iSH;
    return p;
}

KMG_HD PixelTerms pixel_terms(float L, float a, float b) { return pixel_terms_c(L, a, b, chroma(a, b)); }

#if defined(__HIPCC__)
// The same terms with hardware reciprocals (1 ulp) instead of two IEEE divides (~47 issue cycles each on
// gfx950): the weights move by a few u, far inside the tie slack -- for kernels that ORDER by the key and
// settle near-ties with the literal distance ("near-tie repair" below).  Never for values that are used.
__device__ __forceinline__ PixelTerms pixel_terms_fast(float L, float a, float b, float C)
{
    PixelTerms p;
    p.L = L; p.a = a; p.b = b; p.C = C;
    const float iSC = __builtin_amdgcn_rcpf(fmaf(0.045f, C, 1.0f)), iSH = __builtin_amdgcn_rcpf(fmaf(0.015f, C, 1.0f));
    p.wC = iSC * iSC;
    p.wH = iSH * iSH;
    return p;
}
#endif

// Squared CIE94 used only for ordering: dL^2 + dC^2 wC + max(da^2 + db^2 - dC^2, 0) wH.
KMG_HD float cie94_key(const PixelTerms &p, float L2, float a2, float b2, float C2)
{
    float dL = p.L - L2, da = p.a - a2, db = p.b - b2, dC = p.C - C2;
    float dC2 = dC * dC;
    float t = fmaf(db, db, da * da);
    float h = fmaxf(t - dC2, 0.0f);
    return fmaf(h, p.wH, fmaf(dC2, p.wC, dL * dL));
}

// delta_e.wgsl:1-22 with the two chroma values supplied (C1 = chroma(a1, b1), C2 = chroma(a2, b2)): the
// same operations as cie94() above, hence the same float.
KMG_HD float cie94_c(float L1, float a1, float b1, float C1, float L2, float a2, float b2, float C2)
{
    float dL = L1 - L2, da = a1 - a2, db = b1 - b2;
    float dC = C1 - C2;
    float dH = sqrtf(fmaxf((da * da) + (db * db) - (dC * dC), 0.0f));
    float SC = 1.0f + 0.045f * C1;
    float SH = 1.0f + 0.015f * C1;
    float tL = dL / 1.0f, tC = dC / SC, tH = dH / SH;
    return sqrtf(tL * tL + tC * tC + tH * tH);
}

// ---- near-tie repair: the arg-min is that of the LITERAL distance -------------------------------------
// find_centroid.wgsl:32-41 takes the first index that minimises the literal distance_cie94 under strict
// '<'.  The kernels order by cie94_key (no sqrt, no divide per pair); key and literal^2 are two float
// evaluations of the same real quantity T = dL^2 + dC^2/SC^2 + max(da^2 + db^2 - dC^2, 0)/SH^2 from the same
// float inputs (dL, da, db, dC, SC, SH).  With u = 2^-24 per rounding:
//   * every product / quotient / sum of non-negative terms carries at most 9 roundings in the literal form
//     and 6 in the key, i.e. relative errors <= 9u and <= 6u on each term;
//   * the hue term subtracts: h = (da^2 + db^2) - dC^2 is computed with absolute error <= 3u (G + Q)
//     (G = da^2 + db^2, Q = dC^2) in either form, and G <= Q + h(1 + small), so the error it adds to T is
//     <= 3u (h + 2Q) / SH^2 <= 3u (h/SH^2 + 18 Q/SC^2) <= 57u T     ((SC/SH)^2 < 9 for every chroma);
//   hence |key - T| <= 63u T and |literal^2 - T| <= 66u T, and a centroid j can have
//   literal_j <= literal_best only if key_j <= key_best (1 + 66u)(1 + 63u) / ((1 - 66u)(1 - 63u))
//   < key_best (1 + 260u) = key_best (1 + 1.6e-5).
// Kernels that also (a) take the weights from hardware reciprocals (pixel_terms_fast: +6u on two terms) and
// (b) keep a 5-bit candidate position in the low mantissa bits of the key so that arg-min and runner-up are one
// integer min / med3 each (k_cube_scan: the key loses up to 2^-18 = 64u) stay below key_best (1 + 560u) = 3.4e-5.
// The kernels use kTieSlack = 2^-13 = 1.2e-4 (8x that): while scanning they keep the second smallest key; a
// pixel whose second smallest key is <= best (1 + kTieSlack) is re-decided with the literal distance among
// the centroids whose key is within that threshold, in index order with strict '<' -- exactly the
// reference's result, because everything outside the threshold is strictly farther in the literal
// distance.  Candidate sets built from interval bounds keep lo_j <= U (1 + kMaskSlack), kMaskSlack = 2^-12.
#define KMG_TIE_SLACK 0.0001220703125f     // 2^-13
constexpr float kTieSlack = KMG_TIE_SLACK;
constexpr float kMaskSlack = 0.000244140625f;   // 2^-12

// false: the arg-min is that of the key alone (round-1 definition, kept as a switch for A/B timing)
constexpr bool kLiteralArgmin = true;

KMG_HD float tie_threshold(float best) { return fmaf(best, kTieSlack, best); }

// Lab -> fixed point for the exact integer accumulators.
KMG_HD int32_t lab_fix(float x) { return (int32_t)rintf(x * KMG_FIX_SCALE); }

}  // namespace kmg
