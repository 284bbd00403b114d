// kmg_table_dev.h -- device helpers shared by the colour-table translation units (kmg_table.hip,
// kmg_cube.hip).  Compile with -ffp-contract=off.
#pragma once

#include "kmg_table.h"
#include "kmg_device.h"

namespace kmg {

struct KeyRange { float lo, hi; };

__device__ __forceinline__ void abs_range(float x0, float x1, float c, float &m, float &M)
{
    const float d0 = x0 - c, d1 = x1 - c;                  // d0 <= d1 (rounding is monotone)
    M = fmaxf(fabsf(d0), fabsf(d1));
    m = fmaxf(fmaxf(d0, -d1), 0.0f);                       // d0 if the interval is above c, -d1 if below, 0 if it straddles c (one v_max3)
}

// [min, max] of cie94_key(pixel, c) over all pixels whose terms lie inside the bounds: every operation of
// cie94_key is monotone in each operand (IEEE rounding is monotone), so the same operations on the interval
// end points bound the FLOAT key of every colour rigorously -- no epsilons
__device__ __forceinline__ KeyRange key_range(const CellBounds &cb, float L2, float a2, float b2, float C2)
{
    float mL, ML, ma, Ma, mb, Mb, mC, MC;
    abs_range(cb.L0, cb.L1, L2, mL, ML);
    abs_range(cb.a0, cb.a1, a2, ma, Ma);
    abs_range(cb.b0, cb.b1, b2, mb, Mb);
    abs_range(cb.C0, cb.C1, C2, mC, MC);
    const float A0 = mL * mL, A1 = ML * ML;                // dL*dL
    const float D0 = mC * mC, D1 = MC * MC;                // dC2
    const float t0 = fmaf(mb, mb, ma * ma), t1 = fmaf(Mb, Mb, Ma * Ma);
    const float h0 = fmaxf(t0 - D1, 0.0f), h1 = fmaxf(t1 - D0, 0.0f);
    KeyRange r;
    r.lo = fmaf(h0, cb.wH0, fmaf(D0, cb.wC0, A0));
    r.hi = fmaf(h1, cb.wH1, fmaf(D1, cb.wC1, A1));
    return r;
}

// The arg-min of the reference is over the LITERAL distance (find_centroid.wgsl:32-41); the kernels order by
// the cheaper key and repair near-ties with the literal form (kmg_math.h, "near-tie repair").  A centroid
// can win or tie under the literal distance only if its key is within kTieSlack of the smallest key, so the
// candidate sets keep everything whose lower bound is not above U * (1 + kMaskSlack), kMaskSlack > kTieSlack.
__device__ __forceinline__ float mask_threshold(float U) { return fmaf(U, kMaskSlack, U); }

__device__ __forceinline__ void colour_to_lab(const float *s_lut, uint32_t idx, float &L, float &a, float &b)
{
    uint32_t r, g, bl;
    index_to_rgb(idx, r, g, bl);
    linear100_to_lab(s_lut[r], s_lut[g], s_lut[bl], L, a, b);
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ float lane_value(float v, uint32_t src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)src));
}

__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}

__device__ __forceinline__ long long lane_value64(long long v, uint32_t src)
{
    const uint32_t lo = lane_value((uint32_t)v, src), hi = lane_value((uint32_t)((unsigned long long)v >> 32), src);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ long long wave_sum(long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// DPP cross-lane moves (no LDS traffic): src of lane i = lane perm(i) within its row of 16.  For the permutations used
// through these two helpers (quad_perm, row_mirror, row_half_mirror) every lane has a source, so `old` is never taken: old = 0
// with bound_ctrl lets the compiler fold the move into the operation that consumes it (v_min_f32_dpp, v_add_u32_dpp ...:
// one instruction per reduction step instead of copy + v_mov_b32_dpp + operation).  Call them with every lane of the wave
// active: a lane switched off reads as 0, which is the identity of sum, or and unsigned max but not of a minimum.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror: i <-> 7 - i inside each group of 8
constexpr int kDppMirror = 0x140;       // row_mirror: i <-> 15 - i inside each row of 16

// min over the 8 lanes that share lane >> 3 (all of them receive it)
__device__ __forceinline__ float group8_min(float v)
{
    v = fminf(v, dpp_f32<kDppXor1>(v));
    v = fminf(v, dpp_f32<kDppXor2>(v));
    v = fminf(v, dpp_f32<kDppHalfMirror>(v));
    return v;
}

// min over the wave (uniform result)
__device__ __forceinline__ float wave_min(float v)
{
    v = group8_min(v);
    v = fminf(v, dpp_f32<kDppMirror>(v));                        // every lane: the minimum of its row of 16
    const float big = 3.0e38f;
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, big), __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, big), __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)));
    return lane_value(v, 63);
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// sum over the wave of a per-lane u32 whose wave total cannot overflow (e.g. three packed 10-bit counters of
// values <= 8); uniform result
__device__ __forceinline__ uint32_t wave_add_u32(uint32_t v)
{
    v += dpp_u32<kDppXor1>(v);
    v += dpp_u32<kDppXor2>(v);
    v += dpp_u32<kDppHalfMirror>(v);
    v += dpp_u32<kDppMirror>(v);                                 // every lane: the sum of its row of 16
    // rows 1 and 3 take in rows 0 and 2 (row_bcast:15), then rows 2 and 3 the sum of rows 0 + 1 (row_bcast:31):
    // lane 63 holds the total -- one v_readlane (8 issue cycles on gfx950) instead of four
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return lane_value(v, 63);
}

// max over the wave of a per-lane u32 (uniform result): six DPP steps and one v_readlane, no LDS permutes
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v)
{
    v = max(v, dpp_u32<kDppXor1>(v));
    v = max(v, dpp_u32<kDppXor2>(v));
    v = max(v, dpp_u32<kDppHalfMirror>(v));
    v = max(v, dpp_u32<kDppMirror>(v));                          // every lane: the maximum of its row of 16
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return lane_value(v, 63);
}

__device__ __forceinline__ float bayer16(uint32_t i)            // mix_colors.wgsl:13-16
{
    constexpr uint64_t M = 0x5D7F91B36E4CA280ull;               // 0 8 2 10 12 4 14 6 3 11 1 9 15 7 13 5, 4 bits each
    return (float)((M >> (4u * i)) & 15ull);
}

// median of three unsigned values (one v_med3_u32; clang has a builtin for the float form only)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// number of set bits of m below this lane's bit
// A zero the compiler cannot see through, in a VGPR: added to a wave-uniform index it turns the load into a VECTOR
// memory load.  Scalar loads share their wait counter with LDS and return out of order, so a prefetch issued as a
// scalar load is waited for -- in full -- at the next LDS access; a vector load is not.
__device__ __forceinline__ uint32_t opaque_vgpr_zero()
{
    uint32_t z = 0u;
    asm volatile("" : "+v"(z));
    return z;
}

__device__ __forceinline__ uint32_t bits_below_lane(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

}  // namespace kmg
