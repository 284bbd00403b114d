// kmg_cube.hip -- the per-iteration pass of the colour-table strategy over the colour cube (kmg_table.h).  Every phase of it is a
// chain of dependent steps per wave, so what it needs is many resident waves and no waiting for each other.
//
//   k_cube_small  k <= 32: the whole pass in ONE launch, a workgroup owns 64 cells from bounds to pair entries.
//   k_cube_one    32 < k <= 256, images without hot cells (round 6): the same for centroid tables that need candidate LISTS --
//                 1a. cell candidates (one wave per 8x8x8 cell) -- lanes strided over the centroids: interval bounds
//                     [lo_j, hi_j] of the key over the cell (static bounds, kmg_table_dev.h key_range), U = min_j hi_j,
//                     keep lo_j <= U (1 + slack).  One candidate -> the whole cell belongs to it: sums from the
//                     per-cell table of the image, one pair entry, NO per-colour traffic at all.
//                 1b. sub-cell stage -- a thread per 4x4x4 sub-cell bounds the (<= 32) listed candidates over the sub-cell's own
//                     static bounds, then the DOMINANCE test: a candidate that another one beats on EVERY colour of the sub-cell
//                     (an affine model of the key difference with exact residual ranges) leaves its set.  One candidate -> the
//                     sub-cell belongs to it: sums from the per-sub-cell table, 64 labels.
//                 2.  scan of the undecided sub-cells, two per step, ONE colour of each per lane, handed out by an LDS counter;
//                 3.  the pair entry of every cell with more than one label, by the wave that scans the cell's last item.
//   k_cube_stage  everything else (k > 256; images with hot cells -- a photograph's few hundred cells with long candidate
//   k_cube_scan   lists are spread over the device, four waves each): steps 1a + 1b without the dominance test, the scan from
//   k_cube_pairs  the cells' work records, the pair entries (k <= 256) / cell summaries (k > 256) -- three launches.
// Exactness: bounds are float-monotone interval evaluations of the very operations of cie94_key, so the
// arg-min of every colour (and everything within the near-tie threshold of it) survives both prunings; the dominance
// test charges its own rounding explicitly (`dominated`); sums are integers.  tests/test_gpu_table.py compares every
// label / sum with the per-pixel scan and the oracle, and all 2^24 colours with the brute-force arg-min.
// Compile with -ffp-contract=off.

#include "kmg_internal.h"
#include "kmg_table_dev.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>

namespace kmg {

typedef float f32x2 __attribute__((ext_vector_type(2)));


namespace {

constexpr uint32_t kMaxListed = 32;       // candidates per cell the sub-cell stage handles (4 rounds of 8 lanes)
// Cells with more candidates (photographs: the centroids crowd into the few dark cells that hold most of the pixels -- 866
// cells of the test photograph at k = 256, up to 215 candidates) are bounded per sub-cell all the same, from a list of up to
// kMaxLong candidates; their per-sub-cell candidate sets leave as k-bit masks behind the cell masks (kmg_table.h
// cube_masks_bytes), and the scan kernel visits a sub-cell's own set instead of every candidate of the cell.  k <= 256.
constexpr uint32_t kMaxLong = 256;
constexpr uint32_t kLongFlag = 0x200u;    // CellWork::scan_set: the cell's sub-cells have masks of their own

__device__ __forceinline__ uint32_t sel3(uint32_t i, uint32_t x0, uint32_t x1, uint32_t x2)
{
    return i == 0u ? x0 : (i == 1u ? x1 : x2);
}

__device__ __forceinline__ int round_dir(int g, int m)           // rint(2 g / m), |g| <= m, m > 0
{
    const int a = g < 0 ? -g : g;
    const int r = (4 * a >= 3 * m ? 1 : 0) + (4 * a >= m ? 1 : 0);
    return g < 0 ? -r : r;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t group8_or(uint32_t v)
{
    v |= dpp_u32<kDppXor1>(v);
    v |= dpp_u32<kDppXor2>(v);
    v |= dpp_u32<kDppHalfMirror>(v);
    return v;
}

__device__ __forceinline__ uint32_t group8_min_u32(uint32_t v)
{
    v = min(v, dpp_u32<kDppXor1>(v));
    v = min(v, dpp_u32<kDppXor2>(v));
    v = min(v, dpp_u32<kDppHalfMirror>(v));
    return v;
}

__device__ __forceinline__ uint32_t group8_max_u32(uint32_t v)
{
    v = max(v, dpp_u32<kDppXor1>(v));
    v = max(v, dpp_u32<kDppXor2>(v));
    v = max(v, dpp_u32<kDppHalfMirror>(v));
    return v;
}



// ------------------------------------------------------------------------------------------
// pair entry of one cell (kmg_table.h).  Lane l holds the colours 8 l .. 8 l + 7 of the cell
// (l = [r2 g2 b2 r1 r0 g1], q = [g0 b1 b0]); idx[q] their labels, bit q of occ = the colour is occupied.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cell_pair_entry(const uint32_t idx[8], uint32_t occ, uint32_t lane)
{
    // the (up to) three first distinct labels among the occupied colours, their colours and counts (per-lane
    // counts are <= 8, wave totals <= 512: three 10-bit counters share one DPP reduction)
    uint32_t rem = occ;
    uint32_t lab0 = 0, lab1 = 0, lab2 = 0, cnt0 = 0, cnt1 = 0, cnt2 = 0, msk0 = 0, msk1 = 0, msk2 = 0, n_occ = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const unsigned long long any = __ballot(rem != 0u);
        if (any) {
            uint32_t mine = 0;
#pragma unroll
            for (int q = 7; q >= 0; --q) mine = ((rem >> q) & 1u) ? idx[q] : mine;
            const uint32_t v = lane_value(mine, (uint32_t)__builtin_ctzll(any));
            uint32_t mm = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) mm |= (idx[q] == v ? 1u : 0u) << q;
            mm &= rem;
            rem &= ~mm;
            const uint32_t packed = wave_add_u32((uint32_t)__builtin_popcount(mm) | (t == 0 ? (uint32_t)__builtin_popcount(occ) << 10 : 0u));
            const uint32_t c = packed & 1023u;
            if (t == 0) { lab0 = v; msk0 = mm; cnt0 = c; n_occ = packed >> 10; }
            if (t == 1) { lab1 = v; msk1 = mm; cnt1 = c; }
            if (t == 2) { lab2 = v; msk2 = mm; cnt2 = c; }
        }
    }
    // A = the most frequent of them, B = the runner-up
    uint32_t a = 0;
    if (cnt1 > cnt0) a = 1;
    if (cnt2 > sel3(a, cnt0, cnt1, cnt2)) a = 2;
    uint32_t b = a == 0u ? 1u : 0u;
    if (a != 1u && b != 1u && cnt1 > sel3(b, cnt0, cnt1, cnt2)) b = 1;
    if (a != 2u && cnt2 > sel3(b, cnt0, cnt1, cnt2)) b = 2;
    const uint32_t labA = sel3(a, lab0, lab1, lab2), labB = sel3(b, lab0, lab1, lab2);
    const uint32_t mA = sel3(a, msk0, msk1, msk2), mB = sel3(b, msk0, msk1, msk2);
    const uint32_t rest = occ & ~mA;                              // occupied colours with another label
    const uint32_t nA = sel3(a, cnt0, cnt1, cnt2), nR = n_occ - nA;
    if (nR == 0u) return pair_entry(labA, labA, 0u, 0u, 0u);

    // direction: from the centre of mass of A towards the one of the rest, at half-cell resolution
    // (lane bits 5, 4, 3 = r2, g2, b2), rounded to components in -2..2.  Any direction is valid
    // (tlo and w below are exact for it); a good one only makes the slab thin.
    const uint32_t half = ((lane >> 5) & 1u) | (((lane >> 4) & 1u) << 10) | (((lane >> 3) & 1u) << 20);   // r2, g2, b2 of this lane
    const uint32_t hR = wave_add_u32((uint32_t)__builtin_popcount(rest) * half);
    const uint32_t hA = wave_add_u32((uint32_t)__builtin_popcount(mA) * half);
    const int gx = (int)((hR & 1023u) * nA) - (int)((hA & 1023u) * nR);
    const int gy = (int)(((hR >> 10) & 1023u) * nA) - (int)(((hA >> 10) & 1023u) * nR);
    const int gz = (int)((hR >> 20) * nA) - (int)((hA >> 20) * nR);
    const int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy, az = gz < 0 ? -gz : gz;
    const int m = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
    int nx = 2, ny = 0, nz = 0;
    if (m > 0) { nx = round_dir(gx, m); ny = round_dir(gy, m); nz = round_dir(gz, m); }

    // p of this lane's colours: x = r & 7 and the high bits of y, z are lane constants
    const int xl = (int)(((lane >> 5) & 1u) * 4u + ((lane >> 1) & 3u));
    const int yl = (int)(((lane >> 4) & 1u) * 4u + (lane & 1u) * 2u);
    const int zl = (int)(((lane >> 3) & 1u) * 4u);
    const int bias = 7 * ((nx < 0 ? -nx : 0) + (ny < 0 ? -ny : 0) + (nz < 0 ? -nz : 0));
    const int p0 = nx * xl + ny * yl + nz * zl + bias;
    int p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = p0 + ny * (q >> 2) + nz * (q & 3);
    // tlo = lowest p of a colour that is not A, thi = highest p of a colour that is not B
    const uint32_t notB = occ & ~mB;
    uint32_t tl = 63u, hi = 64u;                                  // hi = 63 - thi
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if ((rest >> q) & 1u) tl = min(tl, (uint32_t)p[q]);
        if ((notB >> q) & 1u) hi = min(hi, (uint32_t)(63 - p[q]));
    }
    u16x2 packed = __builtin_bit_cast(u16x2, tl | (hi << 16));
    packed = __builtin_elementwise_min(packed, __builtin_bit_cast(u16x2, dpp_u32<kDppXor1>(__builtin_bit_cast(uint32_t, packed))));
    packed = __builtin_elementwise_min(packed, __builtin_bit_cast(u16x2, dpp_u32<kDppXor2>(__builtin_bit_cast(uint32_t, packed))));
    packed = __builtin_elementwise_min(packed, __builtin_bit_cast(u16x2, dpp_u32<kDppHalfMirror>(__builtin_bit_cast(uint32_t, packed))));
    packed = __builtin_elementwise_min(packed, __builtin_bit_cast(u16x2, dpp_u32<kDppMirror>(__builtin_bit_cast(uint32_t, packed))));
    packed = __builtin_elementwise_min(packed, __builtin_bit_cast(u16x2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)__builtin_bit_cast(uint32_t, packed), 0x142, 0xA, 0xF, false)));
    packed = __builtin_elementwise_min(packed, __builtin_bit_cast(u16x2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)__builtin_bit_cast(uint32_t, packed), 0x143, 0xC, 0xF, false)));
    const uint32_t pk = lane_value(__builtin_bit_cast(uint32_t, packed), 63);
    const int tlo = (int)(pk & 0xFFFFu);
    const int thi = 63 - (int)(pk >> 16);
    const int w = thi + 1 > tlo ? thi + 1 - tlo : 0;
    const uint32_t code = pair_dir_code(nx, ny, nz);
    if (w <= 6) return pair_entry(labA, labB, code, (uint32_t)tlo, (uint32_t)w);
    // the slab is too wide to encode (a third label, or a strongly curved boundary): keep the
    // side that resolves more colours, the other one goes through the per-colour table
    uint32_t low = 0, high = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if ((occ >> q) & 1u) { low += p[q] < tlo ? 1u : 0u; high += p[q] > thi ? 1u : 0u; }
    }
    const uint32_t lh = wave_add_u32(low | (high << 10));
    if ((lh & 1023u) >= (lh >> 10)) return pair_entry(labA, labB, code, (uint32_t)tlo, 7u);
    const int range = 7 * ((nx < 0 ? -nx : nx) + (ny < 0 ? -ny : ny) + (nz < 0 ? -nz : nz));
    return pair_entry(labB, labA, pair_dir_code(-nx, -ny, -nz), (uint32_t)(range - thi), 7u);
}

__device__ __forceinline__ uint32_t merge_state(uint32_t a, uint32_t b)       // label / kSubEmpty / kSubMixed
{
    return (a == kSubEmpty) ? b : ((b == kSubEmpty || b == a) ? a : (uint32_t)kSubMixed);
}

// 64 consecutive labels of a sub-cell, lane l holding label l: four (u8) / two (u16) neighbouring lanes
// combine theirs into one dword through DPP, so a sub-cell is 16 (32) dword stores instead of 64 byte stores
template <typename LabelT>
__device__ __forceinline__ void store_labels64(LabelT *dst64, uint32_t label, uint32_t lane)
{
    if (sizeof(LabelT) == 1) {
        const uint32_t t = label | (dpp_u32<kDppXor1>(label) << 8);          // even lane: own | next << 8
        const uint32_t u = t | (dpp_u32<kDppXor2>(t) << 16);                  // lane % 4 == 0: four labels in order
        if ((lane & 3u) == 0u) reinterpret_cast<uint32_t *>(dst64)[lane >> 2] = u;
    } else {
        const uint32_t t = label | (dpp_u32<kDppXor1>(label) << 16);
        if ((lane & 1u) == 0u) reinterpret_cast<uint32_t *>(dst64)[lane >> 1] = t;
    }
}


// What k_cube_stage leaves for k_cube_scan, one record per cell (128 B)
struct alignas(16) CellWork {
    uint16_t list[kMaxListed];     // the cell's candidates in index order (listed cells)
    unsigned long long br[4];      // round r: bit 8 s + c = sub-cell s keeps candidate 8 r + c of the list
    uint32_t npop;                 // number of candidates of the cell
    uint32_t scan_set;             // bits 0..7: sub-cells whose colours are scanned; bit 8: the cell is listed; bit 9: long list;
                                   // bits 16..23: sub-cells the stage decided as a whole (one candidate)
    uint32_t pad[6];
};
static_assert(sizeof(CellWork) == 128, "CellWork layout");

constexpr uint32_t kPairPending = 0xFFFFFFFEu;   // pair entry of a cell k_cube_pairs still has to derive

// Behind the cells' work records (u32 words), for images with hot cells: the cells with LONG candidate lists -- a photograph's few hundred
// heavy cells, each minutes of a wave's time compared with the others -- are listed by the stage kernel (kLongSegs counters, then the cells,
// kLongSegCap per segment) and scanned first, a PAIR of sub-cells per wave instead of the whole cell.
constexpr uint32_t kLongSegs = 16, kLongSegCap = kCells / kLongSegs, kLongCount = 0, kLongCells = 64;
constexpr size_t kListsWords = kLongCells + kCells;
constexpr uint32_t kItemCands = 12;                               // k_cube_one: candidates an item carries as bytes

// LDS bins: repl copies of k x 4 u64, consecutive copies 32 B further along the bank row
__device__ __forceinline__ void flush_bins(const unsigned long long *bins, uint32_t k, uint32_t repl, uint32_t bin_stride,
                                           int64_t *sums, uint32_t n_rows)
{
    __syncthreads();
    // only the clusters this workgroup met are non-zero
    unsigned long long *row = reinterpret_cast<unsigned long long *>(sums) + (uint64_t)(blockIdx.x % n_rows) * 4ull * k;
    for (uint32_t i = threadIdx.x; i < 4u * k; i += blockDim.x) {
        unsigned long long v = 0ull;
        for (uint32_t r = 0; r < repl; ++r) v += bins[(uint64_t)r * bin_stride + i];
        if (v) atomicAdd(row + i, v);
    }
}

// The sub-cell stage of a cell with more than kMaxListed candidates (k_cube_stage, "long list"): the candidates' mask words
// wait in s_masks; on return s_sub[4 s + w] = word w of the candidates sub-cell s keeps, and lane s < 8 gets
// (number of candidates sub-cell s keeps, the candidate itself if it is one).
__device__ __forceinline__ uint2 long_list_stage(const float4 *s_cent, const unsigned long long *s_masks, uint16_t *s_long,
                                              unsigned long long *s_sub, float4 sb0, float4 sb1, float4 sb2, uint32_t npop,
                                              uint32_t words, uint64_t *sub_masks_cell)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub_of_lane = lane >> 3, cand_of_lane = lane & 7u;
    uint32_t base = 0;
    for (uint32_t w = 0; w < words; ++w) {
        const unsigned long long m = uniform_u64(s_masks[w]);
        if ((m >> lane) & 1ull) s_long[base + bits_below_lane(m)] = (uint16_t)(w * 64u + lane);
        base += (uint32_t)__builtin_popcountll(m);
    }
    if (lane < 32u) s_sub[lane] = 0ull;
    __builtin_amdgcn_wave_barrier();
    CellBounds sb;
    sb.L0 = sb0.x; sb.L1 = sb0.y; sb.a0 = sb0.z; sb.a1 = sb0.w;
    sb.b0 = sb1.x; sb.b1 = sb1.y; sb.C0 = sb1.z; sb.C1 = sb1.w;
    sb.wC0 = sb2.x; sb.wC1 = sb2.y; sb.wH0 = sb2.z; sb.wH1 = sb2.w;
    // two sweeps, eight candidates per round: the threshold of a sub-cell needs every candidate's upper bound first
    const uint32_t rounds = (npop + 7u) >> 3;
    float Usub = 3.0e38f;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t pos = r * 8u + cand_of_lane;
        if (pos < npop) {
            const float4 c = s_cent[s_long[pos]];
            Usub = fminf(Usub, key_range(sb, c.x, c.y, c.z, c.w).hi);
        }
    }
    const float Us = mask_threshold(group8_min(Usub));
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t pos = r * 8u + cand_of_lane;
        if (pos < npop) {
            const uint32_t j = s_long[pos];
            const float4 c = s_cent[j];
            if (key_range(sb, c.x, c.y, c.z, c.w).lo <= Us) atomicOr(&s_sub[sub_of_lane * 4u + (j >> 6)], 1ull << (j & 63u));
        }
    }
    __builtin_amdgcn_wave_barrier();
    // lane 4 s + w holds word w of sub-cell s: stored (the scan kernel reads them), counted, and the single survivor named
    const unsigned long long mine = lane < 32u ? s_sub[lane] : 0ull;
    if (lane < 32u) sub_masks_cell[lane] = mine;
    uint32_t cnt = (uint32_t)__builtin_popcountll(mine);
    uint32_t one = mine ? (lane & 3u) * 64u + (uint32_t)__builtin_ctzll(mine) : 0u;
    cnt += dpp_u32<kDppXor1>(cnt); one += dpp_u32<kDppXor1>(one);
    cnt += dpp_u32<kDppXor2>(cnt); one += dpp_u32<kDppXor2>(one);      // every lane of the four: the sub-cell's count / its one candidate
    uint2 r;
    r.x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane * 4u) << 2), (int)cnt);      // -> lane s < 8
    r.y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane * 4u) << 2), (int)one);
    return r;
}


constexpr uint32_t kAffineFloats = 24;                            // per sub-cell, in 32-bit words: 7 features x (alpha, 3 beta, Rmin, Rmax) as
                                                                  // 42 binary16 values (words 0 .. 20), [21] = C1, [22] = wH1 of the
                                                                  // sub-cell's bounds (binary32), one word of padding: 96 bytes
// A sub-cell's model as the test reads it: element i < 42 = the binary16 value i widened (exact), [42] = C1, [43] = wH1.
// Binary16 costs the test nothing: the residual ranges are taken against the STORED (rounded) model and rounded outward
// (k_sub_affine), so any stored model is a valid one; what the coarser coefficients lose in fit is ~2^-11 of the residuals.
struct HalfModel {
    uint32_t q[kAffineFloats];
    __device__ __forceinline__ float operator[](int i) const
    {
        if (i >= 42) return bits_to_float(q[21 + (i - 42)]);
        const uint32_t h = (i & 1) ? q[i >> 1] >> 16 : q[i >> 1] & 0xFFFFu;
        return __half2float(__ushort_as_half((unsigned short)h));
    }
    __device__ __forceinline__ void load(const float *table, uint64_t sub_cell)
    {
        const uint4 *mp = reinterpret_cast<const uint4 *>(table + sub_cell * kAffineFloats);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const uint4 v = mp[t];
            q[4 * t] = v.x; q[4 * t + 1] = v.y; q[4 * t + 2] = v.z; q[4 * t + 3] = v.w;
        }
    }
};

// The dominance test (see k_cube_small's header for the derivation): true when, for every colour of the sub-cell whose model
// is `mdl` (HalfModel), K_j - K_i >= 2^-11 U, U >= the float upper bound of key_i over the
// sub-cell.  Binary32 throughout: the lower bound is accumulated together with `mag`, the sum of the magnitudes of everything
// that enters it; the evaluation performs fewer than 40 roundings, each relative to a partial sum <= mag, and the inputs' own
// errors -- differences of squares (w2, c0), N = a^2 + b^2 - C^2 -- are bounded by 3u times sums of squares that mag holds as
// well (the features they multiply, wC and wH, are <= 1), so the computed value is within 64 u mag of the real one
// (u = 2^-24) and subtracting that keeps the bound rigorous.  What the error term costs: mag ~ 3e4 for colours, 64 u mag ~ 0.1
// in units of dE^2 -- against differences of tens for a box next to a boundary.
template <typename Model>
__device__ __forceinline__ bool dominated(const Model &mdl, const float4 cj, const float4 ci, float U)
{
    const float Lj2 = cj.x * cj.x, Li2 = ci.x * ci.x, Cj2 = cj.w * cj.w, Ci2 = ci.w * ci.w;
    const float Aj = fmaf(cj.z, cj.z, cj.y * cj.y), Ai = fmaf(ci.z, ci.z, ci.y * ci.y);
    const float dC = cj.w - ci.w;
    const float w[7] = {-2.0f * (cj.x - ci.x), Cj2 - Ci2, -2.0f * dC, -2.0f * (cj.y - ci.y), -2.0f * (cj.z - ci.z), 2.0f * dC,
                        (Aj - Cj2) - (Ai - Ci2)};
    float d = Lj2 - Li2, gx = 0.0f, gy = 0.0f, gz = 0.0f, mx = 0.0f, my = 0.0f, mz = 0.0f;
    float mag = (Lj2 + Li2) + (Cj2 + Ci2) + ((Aj + Cj2) + (Ai + Ci2));
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const float t = w[m] * mdl[6 * m];
        d += t;
        mag += fabsf(t);
        const float tx = w[m] * mdl[6 * m + 1], ty = w[m] * mdl[6 * m + 2], tz = w[m] * mdl[6 * m + 3];
        gx += tx; gy += ty; gz += tz;
        mx += fabsf(tx); my += fabsf(ty); mz += fabsf(tz);
        const float r = fminf(w[m] * mdl[6 * m + 4], w[m] * mdl[6 * m + 5]);
        d += r;
        mag += fabsf(r);
    }
    d -= 1.5f * ((fabsf(gx) + fabsf(gy)) + fabsf(gz));
    mag += 1.5f * ((mx + my) + mz);
    const float cs = mdl[42] + ci.w;                              // C1 + Ci
    const float delta = mdl[43] * (9.5367431640625e-07f * (cs * cs));   // wH1 2^-20 (C1 + Ci)^2
    d -= delta;
    mag += delta;
    // 64 u = 2^-18; 2^-11 (1 + 2^-7)
    return d - 3.814697265625e-06f * mag >= 0.000492095947265625f * U;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// k_cube_stage.  LDS: [centroids kpad x 16 B, kpad = k rounded up to 64][bins k x 32 B (SUMS)]
//                     [candidate list 4 waves x 32 u32][cell masks 4 waves x words u64]
// SUMS = false (output pass of find / reduce in replace mode): no image -- every colour of every cell counts
// once, nothing is accumulated; agg, sub_agg, work and sums are unused.
// flags bit 0: also write the 512 per-colour labels of single-candidate cells (debug / statistics).
// stats (optional, u64[6]): single-candidate cells, other cells, sub-cells decided by their bounds, sub-cells
// scanned, candidates summed over the scanned sub-cells, cells with more than kMaxListed candidates
// ------------------------------------------------------------------------------------------
// (6 waves per SIMD = 80 VGPRs, no spills: 35.9 -> 34.9 us against the 87 the allocator takes by itself; the scan kernel
// forced from 71 to 64 VGPRs spills 7 dwords and loses 2 us)
template <typename LabelT, bool SUMS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_cube_stage(const int64_t *__restrict__ agg,
                                                       const int64_t *__restrict__ sub_agg,
                                                       const uint32_t *__restrict__ work,
                                                       const CellBounds *__restrict__ bounds,
                                                       const CellBounds *__restrict__ sub_bounds,
                                                       const Centroid *__restrict__ cent, uint32_t k,
                                                       uint64_t *__restrict__ masks_out,
                                                       CellWork *__restrict__ cell_work,
                                                       LabelT *__restrict__ colour_labels,
                                                       uint16_t *__restrict__ sub_table,
                                                       int64_t *__restrict__ sums, uint32_t n_rows,
                                                       uint32_t flags, unsigned long long *__restrict__ stats)
{
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 63u) & ~63u;
    const uint32_t words = kpad / 64u;
    float4 *s_cent = smem4;
    unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem4 + kpad);
    const uint32_t n_bins = SUMS ? 4u * k : 0u;
    uint32_t *s_list_all = reinterpret_cast<uint32_t *>(bins + n_bins);
    unsigned long long *s_masks_all = reinterpret_cast<unsigned long long *>(s_list_all + (kBlock / 64) * kMaxListed);
    // long lists (k <= 256): [4 waves][kMaxLong] u16 candidate list, [4 waves][8 sub-cells][4 words] u64 masks
    unsigned long long *s_sub_all = s_masks_all + (kBlock / 64) * words;
    uint16_t *s_long_all = reinterpret_cast<uint16_t *>(s_sub_all + (kBlock / 64) * 32u);

    for (uint32_t i = threadIdx.x; i < kpad; i += kBlock) {
        float4 v = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);       // key ~ 1e36: never a candidate
        if (i < k) { const Centroid c = cent[i]; v = make_float4(c.L, c.a, c.b, c.C); }
        s_cent[i] = v;
    }
    for (uint32_t i = threadIdx.x; i < n_bins; i += kBlock) bins[i] = 0ull;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t *s_list = s_list_all + wv * kMaxListed;
    unsigned long long *s_masks = s_masks_all + wv * words;
    unsigned long long *s_sub = s_sub_all + wv * 32u;
    uint16_t *s_long = s_long_all + wv * kMaxLong;
    uint64_t *sub_masks_out = masks_out + (uint64_t)kCells * ((k + 63u) / 64u);      // [cell][8][4], k <= 256
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + wv);
    const uint32_t n_waves = gridDim.x * (kBlock / 64);
    const uint32_t sub_of_lane = lane >> 3, cand_of_lane = lane & 7u;
    unsigned long long st_single = 0, st_multi = 0, st_decided = 0, st_scanned = 0, st_cands = 0, st_unlisted = 0;

    const uint32_t n_work = SUMS ? __builtin_amdgcn_readfirstlane(work[0]) : kCells;
    // the small per-cell data of the NEXT cell are requested while the current one is worked on
    uint32_t cell_n = 0;
    // All of it through VECTOR loads (kmg_table_dev.h opaque_vgpr_zero): as scalar loads the requests were waited for at
    // the first LDS access of the current cell.  The work-list entry is requested one cell further ahead still, so that no
    // request starts with a load it has to wait for.  (34.5 -> 32 us)
    const uint32_t vz = opaque_vgpr_zero();
    float cbv_n = 0.0f;                                            // lane i < 12: float i of the next cell's bounds
    long long sagg_n = 0, cagg_n = 0, scnt_n = 1;
    // (a wave beyond the end of the work list must not use what lies behind it as a cell: entry 0 is always valid)
    const uint32_t safe_w = wave < n_work ? wave : 0u;
    uint32_t cell_nn = SUMS ? work[1u + safe_w + vz] : safe_w;
#define KMG_REQUEST_CELL(wi_)                                                                                    \
    do {                                                                                                         \
        cell_n = __builtin_amdgcn_readfirstlane(cell_nn);           /* requested one call earlier */             \
        const uint32_t w2_ = (wi_) + n_waves < n_work ? (wi_) + n_waves : safe_w;   /* past the end: a harmless repeat */ \
        cell_nn = SUMS ? work[1u + w2_ + vz] : w2_;                                                              \
        cbv_n = reinterpret_cast<const float *>(bounds + cell_n)[lane & 15u];   /* lane i: float i (VMEM: not tied to LDS waits) */ \
        if (SUMS) {                                                                                              \
            sagg_n = sub_agg[(uint64_t)cell_n * 32u + (lane & 31u)];        /* lane 4 s + j: sum j of sub-cell s */ \
            scnt_n = sub_agg[(uint64_t)cell_n * 32u + 4u * (lane & 7u) + 3u];   /* lane s < 8: pixels in sub-cell s */ \
            cagg_n = agg[4ull * cell_n + (lane & 3u)];                                                           \
        }                                                                                                        \
    } while (0)
    KMG_REQUEST_CELL(wave);
    for (uint32_t wi = wave; wi < n_work; wi += n_waves) {
        const uint32_t cell = cell_n;
        CellBounds cb;
        cb.L0 = lane_value(cbv_n, 0); cb.L1 = lane_value(cbv_n, 1); cb.a0 = lane_value(cbv_n, 2); cb.a1 = lane_value(cbv_n, 3);
        cb.b0 = lane_value(cbv_n, 4); cb.b1 = lane_value(cbv_n, 5); cb.C0 = lane_value(cbv_n, 6); cb.C1 = lane_value(cbv_n, 7);
        cb.wC0 = lane_value(cbv_n, 8); cb.wC1 = lane_value(cbv_n, 9); cb.wH0 = lane_value(cbv_n, 10); cb.wH1 = lane_value(cbv_n, 11);
        const long long sagg = sagg_n, cagg = cagg_n, scnt = scnt_n;
        KMG_REQUEST_CELL(wi + n_waves);
        // the sub-cell bounds of THIS cell: requested now, needed after the cell's candidates are known
        const float4 *sbp = reinterpret_cast<const float4 *>(sub_bounds + (uint64_t)cell * 8u + sub_of_lane);
        const float4 sb0 = sbp[0], sb1 = sbp[1], sb2 = sbp[2];

        uint16_t *sub = sub_table + cell * 8u;
        uint16_t *cell_entry = sub_table + kSubCells + cell;       // 8x8x8 summary, same encoding
        uint32_t *pair_entry_ptr = reinterpret_cast<uint32_t *>(sub_table + kSubCells + kCells) + cell;   // k <= 256
        LabelT *cell_labels = colour_labels + (uint64_t)cell * kCellColours;
        CellWork *cw = cell_work + cell;

        // ---- 1. candidates of the cell ----
        unsigned long long mw[4] = {0ull, 0ull, 0ull, 0ull};      // the mask itself when words <= 4 (k <= 256)
        uint32_t npop = 0, first = 0;
        if (words <= 4u) {
            float U = 3.0e38f, lo[4];
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                lo[w] = 3.0e38f;
                if (w < words) {
                    const float4 c = s_cent[w * 64u + lane];
                    const KeyRange r = key_range(cb, c.x, c.y, c.z, c.w);
                    lo[w] = r.lo;
                    U = fminf(U, r.hi);                            // padding entries: hi ~ 1e36
                }
            }
            const float Us = mask_threshold(wave_min(U));
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                if (w < words) {
                    mw[w] = __ballot(w * 64u + lane < k && lo[w] <= Us);
                    if (npop == 0u && mw[w]) first = w * 64u + (uint32_t)__builtin_ctzll(mw[w]);
                    npop += (uint32_t)__builtin_popcountll(mw[w]);
                }
            }
        } else {
            float U = 3.0e38f;
            for (uint32_t j = lane; j < k; j += 64u) {
                const float4 c = s_cent[j];
                U = fminf(U, key_range(cb, c.x, c.y, c.z, c.w).hi);
            }
            const float Us = mask_threshold(wave_min(U));
            for (uint32_t w = 0; w < words; ++w) {
                const uint32_t j = w * 64u + lane;
                const float4 c = s_cent[j];
                const unsigned long long m = __ballot(j < k && key_range(cb, c.x, c.y, c.z, c.w).lo <= Us);
                if (lane == 0u) s_masks[w] = m;
                if (npop == 0u && m) first = j - lane + (uint32_t)__builtin_ctzll(m);
                npop += (uint32_t)__builtin_popcountll(m);
            }
        }
        {
            uint64_t *out = masks_out + (uint64_t)cell * ((k + 63u) / 64u);
            if (words <= 4u) {
#pragma unroll
                for (uint32_t w = 0; w < 4u; ++w)
                    if (w < words && lane == 0u) out[w] = mw[w];
            } else {
                for (uint32_t w = lane; w < words; w += 64u) out[w] = s_masks[w];
            }
        }

        if (npop == 1u) {
            // the whole cell belongs to `first`: sums from the cell table, no per-colour traffic
            if (sizeof(LabelT) == 1) {
                if (lane == 0u) *pair_entry_ptr = pair_entry(first, first, 0u, 0u, 0u);
            } else {
                if (lane < 8u) sub[lane] = (uint16_t)first;
                if (lane == 0u) *cell_entry = (uint16_t)first;
            }
            if (lane == 0u) { cw->npop = 1u; cw->scan_set = 0u; }
            if (SUMS && lane < 4u) atomicAdd(bins + 4ull * first + lane, (unsigned long long)cagg);
            if (flags & 1u) {
#pragma unroll
                for (uint32_t s = 0; s < 8u; ++s) store_labels64(cell_labels + s * 64u, first, lane);
            }
            st_single += 1;
            continue;
        }
        st_multi += 1;
        // ---- 2. sub-cell stage: list the candidates, bound each over each sub-cell ----
        const bool listed = npop <= kMaxListed;
        uint32_t my_cand = 0;                                       // lane p < npop: the p-th candidate
        unsigned long long br[4] = {0ull, 0ull, 0ull, 0ull};       // round r: bit 8 s + c = sub-cell s keeps candidate 8 r + c
        bool long_cell = false;
        uint32_t long_cnt = 0u, long_one = 0u;                      // long list: lane s < 8 = candidates sub-cell s keeps / the one if it is one
        if (listed) {
            uint32_t base = 0;
            if (words <= 4u) {
#pragma unroll
                for (uint32_t w = 0; w < 4u; ++w) {
                    if (w < words) {
                        if ((mw[w] >> lane) & 1ull) s_list[base + bits_below_lane(mw[w])] = w * 64u + lane;
                        base += (uint32_t)__builtin_popcountll(mw[w]);
                    }
                }
            } else {
                for (uint32_t w = 0; w < words; ++w) {
                    const unsigned long long m = uniform_u64(s_masks[w]);
                    if ((m >> lane) & 1ull) s_list[base + bits_below_lane(m)] = w * 64u + lane;
                    base += (uint32_t)__builtin_popcountll(m);
                }
            }
            __builtin_amdgcn_wave_barrier();
            my_cand = s_list[lane & (kMaxListed - 1u)];
            __builtin_amdgcn_wave_barrier();
            if (lane < kMaxListed) cw->list[lane] = (uint16_t)(lane < npop ? my_cand : 0u);

            CellBounds sb;
            sb.L0 = sb0.x; sb.L1 = sb0.y; sb.a0 = sb0.z; sb.a1 = sb0.w;
            sb.b0 = sb1.x; sb.b1 = sb1.y; sb.C0 = sb1.z; sb.C1 = sb1.w;
            sb.wC0 = sb2.x; sb.wC1 = sb2.y; sb.wH0 = sb2.z; sb.wH1 = sb2.w;
            const uint32_t rounds = (npop + 7u) >> 3;
            // the smallest upper bound AND the candidate that has it: the bound's bit pattern (a non-negative float: patterns
            // order like values) rounded up to a multiple of 32 ulps, the list position in the 5 bits that frees.  The threshold
            // is taken from the rounded bound -- a little above the exact one, which only keeps candidates.
            uint32_t Ubest = 0x7F7FFFE0u;
            float lo[4];
#pragma unroll
            for (uint32_t r = 0; r < 4u; ++r) {
                lo[r] = 3.0e38f;
                if (r < rounds) {
                    const uint32_t pos = r * 8u + cand_of_lane;    // < 32: lanes pos and pos + 32 hold the same candidate
                    const uint32_t j = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)my_cand);
                    if (pos < npop) {
                        const float4 c = s_cent[j];
                        const KeyRange kr = key_range(sb, c.x, c.y, c.z, c.w);
                        lo[r] = kr.lo;
                        Ubest = min(Ubest, ((float_to_bits(kr.hi) + 31u) & ~31u) | pos);
                    }
                }
            }
            Ubest = group8_min_u32(Ubest);
            const float Us = mask_threshold(bits_to_float(Ubest & ~31u));
#pragma unroll
            for (uint32_t r = 0; r < 4u; ++r)
                if (r < rounds) br[r] = __ballot(lo[r] <= Us);
            if (lane < 4u) cw->br[lane] = lane == 0u ? br[0] : (lane == 1u ? br[1] : (lane == 2u ? br[2] : br[3]));
        } else if (words <= 4u && npop <= kMaxLong) {
            // long list (rare on noise, the heavy cells of a photograph): long_list_stage
            long_cell = true;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w)
                if (w < words && lane == 0u) s_masks[w] = mw[w];
            __builtin_amdgcn_wave_barrier();
            const uint2 lr = long_list_stage(s_cent, s_masks, s_long, s_sub, sb0, sb1, sb2, npop, words, sub_masks_out + (uint64_t)cell * 32u);
            long_cnt = lr.x;
            long_one = lr.y;
            st_unlisted += 1;
        } else {
            st_unlisted += 1;
        }

        // what each sub-cell needs -- empty / decided as a whole (one candidate) / scan its colours -- for all
        // eight at once: lane s < 8 is sub-cell s (a scalar loop over the sub-cells costs ~10x the issue slots)
        uint32_t sm_v = 0u;                                         // lane s: candidates (list positions) sub-cell s keeps
        if (lane < 8u) {
#pragma unroll
            for (uint32_t r = 0; r < 4u; ++r) sm_v |= ((uint32_t)(br[r] >> (8u * lane)) & 0xFFu) << (8u * r);
        }
        const bool occupied_v = lane < 8u && (SUMS ? scnt != 0 : true);
        const bool decided_v = occupied_v && ((listed && __builtin_popcount(sm_v) == 1) || (long_cell && long_cnt == 1u));
        const uint32_t decided_set = (uint32_t)__ballot(decided_v);
        const uint32_t scan_set = (uint32_t)__ballot(occupied_v && !decided_v);
        // label of a decided sub-cell, at lane s
        uint32_t X_v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((sm_v ? (uint32_t)__builtin_ctz(sm_v) : 0u)) << 2), (int)my_cand);
        if (long_cell) X_v = long_one;
        {
            // its 64 labels: lane l owns the colours 8 l .. 8 l + 7 of the cell (sub-cell l >> 3), one store
            const uint32_t X = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane >> 3) << 2), (int)X_v);
            if ((decided_set >> (lane >> 3)) & 1u) {
                if (sizeof(LabelT) == 1) {
                    const uint32_t x4 = X * 0x01010101u;
                    *reinterpret_cast<uint2 *>(cell_labels + lane * 8u) = make_uint2(x4, x4);
                } else {
                    const uint32_t x2 = X * 0x00010001u;
                    *reinterpret_cast<uint4 *>(cell_labels + lane * 8u) = make_uint4(x2, x2, x2, x2);
                }
            }
            // its sums: lane 4 s + j holds sum j of sub-cell s
            const uint32_t Xs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane >> 2) & 7u) << 2), (int)X_v);
            if (SUMS && lane < 32u && ((decided_set >> (lane >> 2)) & 1u))
                atomicAdd(bins + 4ull * Xs + (lane & 3u), (unsigned long long)sagg);
        }
        if (sizeof(LabelT) != 1) {
            // summaries: decided -> the label, empty -> kSubEmpty; scanned ones are written by k_cube_scan
            if (lane < 8u && !((scan_set >> lane) & 1u)) sub[lane] = (uint16_t)(decided_v ? X_v : (uint32_t)kSubEmpty);
        }
        st_decided += (uint32_t)__builtin_popcount(decided_set);
        st_scanned += (uint32_t)__builtin_popcount(scan_set);
        if (stats) {
            const uint32_t c = (scan_set >> (lane & 7u)) & 1u ? (listed ? (uint32_t)__builtin_popcount(sm_v) : (long_cell ? long_cnt : npop)) : 0u;
            st_cands += wave_add_u32(lane < 8u ? c : 0u);
        }
        if (lane == 0u) {
            cw->npop = npop;
            cw->scan_set = scan_set | (listed ? 0x100u : 0u) | (long_cell ? kLongFlag : 0u) | (decided_set << 16);
            if (sizeof(LabelT) == 1) *pair_entry_ptr = kPairPending;
            if (long_cell && scan_set && (flags & kCubeSplitLong)) {
                uint32_t *lists = reinterpret_cast<uint32_t *>(cell_work + kCells);
                const uint32_t ls = wave & (kLongSegs - 1u);
                lists[kLongCells + ls * kLongSegCap + atomicAdd(lists + kLongCount + ls, 1u)] = cell;
            }
        }
    }
#undef KMG_REQUEST_CELL

    if (stats && lane == 0u) {
        atomicAdd(stats + 0, st_single); atomicAdd(stats + 1, st_multi); atomicAdd(stats + 2, st_decided);
        atomicAdd(stats + 3, st_scanned); atomicAdd(stats + 4, st_cands); atomicAdd(stats + 5, st_unlisted);
    }
    if (SUMS) flush_bins(bins, k, 1u, 4u * k, sums, n_rows);
}

// ------------------------------------------------------------------------------------------
// k_cube_scan.  LDS: [centroids kpad x 16 B][bins repl x (k x 32 + 32) B (SUMS)][candidates 4 waves x 32 x 16 B]
//                    [labels 4 waves x 512]
// ------------------------------------------------------------------------------------------
// (eight waves per workgroup: they share the centroid table and FOUR copies of the bins in the LDS two workgroups of four waves
// spent on two copies each -- the same waves per CU, half the same-address atomics per copy)
constexpr int kScanBlock = 512;
template <typename LabelT, bool SUMS>
__global__ __launch_bounds__(kScanBlock) void k_cube_scan(const uint32_t *__restrict__ hist,
                                                      const int64_t *__restrict__ sub_agg,
                                                      const uint32_t *__restrict__ work,
                                                      const Centroid *__restrict__ cent, uint32_t k,
                                                      const float4 *__restrict__ lab_table,
                                                      const uint64_t *__restrict__ masks,
                                                      const CellWork *__restrict__ cell_work,
                                                      LabelT *__restrict__ colour_labels,
                                                      uint16_t *__restrict__ sub_table,
                                                      int64_t *__restrict__ sums, uint32_t n_rows, uint32_t repl,
                                                      uint32_t flags)
{
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 63u) & ~63u;
    float4 *s_cent = smem4;
    unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem4 + kpad);
    const uint32_t bin_stride = 4u * k + 4u;                      // u64 per copy (+ 32 B: next copy, other banks)
    const uint32_t n_bins = SUMS ? repl * bin_stride : 0u;
    float4 *s_cc_all = reinterpret_cast<float4 *>(bins + n_bins);              // [4 waves][kMaxListed]: listed candidates by position
    LabelT *s_lbl_all = reinterpret_cast<LabelT *>(s_cc_all + (kScanBlock / 64) * kMaxListed);   // [4 waves][512]: labels of the cell
    for (uint32_t i = threadIdx.x; i < kpad; i += kScanBlock) {
        float4 v = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);
        if (i < k) { const Centroid c = cent[i]; v = make_float4(c.L, c.a, c.b, c.C); }
        s_cent[i] = v;
    }
    for (uint32_t i = threadIdx.x; i < n_bins; i += kScanBlock) bins[i] = 0ull;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + wv);
    const uint32_t n_waves = gridDim.x * (kScanBlock / 64);
    unsigned long long *my_bins = bins + (uint64_t)(lane & (repl - 1u)) * bin_stride;
    float4 *s_cc = s_cc_all + wv * kMaxListed;
    LabelT *s_lbl = s_lbl_all + wv * kCellColours;

    const uint32_t *lists = reinterpret_cast<const uint32_t *>(cell_work + kCells);
    const uint32_t n_work = SUMS ? __builtin_amdgcn_readfirstlane(work[0]) : kCells;
    const uint32_t *cells = SUMS ? work + 1 : nullptr;
    // kCubeSplitLong: the stage kernel's list of cells with long candidate lists comes first, FOUR waves per cell (a pair of its
    // sub-cells each); the walk over the work list then skips those cells
    const bool split_long = (flags & kCubeSplitLong) != 0u;
    uint32_t long_end = 0u, n_long4 = 0u;                           // lane s < kLongSegs: entries of the segments 0 .. s; 4 x all of them
    if (split_long) {
        long_end = lane < kLongSegs ? lists[kLongCount + lane] : 0u;
        for (uint32_t off = 1; off < kLongSegs; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)long_end, off, 64);
            if (lane >= off) long_end += up;
        }
        n_long4 = 4u * lane_value(long_end, kLongSegs - 1u);
    }
    // (Requesting the NEXT cell's work record ahead -- one vector load, lane i = dword i -- was measured: the 8 registers
    // it holds cost a wave per SIMD, 73.6 -> 76.6 us.)
    const uint32_t wi_end = n_work + n_long4;
    for (uint32_t wi = wave; wi < wi_end; wi += n_waves) {
        uint32_t cell, quarter = 4u;                                // quarter < 4: only that pair of the cell's sub-cells to scan
        if (wi < n_long4) {
            const uint32_t e = wi >> 2;
            const uint32_t ls = (uint32_t)__builtin_popcountll(__ballot(lane < kLongSegs && long_end <= e));
            const uint32_t first = ls ? lane_value(long_end, ls - 1u) : 0u;
            cell = __builtin_amdgcn_readfirstlane(lists[kLongCells + ls * kLongSegCap + (e - first)]);
            quarter = wi & 3u;
        } else {
            const uint32_t wj = wi - n_long4;
            cell = cells ? __builtin_amdgcn_readfirstlane(cells[wj]) : wj;
        }
        const CellWork *cw = cell_work + cell;
        uint32_t scan_set = __builtin_amdgcn_readfirstlane(cw->scan_set);
        if ((scan_set & 0xFFu) == 0u) continue;                     // one candidate, or every sub-cell decided
        const bool listed = (scan_set & 0x100u) != 0u;
        const bool long_cell = (scan_set & kLongFlag) != 0u;
        scan_set &= 0xFFu;
        if (quarter < 4u) {
            uint32_t m = scan_set;
            for (uint32_t t = 0; t < 2u * quarter; ++t) m &= m - 1u;
            const uint32_t b0 = m & (0u - m);
            m &= m - 1u;
            scan_set = b0 | (m & (0u - m));
            if (scan_set == 0u) continue;
        } else if (split_long && long_cell) {
            continue;                                               // (taken by four waves above)
        }
        const uint32_t npop = __builtin_amdgcn_readfirstlane(cw->npop);
        const uint32_t my_cand = listed ? (uint32_t)cw->list[lane & (kMaxListed - 1u)] : 0u;   // lane p: the p-th candidate ...
        const unsigned long long br0 = listed ? uniform_u64(cw->br[0]) : 0ull, br1 = listed ? uniform_u64(cw->br[1]) : 0ull;
        const unsigned long long br2 = listed ? uniform_u64(cw->br[2]) : 0ull, br3 = listed ? uniform_u64(cw->br[3]) : 0ull;
        const long long sagg = SUMS ? sub_agg[(uint64_t)cell * 32u + (lane & 31u)] : 0;      // lane 4 s + j: sum j of sub-cell s
        const uint64_t *cmask = masks + (uint64_t)cell * ((k + 63u) / 64u);
        const uint64_t *smask = masks + (uint64_t)kCells * ((k + 63u) / 64u) + (uint64_t)cell * 32u;   // long cells: [8][4]
        uint16_t *sub = sub_table + cell * 8u;
        LabelT *cell_labels = colour_labels + (uint64_t)cell * kCellColours;
        auto submask_of = [&](uint32_t s) {                        // bit p = the sub-cell keeps candidate p of the list
            return (((uint32_t)(br0 >> (8u * s)) & 0xFFu)) | (((uint32_t)(br1 >> (8u * s)) & 0xFFu) << 8) |
                   (((uint32_t)(br2 >> (8u * s)) & 0xFFu) << 16) | (((uint32_t)(br3 >> (8u * s)) & 0xFFu) << 24);
        };

        // TWO sub-cells per step (two independent dependency chains per lane); the next pair's loads are in
        // flight while the current one is scanned.  Two register sets A / B alternate, and the loop body exists
        // once per set: with one loop-carried set the compiler moves components of the freshly loaded (L, a, b, C)
        // right behind the load, i.e. waits for it, and the prefetch is gone.
        float4 A_v0, A_v1, B_v0, B_v1;
        uint32_t A_c0 = 1u, A_c1 = 1u, B_c0 = 1u, B_c1 = 1u, A_s0 = 0u, A_s1 = 8u, B_s0 = 0u, B_s1 = 8u;
#define KMG_REQUEST_COLOURS(X)                                                                                   \
        do {                                                                                                     \
            /* unconditional (an exhausted set re-requests sub-cell 0, unused): a conditional request makes the  \
               compiler shuffle -- and therefore wait for -- the registers it has just requested */               \
            X##_s0 = scan_set ? (uint32_t)__builtin_ctz(scan_set) : 8u;                                          \
            scan_set &= scan_set - 1u;                                                                           \
            X##_s1 = scan_set ? (uint32_t)__builtin_ctz(scan_set) : 8u;                                          \
            scan_set &= scan_set - 1u;                                                                           \
            const uint32_t c0_ = cell * kCellColours + (X##_s0 & 7u) * 64u + lane;                               \
            const uint32_t c1_ = cell * kCellColours + (X##_s1 & 7u) * 64u + lane;                               \
            X##_v0 = lab_table[c0_];                                                                             \
            X##_v1 = lab_table[c1_];                                /* s1 == 8: sub-cell 0 again, unused */      \
            if (SUMS) { X##_c0 = hist[c0_]; X##_c1 = hist[c1_]; }                                                \
        } while (0)
        KMG_REQUEST_COLOURS(A);
        // the listed candidates' (L, a, b, C) by list position, in LDS: one broadcast read per candidate visit
        if (listed && (lane & (kMaxListed - 1u)) < npop) s_cc[lane & (kMaxListed - 1u)] = s_cent[my_cand];
        uint32_t label_set = 0u;
        __builtin_amdgcn_wave_barrier();

        // Measured on gfx950 (tools/issue_rate.hip, cycles per wave-instruction per SIMD): plain VALU 2.6,
        // v_readlane 8, v_cmp + v_cndmask chains 4.5, SALU 4.35, an IEEE divide ~47.  Hence: no readlane and no
        // compare / select per visit -- the key carries the candidate's list position in its 5 low mantissa
        // bits, so arg-min and runner-up are one integer min / med3 (keys are non-negative floats: their bit
        // patterns order like the values; equal keys order by position = centroid index) -- and hardware
        // reciprocals for the per-colour weights.  Both only perturb the ORDERING key (kmg_math.h).
        auto scan_pair = [&](const uint32_t s0, const uint32_t s1, const float4 v0, const float4 v1, const uint32_t cnt0,
                             const uint32_t cnt1) {
            const PixelTerms pt0 = pixel_terms_fast(v0.x, v0.y, v0.z, v0.w), pt1 = pixel_terms_fast(v1.x, v1.y, v1.z, v1.w);
            uint32_t ix0 = 0u, ix1 = 0u;
            if (listed) {
                uint32_t b0 = 0x7F7FFFFFu, r0 = 0x7F7FFFFFu, b1 = 0x7F7FFFFFu, r1 = 0x7F7FFFFFu;   // smallest / runner-up
                const uint32_t sm = submask_of(s0) | (s1 < 8u ? submask_of(s1) : 0u);
                // the two colours of a lane as the halves of packed-f32 operands: cie94_key's operations in the same order (the
                // same two floats), 13 vector instructions for the pair instead of 24.  Packed fp32 issues at half rate on gfx950
                // (tools/valu_rate.hip), so the arithmetic takes as long; what this kernel is short of is issue slots (53 % VALU
                // busy beside as many scalar instructions): cube pass 121 -> 119 us.
                const f32x2 qL = {pt0.L, pt1.L}, qa = {pt0.a, pt1.a}, qb = {pt0.b, pt1.b}, qC = {pt0.C, pt1.C};
                const f32x2 qwC = {pt0.wC, pt1.wC}, qwH = {pt0.wH, pt1.wH};
                for (uint32_t m = sm; m; m &= m - 1u) {
                    const uint32_t pos = (uint32_t)__builtin_ctz(m);
                    const float4 c = s_cc[pos];
                    const f32x2 dL = qL - c.x, da = qa - c.y, db = qb - c.z, dC = qC - c.w;
                    const f32x2 dC2 = dC * dC;
                    const f32x2 t = __builtin_elementwise_fma(db, db, da * da);
                    f32x2 h = t - dC2;
                    h.x = fmaxf(h.x, 0.0f); h.y = fmaxf(h.y, 0.0f);
                    const f32x2 key = __builtin_elementwise_fma(h, qwH, __builtin_elementwise_fma(dC2, qwC, dL * dL));
                    uint32_t u0, u1;                               // (key & ~31) | pos: one bit-field insert each (pos is wave-uniform)
                    asm("v_bfi_b32 %0, 31, %1, %2" : "=v"(u0) : "s"(pos), "v"(float_to_bits(key.x)));
                    asm("v_bfi_b32 %0, 31, %1, %2" : "=v"(u1) : "s"(pos), "v"(float_to_bits(key.y)));
                    r0 = umed3(u0, b0, r0); b0 = min(b0, u0);
                    r1 = umed3(u1, b1, r1); b1 = min(b1, u1);
                }
                uint32_t p0 = b0 & 31u, p1 = b1 & 31u;
                // near-tie repair (kmg_math.h): rare; decided by the literal distance, first minimum wins
                const float thr0 = tie_threshold(bits_to_float(b0 & ~31u)), thr1 = tie_threshold(bits_to_float(b1 & ~31u));
                const bool near0 = bits_to_float(r0 & ~31u) <= thr0, near1 = bits_to_float(r1 & ~31u) <= thr1;
                if (__ballot(near0 || near1)) {
                    float lb0 = 100000.0f, lb1 = 100000.0f;       // find_centroid.wgsl:29-30
                    uint32_t li0 = 0u, li1 = 0u;
                    for (uint32_t m = sm; m; m &= m - 1u) {
                        const uint32_t pos = (uint32_t)__builtin_ctz(m);
                        const float4 c = s_cc[pos];
                        if (near0 && cie94_key(pt0, c.x, c.y, c.z, c.w) <= thr0) {
                            const float d = cie94_c(v0.x, v0.y, v0.z, v0.w, c.x, c.y, c.z, c.w);
                            if (d < lb0) { lb0 = d; li0 = pos; }
                        }
                        if (near1 && cie94_key(pt1, c.x, c.y, c.z, c.w) <= thr1) {
                            const float d = cie94_c(v1.x, v1.y, v1.z, v1.w, c.x, c.y, c.z, c.w);
                            if (d < lb1) { lb1 = d; li1 = pos; }
                        }
                    }
                    p0 = near0 ? li0 : p0;
                    p1 = near1 ? li1 : p1;
                }
                ix0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p0 << 2), (int)my_cand);
                ix1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p1 << 2), (int)my_cand);
            } else {
                // more than kMaxListed candidates (a handful of cells): every candidate of the cell, from the masks
                float best0 = 1.0e10f, second0 = 1.0e10f, best1 = 1.0e10f, second1 = 1.0e10f;
                // (long cells: the candidates of the two sub-cells of this step, not of the whole cell)
                auto cand_word = [&](uint32_t w) {
                    return long_cell ? uniform_u64(smask[s0 * 4u + w] | (s1 < 8u ? smask[s1 * 4u + w] : 0ull)) : uniform_u64(cmask[w]);
                };
                for (uint32_t w = 0; w < (k + 63u) / 64u; ++w) {
                    for (unsigned long long m = cand_word(w); m; m &= m - 1ull) {
                        const uint32_t j = w * 64u + (uint32_t)__builtin_ctzll(m);
                        const float4 c = s_cent[j];
                        const float d0 = cie94_key(pt0, c.x, c.y, c.z, c.w), d1 = cie94_key(pt1, c.x, c.y, c.z, c.w);
                        second0 = __builtin_amdgcn_fmed3f(d0, best0, second0);
                        second1 = __builtin_amdgcn_fmed3f(d1, best1, second1);
                        if (d0 < best0) { best0 = d0; ix0 = j; }
                        if (d1 < best1) { best1 = d1; ix1 = j; }
                    }
                }
                const float thr0 = tie_threshold(best0), thr1 = tie_threshold(best1);
                const bool near0 = second0 <= thr0, near1 = second1 <= thr1;
                if (__ballot(near0 || near1)) {
                    float lb0 = 100000.0f, lb1 = 100000.0f;
                    uint32_t li0 = 0u, li1 = 0u;
                    for (uint32_t w = 0; w < (k + 63u) / 64u; ++w) {
                        for (unsigned long long m = cand_word(w); m; m &= m - 1ull) {
                            const uint32_t j = w * 64u + (uint32_t)__builtin_ctzll(m);
                            const float4 c = s_cent[j];
                            if (near0 && cie94_key(pt0, c.x, c.y, c.z, c.w) <= thr0) {
                                const float d = cie94_c(v0.x, v0.y, v0.z, v0.w, c.x, c.y, c.z, c.w);
                                if (d < lb0) { lb0 = d; li0 = j; }
                            }
                            if (near1 && cie94_key(pt1, c.x, c.y, c.z, c.w) <= thr1) {
                                const float d = cie94_c(v1.x, v1.y, v1.z, v1.w, c.x, c.y, c.z, c.w);
                                if (d < lb1) { lb1 = d; li1 = j; }
                            }
                        }
                    }
                    ix0 = near0 ? li0 : ix0;
                    ix1 = near1 ? li1 : ix1;
                }
            }
            // what a scanned sub-cell leaves behind: labels (staged in LDS, stored at the end of the cell), sums, summary
            auto finish = [&](uint32_t s, uint32_t ix, uint32_t cnt, float vL, float va, float vb) {
                s_lbl[s * 64u + lane] = (LabelT)ix;
                label_set |= 1u << s;
                const bool counts = cnt != 0u;
                // sums: the sub-cell's total (sub-cell table) goes to a reference label R; a colour with another
                // label moves its own contribution from R to that label.  Half of the scanned sub-cells turn out
                // uniform: no per-colour arithmetic and four LDS adds instead of 256.
                const unsigned long long occm = __ballot(counts);
                uint32_t st = kSubEmpty;
                if (occm) {
                    const uint32_t X0 = lane_value(ix, (uint32_t)__builtin_ctzll(occm));
                    const unsigned long long other = __ballot(counts && ix != X0);
                    uint32_t R = X0;
                    if (other) {
                        const uint32_t X1 = lane_value(ix, (uint32_t)__builtin_ctzll(other));
                        if (__builtin_popcountll(__ballot(counts && ix == X1)) > __builtin_popcountll(occm & ~other)) R = X1;
                    }
                    st = other ? (uint32_t)kSubMixed : X0;
                    if (SUMS) {
                        if ((lane >> 2) == s) atomicAdd(bins + 4ull * R + (lane & 3u), (unsigned long long)sagg);
                        if (other && counts && ix != R) {
                            const long long m = (long long)cnt;
                            const long long c0 = m * (long long)lab_fix(vL), c1 = m * (long long)lab_fix(va), c2 = m * (long long)lab_fix(vb);
                            unsigned long long *to = my_bins + 4ull * ix, *from = my_bins + 4ull * R;
                            atomicAdd(to + 0, (unsigned long long)c0); atomicAdd(from + 0, (unsigned long long)(-c0));
                            atomicAdd(to + 1, (unsigned long long)c1); atomicAdd(from + 1, (unsigned long long)(-c1));
                            atomicAdd(to + 2, (unsigned long long)c2); atomicAdd(from + 2, (unsigned long long)(-c2));
                            atomicAdd(to + 3, (unsigned long long)m);  atomicAdd(from + 3, (unsigned long long)(-m));
                        }
                    }
                }
                if (sizeof(LabelT) != 1 && lane == 0u) sub[s] = (uint16_t)st;
            };
            finish(s0, ix0, cnt0, v0.x, v0.y, v0.z);
            if (s1 < 8u) finish(s1, ix1, cnt1, v1.x, v1.y, v1.z);
        };
        for (;;) {
            KMG_REQUEST_COLOURS(B);
            scan_pair(A_s0, A_s1, A_v0, A_v1, A_c0, A_c1);
            if (B_s0 >= 8u) break;
            KMG_REQUEST_COLOURS(A);
            scan_pair(B_s0, B_s1, B_v0, B_v1, B_c0, B_c1);
            if (A_s0 >= 8u) break;
        }
#undef KMG_REQUEST_COLOURS
        // the labels of the scanned sub-cells: lane l owns the colours 8 l .. 8 l + 7 (sub-cell l >> 3), one store
        __builtin_amdgcn_wave_barrier();
        if ((label_set >> (lane >> 3)) & 1u) {
            if (sizeof(LabelT) == 1)
                *reinterpret_cast<uint2 *>(cell_labels + lane * 8u) = *reinterpret_cast<const uint2 *>(s_lbl + lane * 8u);
            else
                *reinterpret_cast<uint4 *>(cell_labels + lane * 8u) = *reinterpret_cast<const uint4 *>(s_lbl + lane * 8u);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (SUMS) flush_bins(bins, k, repl, bin_stride, sums, n_rows);
}

// ------------------------------------------------------------------------------------------
// k_cube_pairs: pair entries (k <= 256) / cell summaries (k > 256) of the cells k_cube_stage left pending.
// occ_bits: one bit per colour of the bound image (NULL: every colour counts).
// ------------------------------------------------------------------------------------------
template <typename LabelT>
__global__ __launch_bounds__(kBlock) void k_cube_pairs(const uint32_t *__restrict__ work, int with_work,
                                                       const uint8_t *__restrict__ occ_bits,
                                                       const LabelT *__restrict__ colour_labels,
                                                       uint16_t *__restrict__ sub_table, uint32_t flags,
                                                       int64_t *__restrict__ sums, uint32_t k, CubeTail tail)
{
    // the tail of the pass (kmg_table.h CubeTail): the stage and scan launches have completed, so the sums are final
    if (tail.acc_out && blockIdx.x == gridDim.x - 1u) {
        __shared__ uint32_t s_count;
        for (uint32_t i = threadIdx.x; i < 4u * k; i += kBlock) tail.acc_out[i] = sums[i];
        if (tail.do_update) update_centroids(sums, k, tail.convergence, tail.cent, tail.n_converged, &s_count, kBlock);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 4u * k; i += kBlock) sums[i] = 0;      // ready for the next pass
    }
    if (flags & kCubeNoEntries) return;                             // the tail only (one workgroup): launch_cube_entries comes later
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (kBlock / 64);
    const uint32_t n_work = with_work ? __builtin_amdgcn_readfirstlane(work[0]) : kCells;
    const uint32_t *cells = with_work ? work + 1 : nullptr;
    // (Requesting the next cell's flag, labels and occupancy ahead was measured: no change, 21.5 us either way.)
    for (uint32_t wi = wave; wi < n_work; wi += n_waves) {
        const uint32_t cell = cells ? __builtin_amdgcn_readfirstlane(cells[wi]) : wi;
        if (sizeof(LabelT) == 1) {
            uint32_t *pair_entry_ptr = reinterpret_cast<uint32_t *>(sub_table + kSubCells + kCells) + cell;
            if (__builtin_amdgcn_readfirstlane(*pair_entry_ptr) != kPairPending) continue;
            const uint2 lv = *reinterpret_cast<const uint2 *>(colour_labels + (uint64_t)cell * kCellColours + lane * 8u);
            const uint32_t occ = occ_bits ? (uint32_t)occ_bits[(uint64_t)cell * 64u + lane] : 0xFFu;
            uint32_t idx[8];
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) { idx[q] = (lv.x >> (8u * q)) & 0xFFu; idx[4u + q] = (lv.y >> (8u * q)) & 0xFFu; }
            const uint32_t e = cell_pair_entry(idx, occ, lane);
            if (lane == 0u) *pair_entry_ptr = e;
        } else {
            // cell summary = merge of the eight sub-cell summaries (k_cube_stage wrote the decided / empty ones,
            // k_cube_scan the scanned ones)
            uint32_t st = lane < 8u ? (uint32_t)sub_table[cell * 8u + lane] : (uint32_t)kSubEmpty;
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) st = merge_state(st, (uint32_t)__shfl_xor(st, off, 64));
            if (lane == 0u) sub_table[kSubCells + cell] = (uint16_t)st;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_cube_small (k <= kSmallMaxK = 32): the whole cube pass of a SMALL centroid table in one launch.
//
// With few centroids the three launches above are chains of per-cell round trips with little work in them (BASELINE
// config 2, 4096^2, k = 16: stage 23.6 + scan 36.5 + pairs 17.4 us, 75 K scanned sub-cells of which 25 K are really mixed).
// Here a workgroup of 16 waves owns 128 cells from start to finish and nothing but the results leaves it:
//   1. stage -- one SUB-CELL per lane (a wave: 8 cells x 8 sub-cells): interval bounds of every centroid over the sub-cell's
//      static bounds (key_range, the same monotone evaluation), U = min hi, candidates = {lo_j <= U (1 + slack)} as a k-bit
//      word; then the DOMINANCE test below removes the candidates another candidate beats on every colour of the sub-cell.
//      One candidate left -> the sub-cell belongs to it (sums from the per-sub-cell table; a cell whose occupied sub-cells
//      all go to one centroid takes the per-cell sums and its pair entry at once, no per-colour traffic).
//   2. scan -- the undecided sub-cells of the workgroup's cells as ONE list in LDS, dealt out to the waves two per step,
//      one colour per lane, the next pair's colours requested while the current one is scanned.
//   3. entries -- the pair entry of every cell with more than one label, from the 512 labels the workgroup holds in LDS
//      (dealt out likewise); the labels leave as one 8-byte store per lane.
// The cells of a wave are dealt out with stride n_work / 8 over the work list and the scan list is shared by 16 waves, so
// that the boundary-heavy regions of the cube are spread evenly (one wave with its own 8 cells and its own list: the
// slowest of the 4096 waves meets 64 undecided sub-cells, the average one 20 -- measured 73 us for the pass).
// Candidate positions are the centroid indices themselves (5 bits in the key).  Sums go to kSmallRepl copies of the LDS
// bins (lane & 15): with 16 clusters a single copy has 64 lanes on 16 addresses.
// The pass's tail (sums hand-over, centroid update) stays with k_cube_pairs' tail workgroup, launched alone: finding the last
// workgroup inside the launch needs a device-scope fence per workgroup, which on this part writes the XCD's L2 back -- measured
// 63 -> 148 us.
//
// Dominance test (sub_affine != NULL).  With the per-colour terms of kmg_math.h as REAL numbers,
//   K_j = (L - Lj)^2 + wC (C - Cj)^2 + wH max(h_j, 0),   h_j = (a - aj)^2 + (b - bj)^2 - (C - Cj)^2,
// and h_j - h_i = (Nj - Ni) - 2 a (aj - ai) - 2 b (bj - bi) + 2 C (Cj - Ci), N = a^2 + b^2 - C^2 of the centroid, so that
//   K_j - K_i >= c0 + sum_m w_m F_m(colour) - wH delta,   F = (L, wC, wC C, wH a, wH b, wH C, wH),
//   c0 = Lj^2 - Li^2, w = (-2 (Lj - Li), Cj^2 - Ci^2, -2 (Cj - Ci), -2 (aj - ai), -2 (bj - bi), 2 (Cj - Ci), Nj - Ni)
// (max(h_j, 0) >= h_j; max(h_i, 0) <= h_i + delta, delta = 2^-20 (C + Ci)^2 >= how far the rounding of the two chromas can
// push h_i below zero: |C - Chat| <= 2.1 u Chat and the reverse triangle inequality).  The static table holds, per sub-cell
// and feature, an affine model alpha + beta . (x - 1.5, y - 1.5, z - 1.5) over the 4 x 4 x 4 colours and the exact range
// [Rmin, Rmax] of its residual (k_sub_affine), hence for every colour of the sub-cell
//   K_j - K_i >= c0 + sum_m w_m alpha_m - 1.5 sum_d |sum_m w_m beta_md| + sum_m min(w_m Rmin_m, w_m Rmax_m) - wH1 delta.
// Evaluated in binary32 with its rounding error charged explicitly (`dominated`).  A candidate j is dropped when this
// lower bound is >= 2^-11 hi_i for the candidate i with the smallest upper bound hi_i: then K_j >= K_i (1 + 2^-11 (1 - 63u))
// for every colour, and with |key - K| <= 560u K for the scan's keys (kmg_math.h) key_j > key_i (1 + 2^-12) -- j is neither
// the arg-min nor within the tie threshold of it, exactly the property the interval test guarantees for what IT removes.
// Skipped when a centroid lies outside |L|, |a|, |b| <= 1024 (the error budget above assumes colour-like magnitudes).
// Honoured flags: bit 0 (labels of uniform cells too), kCubeNoEntries.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kSmallMaxK = kCubeSmallMaxK;
constexpr uint32_t kSmallRepl = 16;
constexpr int kSmallBlock = 512;
constexpr uint32_t kSmallWaves = kSmallBlock / 64;
constexpr uint32_t kSmallCells = kSmallWaves * 8u;               // cells of a workgroup
constexpr uint32_t kSmallEmit = 0x100u;                           // k_cube_small: stage only, records for k_cube_scan / k_cube_pairs
constexpr uint32_t kSmallTests = 2048;                           // dominance tests a workgroup lists per round

// the 7 features of one colour (exact products of binary32 values in binary64)
__device__ __forceinline__ void affine_features(const float4 v, double F[7])
{
    const PixelTerms p = pixel_terms_c(v.x, v.y, v.z, v.w);
    F[0] = (double)p.L; F[1] = (double)p.wC; F[2] = (double)p.wC * (double)p.C;
    F[3] = (double)p.wH * (double)p.a; F[4] = (double)p.wH * (double)p.b; F[5] = (double)p.wH * (double)p.C; F[6] = (double)p.wH;
}

// once per processor: the affine models of the dominance test, one thread per sub-cell
__global__ __launch_bounds__(kBlock) void k_sub_affine(const float4 *__restrict__ lab_table,
                                                       const CellBounds *__restrict__ sub_bounds, float *__restrict__ affine)
{
    const uint32_t sc = blockIdx.x * kBlock + threadIdx.x;
    if (sc >= kSubCells) return;
    const float4 *col = lab_table + (uint64_t)sc * 64u;
    double S0[7], Sx[7], Sy[7], Sz[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) S0[m] = Sx[m] = Sy[m] = Sz[m] = 0.0;
    for (uint32_t q = 0; q < 64u; ++q) {
        double F[7];
        affine_features(col[q], F);
        const double dx = (double)((q >> 4) & 3u) - 1.5, dy = (double)((q >> 2) & 3u) - 1.5, dz = (double)(q & 3u) - 1.5;
#pragma unroll
        for (int m = 0; m < 7; ++m) { S0[m] += F[m]; Sx[m] += F[m] * dx; Sy[m] += F[m] * dy; Sz[m] += F[m] * dz; }
    }
    float al[7], bx[7], by[7], bz[7];
    double rmin[7], rmax[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        // least squares on the regular grid: sum (x - 1.5)^2 over the 64 colours = 80; then to binary16, the precision it is stored in
        al[m] = __half2float(__float2half_rn((float)(S0[m] / 64.0))); bx[m] = __half2float(__float2half_rn((float)(Sx[m] / 80.0)));
        by[m] = __half2float(__float2half_rn((float)(Sy[m] / 80.0))); bz[m] = __half2float(__float2half_rn((float)(Sz[m] / 80.0)));
        rmin[m] = 1.0e300; rmax[m] = -1.0e300;
    }
    // residuals against the ROUNDED model (the one the test uses)
    for (uint32_t q = 0; q < 64u; ++q) {
        double F[7];
        affine_features(col[q], F);
        const double dx = (double)((q >> 4) & 3u) - 1.5, dy = (double)((q >> 2) & 3u) - 1.5, dz = (double)(q & 3u) - 1.5;
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const double r = F[m] - ((double)al[m] + ((double)bx[m] * dx + ((double)by[m] * dy + (double)bz[m] * dz)));
            rmin[m] = fmin(rmin[m], r); rmax[m] = fmax(rmax[m], r);
        }
    }
    uint32_t *out = reinterpret_cast<uint32_t *>(affine) + (uint64_t)sc * kAffineFloats;
    unsigned short h[42];
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        // outward: the binary64 evaluation above is off by < 1e-15 of the terms; then directed roundings to binary32 and on to binary16
        const double pad = 1.0e-12 * (fabs((double)al[m]) + 1.5 * (fabs((double)bx[m]) + fabs((double)by[m]) + fabs((double)bz[m])) + 1.0);
        const double lo = rmin[m] - pad, hi = rmax[m] + pad;
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = nextafterf(flo, -3.0e38f);
        if ((double)fhi < hi) fhi = nextafterf(fhi, 3.0e38f);
        h[6 * m + 0] = __half_as_ushort(__float2half_rn(al[m])); h[6 * m + 1] = __half_as_ushort(__float2half_rn(bx[m]));
        h[6 * m + 2] = __half_as_ushort(__float2half_rn(by[m])); h[6 * m + 3] = __half_as_ushort(__float2half_rn(bz[m]));
        h[6 * m + 4] = __half_as_ushort(__float2half_rd(flo)); h[6 * m + 5] = __half_as_ushort(__float2half_ru(fhi));
    }
#pragma unroll
    for (int w = 0; w < 21; ++w) out[w] = (uint32_t)h[2 * w] | ((uint32_t)h[2 * w + 1] << 16);
    out[21] = float_to_bits(sub_bounds[sc].C1);                    // what the clamp correction of the test needs
    out[22] = float_to_bits(sub_bounds[sc].wH1);
    out[23] = 0u;
}

size_t sub_affine_bytes() { return sizeof(float) * (size_t)kAffineFloats * kSubCells; }

hipError_t launch_sub_affine(const float4 *lab_table, const CellBounds *sub_bounds, float *affine, hipStream_t st)
{
    hipLaunchKernelGGL(k_sub_affine, dim3(kSubCells / kBlock), dim3(kBlock), 0, st, lab_table, sub_bounds, affine);
    return hipGetLastError();
}

template <int KP, bool SUMS>
__global__ __launch_bounds__(kSmallBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_cube_small(const uint32_t *__restrict__ hist, const int64_t *__restrict__ agg,
                                                            const int64_t *__restrict__ sub_agg,
                                                            const uint8_t *__restrict__ occ_bits,
                                                            const uint32_t *__restrict__ work,
                                                            const CellBounds *__restrict__ bounds,
                                                            const CellBounds *__restrict__ sub_bounds,
                                                            const float *__restrict__ sub_affine,
                                                            const Centroid *__restrict__ cent, uint32_t k,
                                                            const float4 *__restrict__ lab_table,
                                                            uint64_t *__restrict__ masks_out,
                                                            CellWork *__restrict__ cell_work,
                                                            uint8_t *__restrict__ colour_labels,
                                                            uint16_t *__restrict__ sub_table,
                                                            int64_t *__restrict__ sums, uint32_t n_rows, uint32_t flags,
                                                            unsigned long long *__restrict__ stats)
{
    extern __shared__ float4 smem4[];
    float4 *s_cent = smem4;
    unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem4 + KP);
    const uint32_t bin_stride = 4u * k + 4u;                       // u64 per copy (+ 32 B: the next copy starts in other banks)
    const uint32_t n_bins = SUMS ? kSmallRepl * bin_stride : 0u;
    uint8_t *s_lbl = reinterpret_cast<uint8_t *>(bins + n_bins);                     // [cells][512 labels]
    uint32_t *s_mask = reinterpret_cast<uint32_t *>(s_lbl + kSmallCells * kCellColours);   // [sub-cells]: candidates
    float *s_U = reinterpret_cast<float *>(s_mask + kSmallBlock);                    // [sub-cells]: smallest upper bound
    uint32_t *s_cell = reinterpret_cast<uint32_t *>(s_U + kSmallBlock);              // [cells]: cell of slot (wave, ci)
    uint32_t *s_count = s_cell + kSmallCells;                                        // [0] entries, [1] pending cells, [2] far centroid, [3] tests
    uint16_t *s_ent = reinterpret_cast<uint16_t *>(s_count + 4);                     // [sub-cells]: sub-cells (thread ids) to scan
    uint16_t *s_pend = s_ent + kSmallBlock;                                          // [cells]: slots whose cell needs an entry
    uint16_t *s_test = s_pend + kSmallCells;                                         // [kSmallTests]: (sub-cell << 5) | candidate
    uint8_t *s_istar = reinterpret_cast<uint8_t *>(s_test + kSmallTests);            // [sub-cells]: the candidate with that bound
    // (requested first: the work list's length heads a chain of three dependent loads -- length, cell, bounds)
    const uint32_t n_work_v = SUMS ? work[opaque_vgpr_zero()] : kCells;
    for (uint32_t i = threadIdx.x; i < (uint32_t)KP; i += kSmallBlock) {
        float4 v = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);        // key ~ 1e36: never a candidate
        if (i < k) { const Centroid c = cent[i]; v = make_float4(c.L, c.a, c.b, c.C); }
        s_cent[i] = v;
    }
    for (uint32_t i = threadIdx.x; i < n_bins; i += kSmallBlock) bins[i] = 0ull;
    if (threadIdx.x < 4u) s_count[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x < k) {
        const float4 c = s_cent[threadIdx.x];
        if (!(fabsf(c.x) <= 1024.0f && fabsf(c.y) <= 1024.0f && fabsf(c.z) <= 1024.0f)) s_count[2] = 1u;
    }
    __syncthreads();
    const bool dominance = sub_affine != nullptr && s_count[2] == 0u;

    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    unsigned long long *my_bins = bins + (uint64_t)(lane & (kSmallRepl - 1u)) * bin_stride;
    uint32_t *pair_entries = reinterpret_cast<uint32_t *>(sub_table + kSubCells + kCells);
    const uint32_t ci = lane >> 3, sub = lane & 7u, slot = wv * 8u + ci;
    unsigned long long st_single = 0, st_multi = 0, st_decided = 0, st_scanned = 0, st_cands = 0;

    const uint32_t n_work = __builtin_amdgcn_readfirstlane(n_work_v);
    const uint32_t tasks = (n_work + 7u) >> 3;                     // wave tasks: task t = cells t, t + tasks, ... of the work list
    for (uint32_t base = blockIdx.x * kSmallWaves; base < tasks; base += gridDim.x * kSmallWaves) {
        // ---- 1a. stage: lane (ci, sub) = sub-cell `sub` of the wave's cell ci; candidates from interval bounds ----
        const uint32_t task = base + wv;
        const uint32_t pos = task + ci * tasks;
        const bool valid = task < tasks && pos < n_work;
        // (every cell occupied: the work list is the identity -- one round trip less)
        const uint32_t cell = valid ? ((SUMS && n_work != kCells) ? work[1u + pos] : pos) : 0u;
        const uint32_t sc = cell * 8u + sub;
        const float4 *cbp = reinterpret_cast<const float4 *>(bounds + cell);
        const float4 cb0 = cbp[0], cb1 = cbp[1], cb2 = cbp[2];
        const float4 *sbp = reinterpret_cast<const float4 *>(sub_bounds + sc);
        const float4 sb0 = sbp[0], sb1 = sbp[1], sb2 = sbp[2];
        // (the sums a decided sub-cell / a uniform cell will hand over are requested with the bounds, not after the decision)
        longlong2 s01 = {0, 0}, s23 = {0, 1};
        long long cell_sum = 0;
        if (SUMS) {
            const longlong2 *sp = reinterpret_cast<const longlong2 *>(sub_agg + (uint64_t)sc * 4u);
            s01 = sp[0]; s23 = sp[1];
            cell_sum = agg[4ull * cell + (sub & 3u)];
        }
        const bool occupied = valid && s23.y != 0;
        uint32_t mask = 0u;
        {
            // the cell's candidates first, the eight lanes of the cell sharing the centroids (two sweeps each time: upper
            // bounds, then lower bounds against the threshold -- each keeps half of key_range)
            CellBounds cb;
            cb.L0 = cb0.x; cb.L1 = cb0.y; cb.a0 = cb0.z; cb.a1 = cb0.w;
            cb.b0 = cb1.x; cb.b1 = cb1.y; cb.C0 = cb1.z; cb.C1 = cb1.w;
            cb.wC0 = cb2.x; cb.wC1 = cb2.y; cb.wH0 = cb2.z; cb.wH1 = cb2.w;
            constexpr int PER = KP / 8;
            float lo[PER], Uc = 3.0e38f;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const float4 c = s_cent[sub * PER + q];
                const KeyRange kr = key_range(cb, c.x, c.y, c.z, c.w);
                lo[q] = kr.lo;
                Uc = fminf(Uc, kr.hi);
            }
            const float Ucs = mask_threshold(group8_min(Uc));
            uint32_t bits = 0u;
#pragma unroll
            for (int q = 0; q < PER; ++q) bits |= (lo[q] <= Ucs ? 1u : 0u) << (sub * PER + q);
            const uint32_t cmask = group8_or(bits);
            // then the sub-cell's own bounds, over the cell's candidates
            CellBounds sb;
            sb.L0 = sb0.x; sb.L1 = sb0.y; sb.a0 = sb0.z; sb.a1 = sb0.w;
            sb.b0 = sb1.x; sb.b1 = sb1.y; sb.C0 = sb1.z; sb.C1 = sb1.w;
            sb.wC0 = sb2.x; sb.wC1 = sb2.y; sb.wH0 = sb2.z; sb.wH1 = sb2.w;
            float U = 3.0e38f;
            uint32_t istar = 0u;
            for (uint32_t m = cmask; m; m &= m - 1u) {
                const uint32_t j = (uint32_t)__builtin_ctz(m);
                const float4 c = s_cent[j];
                const float hi = key_range(sb, c.x, c.y, c.z, c.w).hi;
                istar = hi < U ? j : istar;
                U = fminf(U, hi);
            }
            const float Us = mask_threshold(U);
            for (uint32_t m = cmask; m; m &= m - 1u) {
                const uint32_t j = (uint32_t)__builtin_ctz(m);
                const float4 c = s_cent[j];
                mask |= (key_range(sb, c.x, c.y, c.z, c.w).lo <= Us ? 1u : 0u) << j;
            }
            s_mask[threadIdx.x] = mask;
            // 1b. the dominance tests of this sub-cell: (sub-cell, candidate) for every candidate but the one with the smallest
            // upper bound; a list that overflows drops tests, i.e. keeps candidates
            const uint32_t others = mask & ~(1u << istar);
            if (dominance && occupied && others) {
                s_U[threadIdx.x] = U;
                s_istar[threadIdx.x] = (uint8_t)istar;
                const uint32_t nt = (uint32_t)__builtin_popcount(others);
                uint32_t at = atomicAdd(&s_count[3], nt);
                for (uint32_t m = others; m && at < kSmallTests; m &= m - 1u, ++at)
                    s_test[at] = (uint16_t)((threadIdx.x << 5) | (uint32_t)__builtin_ctz(m));
            }
        }
        if (sub == 0u) s_cell[slot] = cell;
        __syncthreads();
        if (dominance) {
            const uint32_t n_tests = min(s_count[3], kSmallTests);
            for (uint32_t t = threadIdx.x; t < n_tests; t += kSmallBlock) {
                const uint32_t e = (uint32_t)s_test[t] >> 5, j = (uint32_t)s_test[t] & 31u;
                const uint32_t tsc = s_cell[e >> 3] * 8u + (e & 7u);
                HalfModel mdl;
                mdl.load(sub_affine, tsc);
                if (dominated(mdl, s_cent[j], s_cent[s_istar[e]], s_U[e])) atomicAnd(&s_mask[e], ~(1u << j));
            }
            __syncthreads();
            mask = s_mask[threadIdx.x];
        }

        // ---- 1c. what each sub-cell needs: empty / decided as a whole / scan its colours ----
        const uint32_t np = (uint32_t)__builtin_popcount(mask);
        const bool decided = occupied && np == 1u, scan = occupied && np > 1u;
        const uint32_t X = (uint32_t)__builtin_ctz(mask | 0x80000000u);
        const uint32_t um = group8_or(mask);                      // the cell's candidates
        const uint32_t any_scan = group8_or(scan ? 1u : 0u);
        const uint32_t xmin = group8_min_u32(decided ? X : 255u), xmax = group8_max_u32(decided ? X : 0u);
        const bool uniform = valid && !any_scan && xmin == xmax;   // every occupied sub-cell goes to centroid xmin
        if (valid && sub == 0u) masks_out[cell] = (uint64_t)um;
        if (SUMS) {
            if (uniform) {
                if (sub < 4u) atomicAdd(my_bins + 4ull * xmin + sub, (unsigned long long)cell_sum);
            } else if (decided) {
                unsigned long long *to = my_bins + 4ull * X;
                atomicAdd(to + 0, (unsigned long long)s01.x); atomicAdd(to + 1, (unsigned long long)s01.y);
                atomicAdd(to + 2, (unsigned long long)s23.x); atomicAdd(to + 3, (unsigned long long)s23.y);
            }
        }
        {
            const uint32_t x4 = (decided ? X : 0u) * 0x01010101u;
            const uint4 xv = make_uint4(x4, x4, x4, x4);
            if (uniform) {
                if (sub == 0u) pair_entries[cell] = pair_entry(xmin, xmin, 0u, 0u, 0u);
                if (flags & 1u) {
                    const uint32_t m4 = xmin * 0x01010101u;
                    uint4 *dst = reinterpret_cast<uint4 *>(colour_labels + (uint64_t)cell * kCellColours + sub * 64u);
                    dst[0] = dst[1] = dst[2] = dst[3] = make_uint4(m4, m4, m4, m4);
                }
            } else if (valid && !(flags & kSmallEmit)) {
                uint4 *dst = reinterpret_cast<uint4 *>(s_lbl + slot * kCellColours + sub * 64u);
                dst[0] = xv; dst[1] = xv; dst[2] = xv; dst[3] = xv;
            } else if (valid && decided) {                          // (kSmallEmit: the scan and entries are other launches)
                uint4 *dst = reinterpret_cast<uint4 *>(colour_labels + (uint64_t)cell * kCellColours + sub * 64u);
                dst[0] = xv; dst[1] = xv; dst[2] = xv; dst[3] = xv;
            }
        }
        if (flags & kSmallEmit) {
            // ---- stage only: the cell's record for k_cube_scan / k_cube_pairs (identity candidate list: position = centroid) ----
            if (valid) {
                CellWork *cw = cell_work + cell;
                const uint32_t scan8 = group8_or(scan ? 1u << sub : 0u);
                uint32_t lo_w[4], hi_w[4];
#pragma unroll
                for (uint32_t r = 0; r < 4u; ++r) {
                    const uint32_t byte = (mask >> (8u * r)) & 0xFFu;
                    lo_w[r] = group8_or(sub < 4u ? byte << (8u * sub) : 0u);
                    hi_w[r] = group8_or(sub >= 4u ? byte << (8u * (sub - 4u)) : 0u);
                }
                if (sub < 4u) {
                    const uint32_t l = sub == 0u ? lo_w[0] : (sub == 1u ? lo_w[1] : (sub == 2u ? lo_w[2] : lo_w[3]));
                    const uint32_t h = sub == 0u ? hi_w[0] : (sub == 1u ? hi_w[1] : (sub == 2u ? hi_w[2] : hi_w[3]));
                    cw->br[sub] = ((unsigned long long)h << 32) | l;
                }
                const uint32_t p0 = 4u * sub;
                *reinterpret_cast<uint2 *>(cw->list + p0) = make_uint2(p0 | (p0 + 1u) << 16, (p0 + 2u) | (p0 + 3u) << 16);
                if (sub == 0u) {
                    cw->npop = uniform ? 1u : k;
                    cw->scan_set = uniform ? 0u : (scan8 | 0x100u);
                    if (!uniform) pair_entries[cell] = kPairPending;
                }
            }
            if (stats) {
                const unsigned long long scan_b = __ballot(scan);
                st_single += (uint32_t)__builtin_popcountll(__ballot(valid && sub == 0u && __builtin_popcount(um) == 1));
                st_multi += (uint32_t)__builtin_popcountll(__ballot(valid && sub == 0u && __builtin_popcount(um) != 1));
                st_decided += (uint32_t)__builtin_popcountll(__ballot(decided && __builtin_popcount(um) != 1));
                st_scanned += (uint32_t)__builtin_popcountll(scan_b);
                st_cands += wave_add_u32(scan ? np : 0u);
            }
            __syncthreads();
            if (threadIdx.x == 3u) s_count[3] = 0u;
            __syncthreads();
            continue;
        }
        {
            // the workgroup's lists: undecided sub-cells (thread ids), cells that need an entry (slots)
            const unsigned long long scan_b = __ballot(scan);
            const unsigned long long pend_b = __ballot(valid && !uniform && sub == 0u);
            uint32_t e_base = 0u, p_base = 0u;
            if (lane == 0u) {
                if (scan_b) e_base = atomicAdd(&s_count[0], (uint32_t)__builtin_popcountll(scan_b));
                if (pend_b) p_base = atomicAdd(&s_count[1], (uint32_t)__builtin_popcountll(pend_b));
            }
            e_base = __builtin_amdgcn_readfirstlane(e_base);
            p_base = __builtin_amdgcn_readfirstlane(p_base);
            if (scan) s_ent[e_base + bits_below_lane(scan_b)] = (uint16_t)threadIdx.x;
            if (valid && !uniform && sub == 0u) s_pend[p_base + bits_below_lane(pend_b)] = (uint16_t)slot;
            if (stats) {
                const unsigned long long single_b = __ballot(valid && sub == 0u && __builtin_popcount(um) == 1);
                const unsigned long long multi_b = __ballot(valid && sub == 0u && __builtin_popcount(um) != 1);
                const unsigned long long dec_b = __ballot(decided && __builtin_popcount(um) != 1);
                st_single += (uint32_t)__builtin_popcountll(single_b);
                st_multi += (uint32_t)__builtin_popcountll(multi_b);
                st_decided += (uint32_t)__builtin_popcountll(dec_b);
                st_scanned += (uint32_t)__builtin_popcountll(scan_b);
                st_cands += wave_add_u32(scan ? np : 0u);
            }
        }
        __syncthreads();
        const uint32_t n_ent = s_count[0];
        const uint32_t n_pend = s_count[1];

        // ---- 2. scan: the undecided sub-cells of the workgroup, pair p to wave p % 8, one colour per lane ----
        // (its occupancy bytes for phase 3 are requested now: the scan hides their latency)
        uint32_t occ_early[2] = {0xFFu, 0xFFu};
        if (occ_bits && !(flags & kCubeNoEntries)) {
#pragma unroll
            for (uint32_t q = 0; q < 2u; ++q) {
                const uint32_t pi = wv + q * kSmallWaves;
                if (pi < n_pend) occ_early[q] = (uint32_t)occ_bits[(uint64_t)s_cell[s_pend[pi]] * 64u + lane];
            }
        }
        if (2u * wv < n_ent) {
            // three register sets, all requested before the first pair is scanned: a wave meets ~3 pairs, and with one pair
            // in flight every step waited a full memory round trip (2-3 us) for 0.3 us of work
            float4 A_v0, A_v1, B_v0, B_v1, C_v0, C_v1;
            uint32_t A_c0 = 1u, A_c1 = 1u, B_c0 = 1u, B_c1 = 1u, C_c0 = 1u, C_c1 = 1u;
            uint32_t A_e0 = 0u, A_e1 = 0u, B_e0 = 0u, B_e1 = 0u, C_e0 = 0u, C_e1 = 0u;
            long long A_g0 = 0, A_g1 = 0, B_g0 = 0, B_g1 = 0, C_g0 = 0, C_g1 = 0;
            bool A_h1 = false, B_h1 = false, C_h1 = false, A_ok = false, B_ok = false, C_ok = false;
            uint32_t next = 2u * wv;
            // (an exhausted slot re-requests entry 0, unused: a conditional request makes the compiler wait for the registers)
#define KMG_REQUEST_PAIR(X)                                                                                       \
            do {                                                                                                  \
                X##_ok = next < n_ent;                                                                            \
                X##_h1 = next + 1u < n_ent;                                                                       \
                X##_e0 = __builtin_amdgcn_readfirstlane((uint32_t)s_ent[next < n_ent ? next : 0u]);               \
                X##_e1 = __builtin_amdgcn_readfirstlane((uint32_t)s_ent[X##_h1 ? next + 1u : 0u]);                \
                next += 2u * kSmallWaves;                                                                         \
                const uint32_t cell0_ = __builtin_amdgcn_readfirstlane(s_cell[X##_e0 >> 3]);                      \
                const uint32_t cell1_ = __builtin_amdgcn_readfirstlane(s_cell[X##_e1 >> 3]);                      \
                const uint32_t c0_ = cell0_ * kCellColours + (X##_e0 & 7u) * 64u + lane;                          \
                const uint32_t c1_ = cell1_ * kCellColours + (X##_e1 & 7u) * 64u + lane;                          \
                X##_v0 = lab_table[c0_];                                                                          \
                X##_v1 = lab_table[c1_];                                                                          \
                if (SUMS) {                                                                                       \
                    X##_c0 = hist[c0_]; X##_c1 = hist[c1_];                                                       \
                    X##_g0 = sub_agg[((uint64_t)cell0_ * 8u + (X##_e0 & 7u)) * 4u + (lane & 3u)];                 \
                    X##_g1 = sub_agg[((uint64_t)cell1_ * 8u + (X##_e1 & 7u)) * 4u + (lane & 3u)];                 \
                }                                                                                                 \
            } while (0)
            // e = thread id of the sub-cell in the stage: slot e >> 3, sub-cell e & 7
            auto scan_pair = [&](const uint32_t e0, const uint32_t e1, const bool has1, const float4 v0, const float4 v1,
                                 const uint32_t cnt0, const uint32_t cnt1, const long long g0, const long long g1) {
                const PixelTerms pt0 = pixel_terms_fast(v0.x, v0.y, v0.z, v0.w), pt1 = pixel_terms_fast(v1.x, v1.y, v1.z, v1.w);
                uint32_t b0 = 0x7F7FFFFFu, r0 = 0x7F7FFFFFu, b1 = 0x7F7FFFFFu, r1 = 0x7F7FFFFFu;   // smallest / runner-up
                const uint32_t sm = __builtin_amdgcn_readfirstlane(s_mask[e0] | (has1 ? s_mask[e1] : 0u));
                // (the two colours of a lane as the halves of packed-f32 operands, as in k_cube_scan; position = centroid index)
                const f32x2 qL = {pt0.L, pt1.L}, qa = {pt0.a, pt1.a}, qb = {pt0.b, pt1.b}, qC = {pt0.C, pt1.C};
                const f32x2 qwC = {pt0.wC, pt1.wC}, qwH = {pt0.wH, pt1.wH};
                for (uint32_t m = sm; m; m &= m - 1u) {
                    const uint32_t pos = (uint32_t)__builtin_ctz(m);
                    const float4 c = s_cent[pos];
                    const f32x2 dL = qL - c.x, da = qa - c.y, db = qb - c.z, dC = qC - c.w;
                    const f32x2 dC2 = dC * dC;
                    const f32x2 t = __builtin_elementwise_fma(db, db, da * da);
                    f32x2 h = t - dC2;
                    h.x = fmaxf(h.x, 0.0f); h.y = fmaxf(h.y, 0.0f);
                    const f32x2 key = __builtin_elementwise_fma(h, qwH, __builtin_elementwise_fma(dC2, qwC, dL * dL));
                    uint32_t u0, u1;
                    asm("v_bfi_b32 %0, 31, %1, %2" : "=v"(u0) : "s"(pos), "v"(float_to_bits(key.x)));
                    asm("v_bfi_b32 %0, 31, %1, %2" : "=v"(u1) : "s"(pos), "v"(float_to_bits(key.y)));
                    r0 = umed3(u0, b0, r0); b0 = min(b0, u0);
                    r1 = umed3(u1, b1, r1); b1 = min(b1, u1);
                }
                uint32_t p0 = b0 & 31u, p1 = b1 & 31u;
                // near-tie repair (kmg_math.h): rare; decided by the literal distance, first minimum wins
                const float thr0 = tie_threshold(bits_to_float(b0 & ~31u)), thr1 = tie_threshold(bits_to_float(b1 & ~31u));
                const bool near0 = bits_to_float(r0 & ~31u) <= thr0, near1 = bits_to_float(r1 & ~31u) <= thr1;
                if (__ballot(near0 || near1)) {
                    float lb0 = 100000.0f, lb1 = 100000.0f;       // find_centroid.wgsl:29-30
                    uint32_t li0 = 0u, li1 = 0u;
                    for (uint32_t m = sm; m; m &= m - 1u) {
                        const uint32_t pos = (uint32_t)__builtin_ctz(m);
                        const float4 c = s_cent[pos];
                        if (near0 && cie94_key(pt0, c.x, c.y, c.z, c.w) <= thr0) {
                            const float d = cie94_c(v0.x, v0.y, v0.z, v0.w, c.x, c.y, c.z, c.w);
                            if (d < lb0) { lb0 = d; li0 = pos; }
                        }
                        if (near1 && cie94_key(pt1, c.x, c.y, c.z, c.w) <= thr1) {
                            const float d = cie94_c(v1.x, v1.y, v1.z, v1.w, c.x, c.y, c.z, c.w);
                            if (d < lb1) { lb1 = d; li1 = pos; }
                        }
                    }
                    p0 = near0 ? li0 : p0;
                    p1 = near1 ? li1 : p1;
                }
                // what a scanned sub-cell leaves behind: its labels (LDS) and sums -- the sub-cell's total goes to a reference
                // label R, a colour with another label moves its own contribution from R to that label
                auto finish = [&](uint32_t e, uint32_t ix, uint32_t cnt, long long g, float vL, float va, float vb) {
                    s_lbl[(e >> 3) * kCellColours + (e & 7u) * 64u + lane] = (uint8_t)ix;
                    if (!SUMS) return;
                    const bool counts = cnt != 0u;
                    const unsigned long long occm = __ballot(counts);
                    if (!occm) return;
                    const uint32_t X0 = lane_value(ix, (uint32_t)__builtin_ctzll(occm));
                    const unsigned long long other = __ballot(counts && ix != X0);
                    uint32_t R = X0;
                    if (other) {
                        const uint32_t X1 = lane_value(ix, (uint32_t)__builtin_ctzll(other));
                        if (__builtin_popcountll(__ballot(counts && ix == X1)) > __builtin_popcountll(occm & ~other)) R = X1;
                    }
                    if (lane < 4u) atomicAdd(my_bins + 4ull * R + lane, (unsigned long long)g);
                    if (other && counts && ix != R) {
                        const long long m = (long long)cnt;
                        const long long c0 = m * (long long)lab_fix(vL), c1 = m * (long long)lab_fix(va), c2 = m * (long long)lab_fix(vb);
                        unsigned long long *to = my_bins + 4ull * ix, *from = my_bins + 4ull * R;
                        atomicAdd(to + 0, (unsigned long long)c0); atomicAdd(from + 0, (unsigned long long)(-c0));
                        atomicAdd(to + 1, (unsigned long long)c1); atomicAdd(from + 1, (unsigned long long)(-c1));
                        atomicAdd(to + 2, (unsigned long long)c2); atomicAdd(from + 2, (unsigned long long)(-c2));
                        atomicAdd(to + 3, (unsigned long long)m);  atomicAdd(from + 3, (unsigned long long)(-m));
                    }
                };
                finish(e0, p0, cnt0, g0, v0.x, v0.y, v0.z);
                if (has1) finish(e1, p1, cnt1, g1, v1.x, v1.y, v1.z);
            };
            KMG_REQUEST_PAIR(A);
            KMG_REQUEST_PAIR(B);
            KMG_REQUEST_PAIR(C);
            for (;;) {
                scan_pair(A_e0, A_e1, A_h1, A_v0, A_v1, A_c0, A_c1, A_g0, A_g1);
                if (!B_ok) break;
                KMG_REQUEST_PAIR(A);
                scan_pair(B_e0, B_e1, B_h1, B_v0, B_v1, B_c0, B_c1, B_g0, B_g1);
                if (!C_ok) break;
                KMG_REQUEST_PAIR(B);
                scan_pair(C_e0, C_e1, C_h1, C_v0, C_v1, C_c0, C_c1, C_g0, C_g1);
                if (!A_ok) break;
                KMG_REQUEST_PAIR(C);
            }
#undef KMG_REQUEST_PAIR
        }
        __syncthreads();

        // ---- 3. entries: cells with more than one label (or scanned sub-cells): pair entry from the 512 labels in LDS ----
        for (uint32_t pi = wv; pi < n_pend; pi += kSmallWaves) {
            const uint32_t ps = __builtin_amdgcn_readfirstlane((uint32_t)s_pend[pi]);
            const uint32_t pcell = __builtin_amdgcn_readfirstlane(s_cell[ps]);
            const uint2 lv = *reinterpret_cast<const uint2 *>(s_lbl + ps * kCellColours + lane * 8u);
            uint32_t e = kPairPending;
            if (!(flags & kCubeNoEntries)) {
                const uint32_t round = (pi - wv) / kSmallWaves;
                const uint32_t occ = round == 0u ? occ_early[0] : (round == 1u ? occ_early[1] :
                                     (occ_bits ? (uint32_t)occ_bits[(uint64_t)pcell * 64u + lane] : 0xFFu));
                uint32_t idx[8];
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) { idx[q] = (lv.x >> (8u * q)) & 0xFFu; idx[4u + q] = (lv.y >> (8u * q)) & 0xFFu; }
                e = cell_pair_entry(idx, occ, lane);
            }
            if (lane == 0u) pair_entries[pcell] = e;
            *reinterpret_cast<uint2 *>(colour_labels + (uint64_t)pcell * kCellColours + lane * 8u) = lv;
        }
        __syncthreads();
        if (threadIdx.x < 2u) s_count[threadIdx.x] = 0u;
        if (threadIdx.x == 3u) s_count[3] = 0u;
        __syncthreads();
    }
    if (stats && lane == 0u) {
        atomicAdd(stats + 0, st_single); atomicAdd(stats + 1, st_multi); atomicAdd(stats + 2, st_decided);
        atomicAdd(stats + 3, st_scanned); atomicAdd(stats + 4, st_cands);
    }
    if (SUMS) {
        // (flush_bins for this block size)
        __syncthreads();
        unsigned long long *row = reinterpret_cast<unsigned long long *>(sums) + (uint64_t)(blockIdx.x % n_rows) * 4ull * k;
        for (uint32_t i = threadIdx.x; i < 4u * k; i += kSmallBlock) {
            unsigned long long v = 0ull;
            for (uint32_t r = 0; r < kSmallRepl; ++r) v += bins[(uint64_t)r * bin_stride + i];
            if (v) atomicAdd(row + i, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_cube_one (32 < k <= 256, images without hot cells; round 6): the whole cube pass in ONE launch.  Rounds 3-5 ran it as four
// (candidates 16.5 + sub-cell stage and dominance tests 27 + scan 35 + pair entries 16 us, ~3 us between them: 94 us by rocprofv3)
// with work records, items and balanced lists in memory between them; here nothing but the results leaves the workgroup: no
// records, no reservation atomics, one prologue / flush instead of four, no launch boundaries -- 76-79 us on the same box
// (profiles/r06_cube_one_ab.txt).  k_cube_small's shape with candidate lists; a workgroup of 16 waves (ONE per CU) owns 128
// cells, dealt pseudo-randomly over the work list (the cost of a cell follows its place in the cube):
//   1a. candidates   a wave takes its 8 cells one after the other (a lane's four centroids in registers, interval bounds over
//                    the cell, U, ballots); the list goes to LDS.  One candidate: the cell is decided.  More than
//                    kMaxListed: no per-sub-cell sets, every colour against the cell's mask (a handful of cells).
//   1b. sub-cells    a thread per sub-cell (its bounds and sums were requested BEFORE 1a: the candidates hide the round
//                    trip): two sweeps over the list, then the dominance tests of the WAVE's sub-cells as one compact LDS
//                    list, a test per lane (`dominated`; the 96-byte model from L2);
//   1c. decisions    sub-cells with one candidate take their sums and their 64 labels (LDS); cells whose occupied sub-cells
//                    agree take the per-cell sums and their pair entry; what is open becomes ITEMS in LDS -- two sub-cells of
//                    a cell and the union of their candidates (<= 12: as bytes; more: by list position);
//   -- the ONE workgroup barrier between prologue and flush: every wave's items are listed --
//   2.  scan         whoever is free takes the next item (an LDS counter), one colour per lane and sub-cell, three items'
//                    colours in flight: key | position, one integer min / med3 per visit, literal near-tie repair, sums by
//                    moving minority colours between copies of the LDS bins; labels into LDS;
//   3.  entries      a cell's items count down in LDS: the wave that scans the last one derives the pair entry from the 512
//                    labels (cell_pair_entry) and stores them, 8 bytes per lane.
// Waves never wait for each other inside a phase: with a barrier per phase (the first version) a workgroup paid the slowest of
// its waves five times -- 92 us; handing out items dynamically and counting cells down: 82 us; one workgroup of 16 waves per
// CU instead of two of 8 (one pool for 128 cells): 79 us; four copies of the bins instead of two: 76.5 us
// (tools/cube_one_phases.py stamps the phases; profiles/r06_cube_one_phases_*.txt).  What is left is the spread BETWEEN
// workgroups -- the slowest of 256 has 1.2 x the items of the average one.
// Same arithmetic as the launches it replaces, step for step: results are bit-identical.  The tail of the pass (sums hand-over,
// centroid update) rides on the label pass's last workgroup, or is k_cube_pairs' tail workgroup launched alone.
// stats: as k_cube_stage.  flags: bit 0, kCubeNoEntries.
// ------------------------------------------------------------------------------------------
constexpr int kOneBlock = 1024;                                  // 16 waves: ONE workgroup per CU
constexpr uint32_t kOneWaves = kOneBlock / 64, kOneCells = kOneWaves * 8u;   // a wave: 8 cells
constexpr uint32_t kOneItems = kOneCells * 4u;                   // at most four pairs of sub-cells per cell
constexpr uint32_t kOneTests = 384;                              // dominance tests a WAVE lists (~120 on the benchmark image; more are dropped = kept candidates)
constexpr uint32_t kOneRepl = 4;                                 // copies of the LDS bins

constexpr uint32_t kOneTaskSlots = (kCells / kOneCells) * kOneWaves; // 4096 (workgroup, wave) places of a pass
constexpr uint32_t kOneBalWords = 80;                              // CubeBalance: 2 x 32 (task, weight) entries + the workgroup's 16 tasks
size_t cube_balance_bytes() { return sizeof(uint16_t) * 4u * kOneTaskSlots; }

static size_t cube_one_lds_bytes(uint32_t k, bool with_sums)
{
    const size_t bins = with_sums ? sizeof(unsigned long long) * kOneRepl * (4ull * k + 4ull) : 0u;
    return sizeof(float4) * 256u + bins + (size_t)kOneCells * kCellColours + sizeof(uint4) * kOneItems + sizeof(uint16_t) * kOneWaves * kOneTests +
           sizeof(unsigned long long) * kOneCells * 4u + sizeof(uint32_t) * kOneBlock * 2u + (size_t)kOneCells * 64u +
           sizeof(uint32_t) * kOneCells * 3u + sizeof(uint32_t) * (8u + kOneWaves + kOneBalWords) + sizeof(uint16_t) * kOneCells * kMaxListed;
}

#ifdef KMG_TOOLS
// tools build: phase stamps of k_cube_one (s_memrealtime, 100 MHz), [workgroup][8]: start, 1a done (wave 0), 1b swept, tests done, 1c done,
// scan done (wave 0), scan done (all), entries done (wave 0); [8..10] items, pending cells, tests of the workgroup
__device__ unsigned long long g_one_stamps[(kCells / kOneCells) * 12u];
#define KMG_STAMP(i) do { if (threadIdx.x == 0u) g_one_stamps[blockIdx.x * 12u + (i)] = wall_clock64(); } while (0)
#define KMG_STAMP_VALUE(i, v) do { if (threadIdx.x == 0u) g_one_stamps[blockIdx.x * 12u + (i)] = (v); } while (0)
#else
#define KMG_STAMP(i) do { } while (0)
#define KMG_STAMP_VALUE(i, v) do { } while (0)
#endif

template <bool SUMS>
__global__ __launch_bounds__(kOneBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_cube_one(
    const uint32_t *__restrict__ hist, const int64_t *__restrict__ agg, const int64_t *__restrict__ sub_agg,
    const uint8_t *__restrict__ occ_bits, const uint32_t *__restrict__ work, const CellBounds *__restrict__ bounds,
    const CellBounds *__restrict__ sub_bounds, const float *__restrict__ sub_affine, const Centroid *__restrict__ cent, uint32_t k,
    const float4 *__restrict__ lab_table, uint64_t *__restrict__ masks_out, uint8_t *__restrict__ colour_labels,
    uint16_t *__restrict__ sub_table, int64_t *__restrict__ sums, uint32_t n_rows, uint32_t flags,
    unsigned long long *__restrict__ stats, uint16_t *__restrict__ bal_state, uint32_t bal_pass)
{
    extern __shared__ float4 smem4[];
    float4 *s_cent = smem4;                                                            // [256]
    unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem4 + 256);
    const uint32_t bin_stride = 4u * k + 4u;
    const uint32_t n_bins = SUMS ? kOneRepl * bin_stride : 0u;
    uint8_t *s_lbl = reinterpret_cast<uint8_t *>(bins + n_bins);                       // [cells][512]
    uint4 *s_item = reinterpret_cast<uint4 *>(s_lbl + kOneCells * kCellColours);        // [kOneItems]
    uint16_t *s_test = reinterpret_cast<uint16_t *>(s_item + kOneItems);               // [waves][kOneTests]: (lane << 5) | position
    unsigned long long *s_cmask = reinterpret_cast<unsigned long long *>(s_test + kOneWaves * kOneTests);   // [cells][4]
    uint32_t *s_mask = reinterpret_cast<uint32_t *>(s_cmask + kOneCells * 4u);          // [sub-cells]: candidates (list positions)
    uint32_t *s_ref = s_mask + kOneBlock;                                              // [sub-cells]: upper bound | reference position
    uint8_t *s_occ = reinterpret_cast<uint8_t *>(s_ref + kOneBlock);                   // [cells][64]: occupancy bits of the cell's colours
    uint32_t *s_cell = reinterpret_cast<uint32_t *>(s_occ + kOneCells * 64u);           // [cells]
    uint32_t *s_npop = s_cell + kOneCells;                                             // [cells]: candidates | first << 16
    uint32_t *s_left = s_npop + kOneCells;                                             // [cells]: items of the cell still to scan
    uint32_t *s_count = s_left + kOneCells;                                            // [0] items, [2] far centroid, [4] next item, [8 + w] tests of wave w
    uint32_t *s_bal = s_count + 8 + kOneWaves;                                          // [64] entries of the exchange, [16] this workgroup's tasks
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_bal + kOneBalWords);                 // [cells][kMaxListed]: centroid by position

    const uint32_t vz = opaque_vgpr_zero();
    const uint32_t n_work_v = SUMS ? work[vz] : kCells;
    if (threadIdx.x < 256u) {
        float4 v = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);           // key ~ 1e36: never a candidate
        if (threadIdx.x < k) { const Centroid c = cent[threadIdx.x]; v = make_float4(c.L, c.a, c.b, c.C); }
        s_cent[threadIdx.x] = v;
    }
    if (threadIdx.x < 8u + kOneWaves) s_count[threadIdx.x] = 0u;
    if (SUMS) for (uint32_t i = threadIdx.x; i < n_bins; i += kOneBlock) bins[i] = 0ull;
    if (bal_state && (threadIdx.x >> 6) == kOneWaves - 1u) {
        // CubeBalance (kmg_table.h): the tasks of this workgroup in this pass.  A pass costs what its slowest workgroup's scan costs, a
        // workgroup's scan what its items cost (correlation 0.95), and a task's items change little from one Lloyd iteration to the
        // next.  So every pass each workgroup meets one partner (bit (pass - 1) mod 8 of its index: a hypercube, all of it in eight
        // passes): both read both rows of the PREVIOUS pass -- task ids and what their items cost -- and deal the 32 tasks out
        // again, heaviest first, each to the lighter of the two that still has a free wave.  Both compute the same deal from the same
        // data and each writes its own row of THIS pass's state: no task is lost or doubled, no workgroup waits for another.
        // (Four workgroups per deal -- two bits a pass -- even the items out twice as fast and cost the pass 8 us: the deal is
        // serial work in front of the first barrier.)
        const uint32_t bl = threadIdx.x & 63u;
        const uint32_t nw = __builtin_amdgcn_readfirstlane(n_work_v);
        const uint32_t bcpw = min(8u, max(1u, (nw + gridDim.x * kOneWaves - 1u) / (gridDim.x * kOneWaves)));
        const uint32_t btasks = (nw + bcpw - 1u) / bcpw, bused = (btasks + gridDim.x - 1u) / gridDim.x;
        uint16_t *cur = bal_state + (bal_pass & 1u) * 2u * kOneTaskSlots;
        const uint16_t *prev = bal_state + ((bal_pass + 1u) & 1u) * 2u * kOneTaskSlots;
        uint32_t id = 0xFFFFu;
        if (bal_pass == 0u) {
            if (bl < bused && blockIdx.x * bused + bl < btasks) id = blockIdx.x * bused + bl;
        } else {
            const uint32_t partner = blockIdx.x ^ (1u << ((bal_pass - 1u) & 7u));
            const uint32_t lo_wg = min(blockIdx.x, partner), hi_wg = max(blockIdx.x, partner);
            const uint32_t src = ((bl & 16u) ? hi_wg : lo_wg) * kOneWaves + (bl & 15u);
            uint32_t e_id = 0xFFFFu, e_w = 0u;
            if (bl < 32u) { e_id = prev[src]; e_w = prev[kOneTaskSlots + src]; }
            if (e_id >= btasks) { e_id = 0xFFFFu; e_w = 0u; }
            if (bl < 32u) s_bal[bl] = (e_w << 16) | e_id;
            __builtin_amdgcn_wave_barrier();
            uint32_t rank = 0;
            for (uint32_t j = 0; j < 32u; ++j) {
                const uint32_t ow = s_bal[j] >> 16;
                rank += (ow > e_w || (ow == e_w && j < bl)) ? 1u : 0u;
            }
            if (bl < 32u) s_bal[32u + rank] = (e_w << 16) | e_id;
            __builtin_amdgcn_wave_barrier();
            const uint32_t sorted = bl < 32u ? s_bal[32u + bl] : 0u;
            const bool i_am_hi = blockIdx.x == hi_wg;
            uint32_t t0 = 0, t1 = 0, c0 = 0, c1 = 0;
#pragma unroll
            for (uint32_t r = 0; r < 32u; ++r) {
                const uint32_t e = lane_value(sorted, r);
                const bool to_hi = c0 >= kOneWaves || (c1 < kOneWaves && t1 < t0);
                if (to_hi) { if (i_am_hi && bl == c1) id = e & 0xFFFFu; ++c1; t1 += e >> 16; }
                else       { if (!i_am_hi && bl == c0) id = e & 0xFFFFu; ++c0; t0 += e >> 16; }
            }
        }
        if (bl < kOneWaves) { cur[blockIdx.x * kOneWaves + bl] = (uint16_t)id; s_bal[64u + bl] = id; }
    }
    __syncthreads();
    if (threadIdx.x < k) {
        const float4 c = s_cent[threadIdx.x];
        if (!(fabsf(c.x) <= 1024.0f && fabsf(c.y) <= 1024.0f && fabsf(c.z) <= 1024.0f)) s_count[2] = 1u;
    }
    __syncthreads();
    const bool dominance = sub_affine != nullptr && s_count[2] == 0u;

    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t ci = lane >> 3, sub = lane & 7u, slot = wv * 8u + ci;
    unsigned long long *my_bins = bins + (uint64_t)(lane & (kOneRepl - 1u)) * bin_stride;
    uint32_t *pair_entries = reinterpret_cast<uint32_t *>(sub_table + kSubCells + kCells);
    const uint32_t words = (k + 63u) / 64u;
    unsigned long long st_single = 0, st_multi = 0, st_unlisted = 0, st_decided = 0, st_scanned = 0, st_cands = 0, st_removed = 0, st_one = 0;
    float4 c4[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) c4[w] = s_cent[w * 64 + lane];      // this lane's four centroids: the same for every cell

    const uint32_t n_work = __builtin_amdgcn_readfirstlane(n_work_v);
    // One batch per workgroup: the grid is kCells / kOneCells whatever the list's length.  A wave takes `cpw` cells -- 8 of a full
    // list; a SHORT list (one rank's share of the cube in a cell-sharded loop, kmg_lloyd_set_cell_share: 4096 cells at 8 ranks) is
    // spread over ALL waves of all workgroups, fewer cells each, so that phase 1a (a wave's cells one after the other) shrinks
    // with the share instead of leaving most waves idle behind two busy ones.
    const uint32_t cpw = min(8u, max(1u, (n_work + gridDim.x * kOneWaves - 1u) / (gridDim.x * kOneWaves)));
    const uint32_t tasks = (n_work + cpw - 1u) / cpw;              // wave tasks: task t = cells t, t + tasks, ... (cpw of them) of the work list
    const uint32_t waves_used = (tasks + gridDim.x - 1u) / gridDim.x;       // <= kOneWaves: the grid holds kCells / 8 tasks
    if (!bal_state && blockIdx.x * waves_used >= tasks) return;
    {
        KMG_STAMP(0);
        // (wave task t -> place (t P) mod tasks of the work list, P a prime that does not divide tasks: the cells of a workgroup
        // are spread over the whole cube instead of lying along a few lines of it -- the cost of a cell follows its position)
        uint32_t task = wv < waves_used ? blockIdx.x * waves_used + wv : tasks;
        if (bal_state) task = min(s_bal[64u + wv], tasks);          // (0xFFFF: a wave without a task)
        const uint32_t prime = tasks % 2053u ? 2053u : 1031u;
        const uint32_t pos = (uint32_t)(((uint64_t)task * prime) % tasks) + ci * tasks;
        const bool valid = task < tasks && ci < cpw && pos < n_work;
        const uint32_t cell = valid ? ((SUMS && n_work != kCells) ? work[1u + pos] : pos) : 0u;
        const uint32_t sc = cell * 8u + sub;
        // ---- requests: the bounds of the wave's eight cells (lane 16 c + i: float i of cell c / c + 4), then everything 1b needs ----
        const uint32_t cell_a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane >> 4) * 8u) << 2), (int)cell);
        const uint32_t cell_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((4u + (lane >> 4)) * 8u) << 2), (int)cell);
        const float cbv_a = reinterpret_cast<const float *>(bounds + cell_a)[lane & 15u];
        const float cbv_b = reinterpret_cast<const float *>(bounds + cell_b)[lane & 15u];
        const float4 *sbp = reinterpret_cast<const float4 *>(sub_bounds + sc);
        const float4 sb0 = sbp[0], sb1 = sbp[1], sb2 = sbp[2];
        longlong2 g01 = {0, 0}, g23 = {0, 1};
        long long cell_sum = 0;
        if (SUMS) {
            const longlong2 *sp = reinterpret_cast<const longlong2 *>(sub_agg + (uint64_t)sc * 4u);
            g01 = sp[0]; g23 = sp[1];
            cell_sum = agg[4ull * cell + (sub & 3u)];
        }
        // (the occupancy bits of the thread's 8 x 8 colours: phase 3 reads them from LDS)
        uint2 occ8 = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        if (occ_bits) occ8 = *reinterpret_cast<const uint2 *>(occ_bits + (uint64_t)cell * 64u + sub * 8u);
        if (sub == 0u) s_cell[slot] = cell;

        // ---- 1a. the candidates of the wave's cells, one cell at a time ----
        // The cell's twelve bounds reach every lane as VECTOR registers, through LDS (three broadcast ds_read_b128 from the wave's
        // own label block, which nothing writes before this wave's 1c): as twelve v_readlane they cost 8 issue cycles each and
        // then sat in scalar registers -- a scalar operand costs a vector instruction 4.1 issue cycles instead of 2.3-2.6
        // (tools/valu_rate.hip), and most instructions of the interval evaluations had one.  73.5 -> 71.7 us.
        float *s_cb = reinterpret_cast<float *>(s_lbl + (size_t)wv * 8u * kCellColours);
        s_cb[lane] = cbv_a;
        s_cb[64u + lane] = cbv_b;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c = 0; c < cpw; ++c) {
            const uint32_t ccell = lane_value(cell, c * 8u);
            const bool cvalid = lane_value((uint32_t)valid, c * 8u) != 0u;
            const float4 *cq = reinterpret_cast<const float4 *>(s_cb + c * 16u);
            const float4 q0 = cq[0], q1 = cq[1], q2 = cq[2];
            CellBounds cb;
            cb.L0 = q0.x; cb.L1 = q0.y; cb.a0 = q0.z; cb.a1 = q0.w;
            cb.b0 = q1.x; cb.b1 = q1.y; cb.C0 = q1.z; cb.C1 = q1.w;
            cb.wC0 = q2.x; cb.wC1 = q2.y; cb.wH0 = q2.z; cb.wH1 = q2.w;
            float lo[4], U = 3.0e38f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const KeyRange r = key_range(cb, c4[w].x, c4[w].y, c4[w].z, c4[w].w);
                lo[w] = r.lo;
                U = fminf(U, r.hi);                                 // padding entries: hi ~ 1e36
            }
            const float Us = mask_threshold(wave_min(U));
            unsigned long long mw[4];
            uint32_t npop = 0, first = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                mw[w] = __ballot(w * 64u + lane < k && lo[w] <= Us);
                if (npop == 0u && mw[w]) first = w * 64u + (uint32_t)__builtin_ctzll(mw[w]);
                npop += (uint32_t)__builtin_popcountll(mw[w]);
            }
            const uint32_t cslot = wv * 8u + c;
            if (cvalid && lane < words) masks_out[(uint64_t)ccell * words + lane] = lane == 0u ? mw[0] : (lane == 1u ? mw[1] : (lane == 2u ? mw[2] : mw[3]));
            if (lane == 0u) s_npop[cslot] = npop | (first << 16);
            if (npop > 1u && npop <= kMaxListed) {
                uint32_t at = 0;
#pragma unroll
                for (uint32_t w = 0; w < 4u; ++w) {
                    if ((mw[w] >> lane) & 1ull) s_list[cslot * kMaxListed + at + bits_below_lane(mw[w])] = (uint16_t)(w * 64u + lane);
                    at += (uint32_t)__builtin_popcountll(mw[w]);
                }
            } else if (npop > kMaxListed) {
                if (lane < 4u) s_cmask[cslot * 4u + lane] = lane == 0u ? mw[0] : (lane == 1u ? mw[1] : (lane == 2u ? mw[2] : mw[3]));
            }
        }
        __builtin_amdgcn_wave_barrier();                            // (s_list / s_npop of a cell: written and read by this wave only)
        KMG_STAMP(1);

        // ---- 1b. the sub-cell stage of the listed cells: the thread's own sub-cell, two sweeps over the list ----
        const uint32_t np_first = s_npop[slot];
        const uint32_t npop = valid ? (np_first & 0xFFFFu) : 1u, first = np_first >> 16;
        const bool listed = valid && npop > 1u && npop <= kMaxListed;
        const bool unlisted = valid && npop > kMaxListed;
        const bool occupied = valid && (SUMS ? g23.y != 0 : true);
        uint32_t sm = 0u, istar = 0u, ubits = 0u;
        if (listed) {
            CellBounds sb;
            sb.L0 = sb0.x; sb.L1 = sb0.y; sb.a0 = sb0.z; sb.a1 = sb0.w;
            sb.b0 = sb1.x; sb.b1 = sb1.y; sb.C0 = sb1.z; sb.C1 = sb1.w;
            sb.wC0 = sb2.x; sb.wC1 = sb2.y; sb.wH0 = sb2.z; sb.wH1 = sb2.w;
            uint32_t Ubest = 0x7F7FFFE0u;                           // (upper bound rounded up to a multiple of 32 ulps) | position
            for (uint32_t p = 0; p < npop; ++p) {
                const float4 c = s_cent[s_list[slot * kMaxListed + p]];
                Ubest = min(Ubest, ((float_to_bits(key_range(sb, c.x, c.y, c.z, c.w).hi) + 31u) & ~31u) | p);
            }
            const float Us = mask_threshold(bits_to_float(Ubest & ~31u));
            for (uint32_t p = 0; p < npop; ++p) {
                const float4 c = s_cent[s_list[slot * kMaxListed + p]];
                sm |= (key_range(sb, c.x, c.y, c.z, c.w).lo <= Us ? 1u : 0u) << p;
            }
            istar = Ubest & 31u;
            ubits = Ubest & ~31u;
        }
        const uint32_t np0 = (uint32_t)__builtin_popcount(sm);
        const bool was_decided = listed && occupied && np0 == 1u;   // decided by its bounds
        const bool mine = listed && occupied && np0 > 1u;            // open: the dominance tests, then the scan if still open
        s_mask[threadIdx.x] = sm;
        *reinterpret_cast<uint2 *>(s_occ + slot * 64u + sub * 8u) = occ8;
        // the dominance tests of the WAVE's sub-cells: (sub-cell, candidate other than the reference) as one compact list, a test per
        // lane; the sub-cell's model comes from memory (L2: 96 bytes, read by the one or two lanes that test it).  No workgroup barrier:
        // a wave's cells are its own until their items are listed.
        const uint32_t others = (mine && dominance) ? sm & ~(1u << istar) : 0u;
        if (others) {
            s_ref[threadIdx.x] = ubits | istar;
            const uint32_t nt = (uint32_t)__builtin_popcount(others);
            uint32_t at = atomicAdd(&s_count[8u + wv], nt);
            for (uint32_t m = others; m && at < kOneTests; m &= m - 1u, ++at)
                s_test[wv * kOneTests + at] = (uint16_t)((lane << 5) | (uint32_t)__builtin_ctz(m));
        }
        KMG_STAMP(2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (dominance) {
            const uint32_t n_tests = min(__builtin_amdgcn_readfirstlane(s_count[8u + wv]), kOneTests);
            for (uint32_t t = lane; t < n_tests; t += 64u) {
                const uint32_t ent = (uint32_t)s_test[wv * kOneTests + t];
                const uint32_t e = wv * 64u + (ent >> 5), tp = ent & 31u;
                const uint32_t r = s_ref[e];
                const float4 cj = s_cent[s_list[(e >> 3) * kMaxListed + tp]], ci4 = s_cent[s_list[(e >> 3) * kMaxListed + (r & 31u)]];
                HalfModel mdl;
                mdl.load(sub_affine, (uint64_t)s_cell[e >> 3] * 8u + (e & 7u));
                if (dominated(mdl, cj, ci4, bits_to_float(r & ~31u))) atomicAnd(&s_mask[e], ~(1u << tp));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        KMG_STAMP(3);

        // ---- 1c. what is left of every sub-cell's set: decided / open; uniform cells; items and pending cells ----
        const uint32_t nm = s_mask[threadIdx.x];
        const uint32_t np = (uint32_t)__builtin_popcount(nm);
        const bool one = mine && np == 1u, still = mine && np > 1u;
        const uint32_t X = listed ? (uint32_t)s_list[slot * kMaxListed + (nm ? (uint32_t)__builtin_ctz(nm) : 0u)] : first;
        const bool open = still || (unlisted && occupied);
        const uint32_t scan8 = group8_or(open ? 1u << sub : 0u);
        const bool settled = one || was_decided;
        const uint32_t xmin = group8_min_u32(settled ? X : 255u), xmax = group8_max_u32(settled ? X : 0u);
        const bool single = valid && npop == 1u;
        // every occupied sub-cell of the cell went to one centroid (a cell without occupied sub-cells does not occur: the work list)
        const bool uniform = single || (listed && scan8 == 0u && xmin == xmax);
        const uint32_t Xu = single ? first : xmin;
        if (SUMS) {
            if (uniform) {
                if (sub < 4u) atomicAdd(my_bins + 4ull * Xu + sub, (unsigned long long)cell_sum);
            } else if (settled) {
                unsigned long long *to = my_bins + 4ull * X;
                atomicAdd(to + 0, (unsigned long long)g01.x); atomicAdd(to + 1, (unsigned long long)g01.y);
                atomicAdd(to + 2, (unsigned long long)g23.x); atomicAdd(to + 3, (unsigned long long)g23.y);
            }
        }
        {
            const uint32_t x4 = (settled ? X : 0u) * 0x01010101u;
            const uint4 xv = make_uint4(x4, x4, x4, x4);
            if (uniform) {
                if (sub == 0u) pair_entries[cell] = pair_entry(Xu, Xu, 0u, 0u, 0u);
                if (flags & 1u) {
                    const uint32_t m4 = Xu * 0x01010101u;
                    uint4 *dst = reinterpret_cast<uint4 *>(colour_labels + (uint64_t)cell * kCellColours + sub * 64u);
                    dst[0] = dst[1] = dst[2] = dst[3] = make_uint4(m4, m4, m4, m4);
                }
            } else if (valid) {
                uint4 *dst = reinterpret_cast<uint4 *>(s_lbl + slot * kCellColours + sub * 64u);
                dst[0] = xv; dst[1] = xv; dst[2] = xv; dst[3] = xv;
            }
        }
        unsigned long long pend0_b = 0ull;
        {
            // items: the open sub-cells of a cell, two per item in ascending order; the lane that heads a pair builds the item
            const uint32_t rank = (uint32_t)__builtin_popcount(scan8 & ((1u << sub) - 1u));
            const bool head = open && (rank & 1u) == 0u;
            const uint32_t after = scan8 & ~((2u << sub) - 1u);
            const uint32_t s1 = after ? (uint32_t)__builtin_ctz(after) : 8u;
            const uint32_t un = listed ? (nm | ((head && s1 < 8u) ? s_mask[threadIdx.x - sub + s1] : 0u)) : 0u;
            const uint32_t n_un = (uint32_t)__builtin_popcount(un);
            const uint32_t type = unlisted ? 2u : (n_un > kItemCands ? 1u : 0u);
            // a cell that needs an entry gets it from the wave that scans the LAST of its items (s_left counts them down);
            // one without items (its sub-cells were all decided, to different centroids) from its own wave, after the scan
            const bool pend = valid && !uniform && sub == 0u;
            if (pend) s_left[slot] = ((uint32_t)__builtin_popcount(scan8) + 1u) >> 1;
            pend0_b = __ballot(pend && scan8 == 0u);
            const unsigned long long item_b = __ballot(head);
            if (bal_state) {
                // this task's weight in the next pass's deal: what its items will cost the scan (a fixed part + a part per candidate)
                const uint32_t cost = wave_add_u32(head ? 4u + (type == 2u ? npop : n_un) : 0u);
                if (lane == 0u) bal_state[(bal_pass & 1u) * 2u * kOneTaskSlots + kOneTaskSlots + blockIdx.x * kOneWaves + wv] = (uint16_t)min(cost, 65535u);
            }
            uint32_t i_base = 0u;
            if (lane == 0u && item_b) i_base = atomicAdd(&s_count[0], (uint32_t)__builtin_popcountll(item_b));
            i_base = __builtin_amdgcn_readfirstlane(i_base);
            if (head) {
                uint32_t iw0 = un, iw1 = 0u, iw2 = 0u;                // (type 1: the positions; type 2: unused)
                if (type == 0u) {
                    iw0 = 0u;
                    uint32_t q = 0u;
                    for (uint32_t m = un; m; m &= m - 1u, ++q) {
                        const uint32_t cc = (uint32_t)s_list[slot * kMaxListed + (uint32_t)__builtin_ctz(m)] << (8u * (q & 3u));
                        iw0 |= q < 4u ? cc : 0u; iw1 |= (q >= 4u && q < 8u) ? cc : 0u; iw2 |= q >= 8u ? cc : 0u;
                    }
                }
                s_item[i_base + bits_below_lane(item_b)] = make_uint4(slot | (sub << 7) | (s1 << 10) | (n_un << 14) | (type << 30), iw0, iw1, iw2);
            }
            if (stats) {
                st_single += (uint32_t)__builtin_popcountll(__ballot(single && sub == 0u));
                st_multi += (uint32_t)__builtin_popcountll(__ballot(valid && !single && sub == 0u));
                st_unlisted += (uint32_t)__builtin_popcountll(__ballot(unlisted && sub == 0u));
                st_decided += (uint32_t)__builtin_popcountll(__ballot(was_decided));
                st_scanned += (uint32_t)__builtin_popcountll(__ballot(mine || (unlisted && occupied)));
                st_cands += wave_add_u32(mine ? np0 : ((unlisted && occupied) ? npop : 0u));
                st_removed += wave_add_u32((uint32_t)__builtin_popcount(sm & ~nm));
                st_one += (uint32_t)__builtin_popcountll(__ballot(one));
            }
        }
        KMG_STAMP(4);
        __syncthreads();
        const uint32_t n_items = s_count[0];
        KMG_STAMP_VALUE(8, n_items); KMG_STAMP_VALUE(10, s_count[8]);

        // ---- 3. (defined first: 2. calls it) the pair entry of a cell from its 512 labels in LDS; the labels leave ----
        auto cell_entry = [&](const uint32_t ps) {
            const uint32_t pcell = __builtin_amdgcn_readfirstlane(s_cell[ps]);
            const uint2 lv = *reinterpret_cast<const uint2 *>(s_lbl + ps * kCellColours + lane * 8u);
            uint32_t e = kPairPending;
            if (!(flags & kCubeNoEntries)) {
                const uint32_t occ = (uint32_t)s_occ[ps * 64u + lane];
                uint32_t idx[8];
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) { idx[q] = (lv.x >> (8u * q)) & 0xFFu; idx[4u + q] = (lv.y >> (8u * q)) & 0xFFu; }
                e = cell_pair_entry(idx, occ, lane);
            }
            if (lane == 0u) pair_entries[pcell] = e;
            *reinterpret_cast<uint2 *>(colour_labels + (uint64_t)pcell * kCellColours + lane * 8u) = lv;
        };

        // ---- 2. scan: the items handed out one by one (an LDS counter: whoever is free takes the next), one colour of each of the
        // item's two sub-cells per lane, three items' colours in flight ----
        {
            uint32_t A_h = 0u, A_w0 = 0u, A_w1 = 0u, A_w2 = 0u, B_h = 0u, B_w0 = 0u, B_w1 = 0u, B_w2 = 0u, C_h = 0u, C_w0 = 0u, C_w1 = 0u, C_w2 = 0u;
            float4 A_v0, A_v1, B_v0, B_v1, C_v0, C_v1;
            uint32_t A_c0 = 1u, A_c1 = 1u, B_c0 = 1u, B_c1 = 1u, C_c0 = 1u, C_c1 = 1u;
            long long A_g = 0, B_g = 0, C_g = 0;
            bool A_ok = false, B_ok = false, C_ok = false;
            // (an exhausted set re-requests item 0, unused: a conditional request makes the compiler wait for the registers)
#define KMG_REQUEST_ITEM(X)                                                                                       \
            do {                                                                                                  \
                uint32_t next_ = 0u;                                                                              \
                if (lane == 0u) next_ = atomicAdd(&s_count[4], 1u);                                               \
                next_ = __builtin_amdgcn_readfirstlane(next_);                                                    \
                X##_ok = next_ < n_items;                                                                         \
                const uint4 it_ = s_item[next_ < n_items ? next_ : 0u];                                           \
                X##_h = __builtin_amdgcn_readfirstlane(it_.x); X##_w0 = __builtin_amdgcn_readfirstlane(it_.y);    \
                X##_w1 = __builtin_amdgcn_readfirstlane(it_.z); X##_w2 = __builtin_amdgcn_readfirstlane(it_.w);   \
                const uint32_t cell_ = __builtin_amdgcn_readfirstlane(s_cell[X##_h & 127u]);                       \
                const uint32_t c0_ = cell_ * kCellColours + ((X##_h >> 7) & 7u) * 64u + lane;                     \
                const uint32_t c1_ = cell_ * kCellColours + ((X##_h >> 10) & 7u) * 64u + lane;   /* s1 == 8: sub-cell 0, unused */ \
                X##_v0 = lab_table[c0_]; X##_v1 = lab_table[c1_];                                                 \
                if (SUMS) { X##_c0 = hist[c0_]; X##_c1 = hist[c1_]; X##_g = sub_agg[(uint64_t)cell_ * 32u + (lane & 31u)]; } \
            } while (0)
            auto scan_item = [&](const uint32_t hdr, const uint32_t w0, const uint32_t w1, const uint32_t w2, const float4 v0, const float4 v1,
                                 const uint32_t cnt0, const uint32_t cnt1, const long long sagg) {
                const uint32_t islot = hdr & 127u, s0 = (hdr >> 7) & 7u, s1 = (hdr >> 10) & 15u, n = (hdr >> 14) & 63u, type = hdr >> 30;
                const PixelTerms pt0 = pixel_terms_fast(v0.x, v0.y, v0.z, v0.w), pt1 = pixel_terms_fast(v1.x, v1.y, v1.z, v1.w);
                uint32_t ix0 = 0u, ix1 = 0u;
                if (type == 0u) {
                    // (k_cube_scan's item loop: lane p < n holds the p-th candidate; key | position, one integer min / med3 per visit)
                    const uint32_t my_cand = ((lane < 4u ? w0 : (lane < 8u ? w1 : w2)) >> (8u * (lane & 3u))) & 0xFFu;
                    uint32_t b0 = 0x7F7FFFFFu, r0 = 0x7F7FFFFFu, b1 = 0x7F7FFFFFu, r1 = 0x7F7FFFFFu;   // smallest / runner-up
                    const f32x2 qL = {pt0.L, pt1.L}, qa = {pt0.a, pt1.a}, qb = {pt0.b, pt1.b}, qC = {pt0.C, pt1.C};
                    const f32x2 qwC = {pt0.wC, pt1.wC}, qwH = {pt0.wH, pt1.wH};
                    auto cand_at = [&](uint32_t p) { return ((p < 4u ? w0 : (p < 8u ? w1 : w2)) >> (8u * (p & 3u))) & 0xFFu; };
                    for (uint32_t p = 0; p < n; ++p) {
                        const float4 c = s_cent[cand_at(p)];
                        const f32x2 dL = qL - c.x, da = qa - c.y, db = qb - c.z, dC = qC - c.w;
                        const f32x2 dC2 = dC * dC;
                        const f32x2 t = __builtin_elementwise_fma(db, db, da * da);
                        f32x2 h = t - dC2;
                        h.x = fmaxf(h.x, 0.0f); h.y = fmaxf(h.y, 0.0f);
                        const f32x2 key = __builtin_elementwise_fma(h, qwH, __builtin_elementwise_fma(dC2, qwC, dL * dL));
                        uint32_t u0, u1;                               // (key & ~31) | position (wave-uniform)
                        asm("v_bfi_b32 %0, 31, %1, %2" : "=v"(u0) : "s"(p), "v"(float_to_bits(key.x)));
                        asm("v_bfi_b32 %0, 31, %1, %2" : "=v"(u1) : "s"(p), "v"(float_to_bits(key.y)));
                        r0 = umed3(u0, b0, r0); b0 = min(b0, u0);
                        r1 = umed3(u1, b1, r1); b1 = min(b1, u1);
                    }
                    uint32_t p0 = b0 & 31u, p1 = b1 & 31u;
                    // near-tie repair (kmg_math.h): rare; decided by the literal distance, first minimum wins
                    const float thr0 = tie_threshold(bits_to_float(b0 & ~31u)), thr1 = tie_threshold(bits_to_float(b1 & ~31u));
                    const bool near0 = bits_to_float(r0 & ~31u) <= thr0, near1 = bits_to_float(r1 & ~31u) <= thr1;
                    if (__ballot(near0 || near1)) {
                        float lb0 = 100000.0f, lb1 = 100000.0f;       // find_centroid.wgsl:29-30
                        uint32_t li0 = 0u, li1 = 0u;
                        for (uint32_t p = 0; p < n; ++p) {
                            const float4 c = s_cent[cand_at(p)];
                            if (near0 && cie94_key(pt0, c.x, c.y, c.z, c.w) <= thr0) {
                                const float d = cie94_c(v0.x, v0.y, v0.z, v0.w, c.x, c.y, c.z, c.w);
                                if (d < lb0) { lb0 = d; li0 = p; }
                            }
                            if (near1 && cie94_key(pt1, c.x, c.y, c.z, c.w) <= thr1) {
                                const float d = cie94_c(v1.x, v1.y, v1.z, v1.w, c.x, c.y, c.z, c.w);
                                if (d < lb1) { lb1 = d; li1 = p; }
                            }
                        }
                        p0 = near0 ? li0 : p0;
                        p1 = near1 ? li1 : p1;
                    }
                    ix0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p0 << 2), (int)my_cand);
                    ix1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p1 << 2), (int)my_cand);
                } else {
                    // rare: more than kItemCands candidates for the pair (by list position, w0 = their mask), or a cell without a
                    // list (every candidate of the cell's mask) -- k_cube_scan's general loop: float keys, compare and select
                    float best0 = 1.0e10f, second0 = 1.0e10f, best1 = 1.0e10f, second1 = 1.0e10f;
                    const uint32_t rounds = type == 1u ? 1u : words;
                    auto cand_word = [&](uint32_t w) -> unsigned long long { return type == 1u ? (unsigned long long)w0 : uniform_u64(s_cmask[islot * 4u + w]); };
                    auto cand_of = [&](uint32_t w, uint32_t bit) -> uint32_t {
                        return type == 1u ? __builtin_amdgcn_readfirstlane((uint32_t)s_list[islot * kMaxListed + bit]) : w * 64u + bit;
                    };
                    for (uint32_t w = 0; w < rounds; ++w) {
                        for (unsigned long long m = cand_word(w); m; m &= m - 1ull) {
                            const uint32_t j = cand_of(w, (uint32_t)__builtin_ctzll(m));
                            const float4 c = s_cent[j];
                            const float d0 = cie94_key(pt0, c.x, c.y, c.z, c.w), d1 = cie94_key(pt1, c.x, c.y, c.z, c.w);
                            second0 = __builtin_amdgcn_fmed3f(d0, best0, second0);
                            second1 = __builtin_amdgcn_fmed3f(d1, best1, second1);
                            if (d0 < best0) { best0 = d0; ix0 = j; }
                            if (d1 < best1) { best1 = d1; ix1 = j; }
                        }
                    }
                    const float thr0 = tie_threshold(best0), thr1 = tie_threshold(best1);
                    const bool near0 = second0 <= thr0, near1 = second1 <= thr1;
                    if (__ballot(near0 || near1)) {
                        float lb0 = 100000.0f, lb1 = 100000.0f;
                        uint32_t li0 = 0u, li1 = 0u;
                        for (uint32_t w = 0; w < rounds; ++w) {
                            for (unsigned long long m = cand_word(w); m; m &= m - 1ull) {
                                const uint32_t j = cand_of(w, (uint32_t)__builtin_ctzll(m));
                                const float4 c = s_cent[j];
                                if (near0 && cie94_key(pt0, c.x, c.y, c.z, c.w) <= thr0) {
                                    const float d = cie94_c(v0.x, v0.y, v0.z, v0.w, c.x, c.y, c.z, c.w);
                                    if (d < lb0) { lb0 = d; li0 = j; }
                                }
                                if (near1 && cie94_key(pt1, c.x, c.y, c.z, c.w) <= thr1) {
                                    const float d = cie94_c(v1.x, v1.y, v1.z, v1.w, c.x, c.y, c.z, c.w);
                                    if (d < lb1) { lb1 = d; li1 = j; }
                                }
                            }
                        }
                        ix0 = near0 ? li0 : ix0;
                        ix1 = near1 ? li1 : ix1;
                    }
                }
                // what a scanned sub-cell leaves behind: its 64 labels (LDS) and its sums -- the sub-cell's total goes to a reference
                // label R, a colour with another label moves its own contribution from R to that label
                auto finish = [&](uint32_t s, uint32_t ix, uint32_t cnt, float vL, float va, float vb) {
                    s_lbl[islot * kCellColours + s * 64u + lane] = (uint8_t)ix;
                    if (!SUMS) return;
                    const bool counts = cnt != 0u;
                    const unsigned long long occm = __ballot(counts);
                    if (!occm) return;
                    const uint32_t X0 = lane_value(ix, (uint32_t)__builtin_ctzll(occm));
                    const unsigned long long other = __ballot(counts && ix != X0);
                    uint32_t R = X0;
                    if (other) {
                        const uint32_t X1 = lane_value(ix, (uint32_t)__builtin_ctzll(other));
                        if (__builtin_popcountll(__ballot(counts && ix == X1)) > __builtin_popcountll(occm & ~other)) R = X1;
                    }
                    if ((lane >> 2) == s) atomicAdd(bins + 4ull * R + (lane & 3u), (unsigned long long)sagg);
                    if (other && counts && ix != R) {
                        const long long m = (long long)cnt;
                        const long long c0 = m * (long long)lab_fix(vL), c1 = m * (long long)lab_fix(va), c2 = m * (long long)lab_fix(vb);
                        unsigned long long *to = my_bins + 4ull * ix, *from = my_bins + 4ull * R;
                        atomicAdd(to + 0, (unsigned long long)c0); atomicAdd(from + 0, (unsigned long long)(-c0));
                        atomicAdd(to + 1, (unsigned long long)c1); atomicAdd(from + 1, (unsigned long long)(-c1));
                        atomicAdd(to + 2, (unsigned long long)c2); atomicAdd(from + 2, (unsigned long long)(-c2));
                        atomicAdd(to + 3, (unsigned long long)m);  atomicAdd(from + 3, (unsigned long long)(-m));
                    }
                };
                finish(s0, ix0, cnt0, v0.x, v0.y, v0.z);
                if (s1 < 8u) finish(s1, ix1, cnt1, v1.x, v1.y, v1.z);
                // the cell's items count down: the wave that scans the last one finds all 512 labels in LDS (the release / acquire
                // pair orders this wave's label writes before its decrement, and the other waves' before this wave's reads)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                uint32_t left = 0u;
                if (lane == 0u) left = atomicSub(&s_left[islot], 1u);
                left = __builtin_amdgcn_readfirstlane(left);
                if (left == 1u) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    cell_entry(islot);
                }
            };
            KMG_REQUEST_ITEM(A);
            KMG_REQUEST_ITEM(B);
            KMG_REQUEST_ITEM(C);
            for (;;) {
                if (!A_ok) break;
                scan_item(A_h, A_w0, A_w1, A_w2, A_v0, A_v1, A_c0, A_c1, A_g);
                if (!B_ok) break;
                KMG_REQUEST_ITEM(A);
                scan_item(B_h, B_w0, B_w1, B_w2, B_v0, B_v1, B_c0, B_c1, B_g);
                if (!C_ok) break;
                KMG_REQUEST_ITEM(B);
                scan_item(C_h, C_w0, C_w1, C_w2, C_v0, C_v1, C_c0, C_c1, C_g);
                KMG_REQUEST_ITEM(C);
            }
#undef KMG_REQUEST_ITEM
        }
        KMG_STAMP(5);
        // the wave's own cells without items
        for (unsigned long long m = pend0_b; m; m &= m - 1ull) cell_entry(wv * 8u + ((uint32_t)__builtin_ctzll(m) >> 3));
        KMG_STAMP(7);
    }
    if (stats && lane == 0u) {
        atomicAdd(stats + 0, st_single); atomicAdd(stats + 1, st_multi); atomicAdd(stats + 2, st_decided); atomicAdd(stats + 3, st_scanned);
        atomicAdd(stats + 4, st_cands); atomicAdd(stats + 5, st_unlisted); atomicAdd(stats + 6, st_removed); atomicAdd(stats + 7, st_one);
    }
    if (SUMS) flush_bins(bins, k, kOneRepl, bin_stride, sums, n_rows);
}

#ifdef KMG_TOOLS
extern "C" KMG_API int kmg_tools_cube_one_stamps(unsigned long long *out, uint32_t n)
try {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_one_stamps), sizeof(unsigned long long) * std::min<size_t>(n, (kCells / kOneCells) * 12u));
}
KMG_ABI_CATCH
#endif

static uint32_t grid_or(const char *e, uint32_t dflt)
{
    if (e) { const int v = atoi(e); if (v >= 1 && v <= 65536) return (uint32_t)v; }
    return dflt;
}
#define env_grid(name, dflt) grid_or(KMG_TOOLS_ENV(name), (dflt))

uint32_t cube_replicas(uint32_t k)
{
    // copies of the scan kernel's LDS bins (lanes spread their atomic adds over them).  Measured at k = 256 (round 3,
    // profiles/r03_*): 1 copy 73.4 us, 2 copies 69.5 us, 4 copies 97.6 us (the bins then take the LDS of a workgroup per CU)
    uint32_t r = 4;
    if (const char *e = KMG_TOOLS_ENV("KMG_CUBE_REPL")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) r = (uint32_t)v; }
    while (r > 1u && (uint64_t)r * (k * 32ull + 32ull) > 33024ull) r >>= 1;
    return r;
}

size_t cube_work_bytes() { return sizeof(CellWork) * (size_t)kCells + sizeof(uint32_t) * kListsWords; }

// the cube pass of this k (and these caller flags) is ONE launch: k_cube_small, or k_cube_one (32 < k <= 256, images without hot
// cells: a photograph's few hundred cells with long candidate lists want the stage / scan launches, which spread them over the
// device) -- its tail can ride on a label pass
bool cube_single_launch(uint32_t k, uint32_t flags)
{
    return k <= kSmallMaxK || (k <= 256u && !(flags & kCubeHot));
}

hipError_t launch_cube(const uint32_t *hist, const int64_t *agg, const int64_t *sub_agg, const uint8_t *occ_bits,
                       const uint32_t *work, const CellBounds *bounds, const CellBounds *sub_bounds, const Centroid *cent,
                       uint32_t k, const float4 *lab_table, uint64_t *masks, void *cell_work, void *colour_labels,
                       uint16_t *sub_table, int64_t *sums, uint32_t n_rows, uint32_t flags, unsigned long long *stats,
                       hipStream_t st, const CubeTail *tail, const float *sub_affine, const CubeBalance *balance)
{
    const CubeTail tl = (tail && hist && n_rows <= 1u) ? *tail : CubeTail();
    const uint32_t kpad = (k + 63u) & ~63u;
    const bool with_sums = hist != nullptr;
    const uint32_t repl = with_sums ? cube_replicas(k) : 1u;
    const size_t lds_stage = sizeof(float4) * kpad + (with_sums ? sizeof(unsigned long long) * 4ull * k : 0) +
                             sizeof(uint32_t) * (kBlock / 64) * kMaxListed + sizeof(unsigned long long) * (kBlock / 64) * (kpad / 64u) +
                             (k <= 256 ? (kBlock / 64) * (32u * sizeof(unsigned long long) + kMaxLong * sizeof(uint16_t)) : 0u);
    const size_t lds_scan = sizeof(float4) * kpad + (with_sums ? sizeof(unsigned long long) * (4ull * k + 4ull) * repl : 0) +
                            sizeof(float4) * (kScanBlock / 64) * kMaxListed + (k <= 256 ? 1u : 2u) * (kScanBlock / 64) * kCellColours;
    {
        // the largest k (KMG_MAX_K) takes ~150 KiB per workgroup: refuse a launch the device cannot hold instead of failing in it
        const size_t lds_max = device_info().lds_max;
        if (lds_stage > lds_max || lds_scan > lds_max) return hipErrorInvalidValue;
    }
    const size_t lds_max_dev = device_info().lds_max;
    // small centroid tables: the whole pass in one launch (k_cube_small) + the tail workgroup
    static const bool small_on = tools_env_int(KMG_TOOLS_ENV("KMG_CUBE_SMALL"), 1) != 0;
    if (k <= kSmallMaxK && small_on) {
        static const uint32_t g_small = env_grid("KMG_SMALL_GRID", kCells / kSmallCells);
        const uint32_t kp = k <= 8u ? 8u : (k <= 16u ? 16u : 32u);
        const size_t lds = sizeof(float4) * kp + (with_sums ? sizeof(unsigned long long) * (4ull * k + 4ull) * kSmallRepl : 0) +
                           (size_t)kSmallCells * kCellColours + sizeof(uint32_t) * kSmallBlock * 2u + sizeof(uint32_t) * kSmallCells +
                           sizeof(uint32_t) * 4u + sizeof(uint16_t) * (kSmallBlock + kSmallCells + kSmallTests) + kSmallBlock;
        if (!n_rows) n_rows = 1u;
        // KMG_CUBE_SMALL=2: k_cube_small as the stage only, then the scan and entries launches of the general pass
        static const bool split = tools_env_int(KMG_TOOLS_ENV("KMG_CUBE_SMALL"), 1) == 2;
        const uint32_t sflags = (flags & ~0x100u) | (split ? kSmallEmit : 0u);
#define KMG_SMALL(KP, S)                                                                                                    \
        hipLaunchKernelGGL((k_cube_small<KP, S>), dim3(g_small), dim3(kSmallBlock), lds, st, hist, agg, sub_agg, occ_bits,   \
                           work, bounds, sub_bounds, sub_affine, cent, k, lab_table, masks, (CellWork *)cell_work,            \
                           (uint8_t *)colour_labels, sub_table, sums, n_rows, sflags, stats)
        if (with_sums) { if (kp == 8u) KMG_SMALL(8, true); else if (kp == 16u) KMG_SMALL(16, true); else KMG_SMALL(32, true); }
        else           { if (kp == 8u) KMG_SMALL(8, false); else if (kp == 16u) KMG_SMALL(16, false); else KMG_SMALL(32, false); }
#undef KMG_SMALL
        if (split) {
            static const uint32_t g_scan2 = env_grid("KMG_SCAN_GRID", kCubeGrid), g_pairs2 = env_grid("KMG_PAIRS_GRID", kCubeGrid);
            const uint32_t kpad2 = (k + 63u) & ~63u, repl2 = with_sums ? cube_replicas(k) : 1u;
            const size_t lds_scan2 = sizeof(float4) * kpad2 + (with_sums ? sizeof(unsigned long long) * (4ull * k + 4ull) * repl2 : 0) +
                                     sizeof(float4) * (kScanBlock / 64) * kMaxListed + (kScanBlock / 64) * kCellColours;
            if (with_sums)
                hipLaunchKernelGGL((k_cube_scan<uint8_t, true>), dim3(g_scan2), dim3(kScanBlock), lds_scan2, st, hist, sub_agg, work, cent, k,
                                   lab_table, masks, (const CellWork *)cell_work, (uint8_t *)colour_labels, sub_table, sums, n_rows, repl2, flags);
            else
                hipLaunchKernelGGL((k_cube_scan<uint8_t, false>), dim3(g_scan2), dim3(kScanBlock), lds_scan2, st, hist, sub_agg, work, cent, k,
                                   lab_table, masks, (const CellWork *)cell_work, (uint8_t *)colour_labels, sub_table, sums, n_rows, repl2, flags);
            hipLaunchKernelGGL((k_cube_pairs<uint8_t>), dim3((flags & kCubeNoEntries) ? 1u : g_pairs2), dim3(kBlock), 0, st, work,
                               with_sums ? 1 : 0, occ_bits, (const uint8_t *)colour_labels, sub_table, flags, sums, k, tl);
            return hipGetLastError();
        }
        if (tl.acc_out)
            hipLaunchKernelGGL((k_cube_pairs<uint8_t>), dim3(1), dim3(kBlock), 0, st, work, 1, occ_bits,
                               (const uint8_t *)colour_labels, sub_table, flags | kCubeNoEntries, sums, k, tl);
        return hipGetLastError();
    }
    // 32 < k <= 256 without hot cells: the whole pass in one launch (k_cube_one) + the tail workgroup
    if (cube_single_launch(k, flags)) {
        // one workgroup of 16 waves per CU (150 KiB of LDS at k = 256): its 128 cells share one item pool
        const size_t lds = cube_one_lds_bytes(k, with_sums);
        if (lds > lds_max_dev) return hipErrorInvalidValue;
        if (!n_rows) n_rows = 1u;
        static const bool balance_on = tools_env_int(KMG_TOOLS_ENV("KMG_BALANCE"), 1) != 0;      // (tools build: 0 = the fixed deal)
        if (!balance_on) balance = nullptr;
        if (with_sums)
            hipLaunchKernelGGL((k_cube_one<true>), dim3(kCells / kOneCells), dim3(kOneBlock), lds, st, hist, agg, sub_agg, occ_bits, work, bounds,
                               sub_bounds, sub_affine, cent, k, lab_table, masks, (uint8_t *)colour_labels, sub_table, sums, n_rows, flags, stats,
                               balance ? balance->state : nullptr, balance ? balance->pass : 0u);
        else
            hipLaunchKernelGGL((k_cube_one<false>), dim3(kCells / kOneCells), dim3(kOneBlock), lds, st, hist, agg, sub_agg, occ_bits, work, bounds,
                               sub_bounds, sub_affine, cent, k, lab_table, masks, (uint8_t *)colour_labels, sub_table, sums, n_rows, flags, stats,
                               (uint16_t *)nullptr, 0u);
        if (tl.acc_out)
            hipLaunchKernelGGL((k_cube_pairs<uint8_t>), dim3(1), dim3(kBlock), 0, st, work, 1, occ_bits,
                               (const uint8_t *)colour_labels, sub_table, flags | kCubeNoEntries, sums, k, tl);
        return hipGetLastError();
    }
    // (the scan kernel's cells differ a lot in cost: finer hand-out, 2 cells per wave, measured 83 -> 75 us)
    // (stage: 1536 workgroups = one full round at its 6 waves per SIMD: 33 -> 30-31 us against 2048, round 3)
    static const uint32_t g_stage = env_grid("KMG_CUBE_GRID", 1536u), g_scan = env_grid("KMG_SCAN_GRID", kCubeGrid),
                          g_pairs = env_grid("KMG_PAIRS_GRID", kCubeGrid);
    CellWork *cw = (CellWork *)cell_work;
    if (!n_rows) n_rows = 1u;
    if (k <= 256u && tools_env_int(KMG_TOOLS_ENV("KMG_SPLIT_LONG"), 1) != 0) {
        // (the stage kernel's workgroups append to the list of long-list cells as they meet them: its counters are cleared ahead of
        // the launch, not by one of its workgroups)
        flags |= kCubeSplitLong;
        hipError_t e = hipMemsetAsync(reinterpret_cast<uint32_t *>((CellWork *)cell_work + kCells) + kLongCount, 0, sizeof(uint32_t) * kLongSegs, st);
        if (e != hipSuccess) return e;
    }
#define KMG_CUBE(T, S)                                                                                                      \
    do {                                                                                                                    \
        hipLaunchKernelGGL((k_cube_stage<T, S>), dim3(g_stage), dim3(kBlock), lds_stage, st, agg, sub_agg, work, bounds,     \
                           sub_bounds, cent, k, masks, cw, (T *)colour_labels, sub_table, sums, n_rows, flags, stats);      \
        hipLaunchKernelGGL((k_cube_scan<T, S>), dim3(g_scan), dim3(kScanBlock), lds_scan, st, hist, sub_agg, work, cent, k,   \
                           lab_table, masks, cw, (T *)colour_labels, sub_table, sums, n_rows, repl, flags);                 \
        hipLaunchKernelGGL((k_cube_pairs<T>), dim3((flags & kCubeNoEntries) ? 1u : g_pairs), dim3(kBlock), 0, st, work,   \
                           S ? 1 : 0, occ_bits, (const T *)colour_labels, sub_table, flags, sums, k, tl);                   \
    } while (0)
    if (k <= 256) { if (with_sums) KMG_CUBE(uint8_t, true); else KMG_CUBE(uint8_t, false); }
    else          { if (with_sums) KMG_CUBE(uint16_t, true); else KMG_CUBE(uint16_t, false); }
#undef KMG_CUBE
    return hipGetLastError();
}

hipError_t launch_cube_entries(const uint32_t *work, const uint8_t *occ_bits, const void *colour_labels, uint16_t *sub_table,
                               uint32_t k, hipStream_t st)
{
    static const uint32_t g_pairs = env_grid("KMG_PAIRS_GRID", kCubeGrid);
    if (k <= 256)
        hipLaunchKernelGGL((k_cube_pairs<uint8_t>), dim3(g_pairs), dim3(kBlock), 0, st, work, work ? 1 : 0, occ_bits,
                           (const uint8_t *)colour_labels, sub_table, 0u, (int64_t *)nullptr, k, CubeTail());
    else
        hipLaunchKernelGGL((k_cube_pairs<uint16_t>), dim3(g_pairs), dim3(kBlock), 0, st, work, work ? 1 : 0, occ_bits,
                           (const uint16_t *)colour_labels, sub_table, 0u, (int64_t *)nullptr, k, CubeTail());
    return hipGetLastError();
}

}  // namespace kmg
