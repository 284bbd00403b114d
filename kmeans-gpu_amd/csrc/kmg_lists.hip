// kmg_lists.hip -- the ordered-dither and meld output passes on large images with k <= 512 (mix_colors.wgsl:50-83, :29-48 + :85-90):
// candidate lists per cell of a grid over Lab, and the passes that walk them.
//
// A dithered pixel is compared with the centroids at Lab(pixel) + off (1, 1, 1), off one of 16 Bayer offsets.  Rounds 1-3a
// pruned per (RGB cell, Bayer index): 2^19 slots whose 32-byte records (16.7 MiB) are fetched once per pixel from all over the
// table -- 3.3 GB of Infinity Cache traffic per 8192^2 image at k = 256 (profiles/r03c_apply_pmc.txt), which bounded the pass at
// 1.15 ms however few instructions the list walk took.  The shifted point is an ordinary point of Lab, so the lists here belong to
// cells of a 4 x 4 x 4 grid over Lab itself: no Bayer dimension, a tighter box than an RGB cell swept by 16 offsets, and the
// cells an image can reach (the sRGB gamut widened by the offsets: ~20 000 of 207 360) make a table of well under 1 MiB that
// stays in every XCD's L2.  The meld pass needs a pixel's two closest centroids: the same grid (no offset), lists that keep
// whatever can be among the two.
//
// Exactness is the near-tie scheme of kmg_math.h: the lists keep every centroid whose lower key bound over the cell is not above
// (1 + kMaskSlack) times the smallest (meld: second smallest) upper bound, the passes order by a key and re-decide near-ties
// with the literal distance.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kmg_internal.h"
#include "kmg_table_dev.h"

namespace kmg {

// ---- the grid ---------------------------------------------------------------------------------------------------------------
// L in [-24, 136), a and b in [-144, 144): 40 x 72 x 72 cells of 4 units.  sRGB spans L 0..100, a -87..99, b -108..95; the
// offsets are threshold (bayer / 16 - 1/2) with threshold = palette diameter / sqrt(k) (<= 37 for k >= 16).  Cells on the rim
// of the grid are unbounded outwards and are marked "scan everything" (count 255): exact for any input, merely slow.
constexpr int kLabNL = 40, kLabNA = 72;
static_assert(kLabCells == (uint32_t)(kLabNL * kLabNA * kLabNA), "grid size (kmg_table.h)");
constexpr float kLabStep = 4.0f, kLabL0 = -24.0f, kLabA0 = -144.0f;

__device__ __forceinline__ uint32_t lab_axis_index(float x, float origin, int n)
{
    // (x - origin) / 4 clamped to [0, n - 1], rounded down; NaN -> a rim cell
    const float t = (x - origin) * (1.0f / kLabStep);
    return (uint32_t)__builtin_amdgcn_fmed3f(t, 0.0f, (float)(n - 1));
}

__device__ __forceinline__ uint32_t lab_cell_index(float L, float a, float b)
{
    const uint32_t iL = lab_axis_index(L, kLabL0, kLabNL), ia = lab_axis_index(a, kLabA0, kLabNA), ib = lab_axis_index(b, kLabA0, kLabNA);
    return (iL * (uint32_t)kLabNA + ia) * (uint32_t)kLabNA + ib;
}

// (the L interval of a cell alone: what changes along a column of the grid)
__device__ __forceinline__ void lab_cell_L_bounds(uint32_t iL, CellBounds &s)
{
    constexpr float e = 0.0009765625f;
    s.L0 = kLabL0 + kLabStep * (float)iL - e; s.L1 = kLabL0 + kLabStep * (float)(iL + 1u) + e;
}

// bounds of the per-pixel terms over the points that lab_cell_index sends to interior cell (iL, ia, ib): the cell's box widened
// by 2^-10 on every side (x - origin rounds once, by at most 2^-17 at these magnitudes; the scaling by 1/4 is exact)
__device__ __forceinline__ CellBounds lab_cell_bounds(uint32_t iL, uint32_t ia, uint32_t ib)
{
    constexpr float e = 0.0009765625f;
    CellBounds s;
    lab_cell_L_bounds(iL, s);
    s.a0 = kLabA0 + kLabStep * (float)ia - e; s.a1 = kLabA0 + kLabStep * (float)(ia + 1u) + e;
    s.b0 = kLabA0 + kLabStep * (float)ib - e; s.b1 = kLabA0 + kLabStep * (float)(ib + 1u) + e;
    float ma, Ma, mb, Mb;
    abs_range(s.a0, s.a1, 0.0f, ma, Ma);
    abs_range(s.b0, s.b1, 0.0f, mb, Mb);
    s.C0 = chroma(ma, mb);                                      // sqrtf(a*a + b*b) is monotone in |a|, |b|
    s.C1 = chroma(Ma, Mb);
    const PixelTerms lo = pixel_terms_c(0.0f, 0.0f, 0.0f, s.C0), hi = pixel_terms_c(0.0f, 0.0f, 0.0f, s.C1);
    s.wC0 = hi.wC; s.wC1 = lo.wC;                               // the weights decrease with C
    s.wH0 = hi.wH; s.wH1 = lo.wH;
    s.pad[0] = s.pad[1] = s.pad[2] = s.pad[3] = 0.0f;
    return s;
}

// ---- candidate lists --------------------------------------------------------------------------------------------------------
// lists: 2 x kLabCells records of kListBytes.  Record c: [count][index 0 .. 30], record kLabCells + c: [index 31 .. 62] (written
// for every cell, read only when count > 31); count 255 = scan all centroids (more than kListMax candidates, or a rim cell).
// Every byte behind the entries names a centroid OUTSIDE the list -- its key is above (1 + kMaskSlack) times the best one's, so
// it can neither win nor look like a near-tie --, or index 255 when k < 256, which the pass keeps far away: the pass walks whole
// list words without asking which bytes are entries.
// One wave per cell; lane j (+ 64, 128, 192) bounds centroid j over the cell (key_range, kmg_table_dev.h).
// TWO: the lists of the meld pass -- every centroid that can be one of the two closest: lower bound not above (1 + kMaskSlack)
// times the SECOND smallest upper bound (k >= 2).
// the part of the grid an image can reach (launch_lab_candidates): cells [L0, L0 + nL) x [a0, a0 + nA) x [b0, b0 + nB), all interior
struct LabReach { uint32_t L0, a0, b0, nL, nA, nB; };
constexpr uint32_t kCellsPerWave = 8;

// (m1 <= m2) <- the two smallest of {m1, m2, o1, o2}, o1 <= o2
__device__ __forceinline__ void two_smallest(float &m1, float &m2, float o1, float o2)
{
    const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
    m2 = fminf(hi, fminf(m2, o2));
    m1 = lo;
}

// HALVES = 2 (256 < k <= 512): the centroids 0 .. 255 and 256 .. k - 1 have a list each -- bytes again, index - 256
// in the second --, laid out [first records of half 0][of half 1][continuation records of half 0][of half 1].
__host__ __device__ __forceinline__ size_t list_record_offset(uint32_t halves, uint32_t half, uint32_t part, uint32_t cell)
{
    return ((size_t)(part * halves + half) * kLabCells + cell) * kListBytes;
}

template <bool TWO, int HALVES>
__global__ __launch_bounds__(kBlock) void k_lab_candidates(const Centroid *__restrict__ cent, uint32_t k, LabReach reach,
                                                           uint8_t *__restrict__ lists)
{
    constexpr uint32_t W = 4u * HALVES;
    __shared__ uint8_t s_rec_all[kBlock / 64][HALVES][2 * kListBytes];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // a wave takes kCellsPerWave cells of the reachable box that differ in L only (the rest of the table says "scan everything":
    // the launcher's memset): its centroids stay in registers, and so do the chroma bounds and weights of its (a, b) column
    const uint32_t groups_L = (reach.nL + kCellsPerWave - 1u) / kCellsPerWave;
    const uint32_t wave = blockIdx.x * (kBlock / 64) + wv;
    if (wave >= groups_L * reach.nA * reach.nB) return;
    const uint32_t gL = wave % groups_L, ib = reach.b0 + (wave / groups_L) % reach.nB, ia = reach.a0 + wave / (groups_L * reach.nB);
    CellBounds cb = lab_cell_bounds(reach.L0, ia, ib);
    Centroid c[W];
#pragma unroll
    for (uint32_t w = 0; w < W; ++w) {
        const uint32_t j = w * 64u + lane;
        c[w] = cent[j < k ? j : 0u];
    }
    for (uint32_t u = 0; u < kCellsPerWave; ++u) {
        if (gL * kCellsPerWave + u >= reach.nL) return;
        const uint32_t iL = reach.L0 + gL * kCellsPerWave + u;
        const uint32_t cell = (iL * (uint32_t)kLabNA + ia) * (uint32_t)kLabNA + ib;
        lab_cell_L_bounds(iL, cb);
        float U = 3.0e38f, U2 = 3.0e38f, lo[W];
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) {
            lo[w] = 3.0e38f;
            if (w * 64u + lane < k) {
                const KeyRange r = key_range(cb, c[w].L, c[w].a, c[w].b, c[w].C);
                lo[w] = r.lo;
                if (TWO) two_smallest(U, U2, r.hi, 3.0e38f); else U = fminf(U, r.hi);
            }
        }
        if (TWO) {
            for (int off = 32; off > 0; off >>= 1) {
                const float o1 = __shfl_xor(U, off, 64), o2 = __shfl_xor(U2, off, 64);
                two_smallest(U, U2, o1, o2);
            }
            U = mask_threshold(U2);
        } else {
            U = mask_threshold(wave_min(U));                        // keep what can still be a near-tie (kmg_math.h)
        }
        unsigned long long m[W];
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) m[w] = __ballot(lo[w] <= U);   // (lo = 3e38 beyond k)
        __builtin_amdgcn_wave_barrier();                            // (the previous cell's records have left)
#pragma unroll
        for (uint32_t h = 0; h < (uint32_t)HALVES; ++h) {
            uint8_t *s_rec = s_rec_all[wv][h];
            uint32_t n_cand = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) n_cand += (uint32_t)__builtin_popcountll(m[4u * h + w]);
            // a byte that is no entry: a centroid of this half outside the list, or -- the half has fewer than 256 -- index 255
            uint32_t pad = 255u;
            if (k >= 256u * (h + 1u)) {
#pragma unroll
                for (int w = 3; w >= 0; --w)
                    if (~m[4u * h + w]) pad = (uint32_t)w * 64u + (uint32_t)__builtin_ctzll(~m[4u * h + w]);
            }
            if (lane < 16u) reinterpret_cast<uint32_t *>(s_rec)[lane] = pad * 0x01010101u;
            __builtin_amdgcn_wave_barrier();
            if (n_cand <= kListMax) {
                uint32_t base = 1u;                                 // byte 0 is the count
#pragma unroll
                for (uint32_t w = 0; w < 4u; ++w) {
                    const unsigned long long mw = m[4u * h + w];
                    if ((mw >> lane) & 1ull) s_rec[base + bits_below_lane(mw)] = (uint8_t)(w * 64u + lane);
                    base += (uint32_t)__builtin_popcountll(mw);
                }
            }
            if (lane == 0u) s_rec[0] = (uint8_t)(n_cand <= kListMax ? n_cand : 255u);
            __builtin_amdgcn_wave_barrier();
            const uint4 *src = reinterpret_cast<const uint4 *>(s_rec);
            if (lane < 2u) reinterpret_cast<uint4 *>(lists + list_record_offset(HALVES, h, 0u, cell))[lane] = src[lane];
            else if (lane < 4u) reinterpret_cast<uint4 *>(lists + list_record_offset(HALVES, h, 1u, cell))[lane - 2u] = src[lane];
        }
    }
}

size_t lab_list_bytes(uint32_t k) { return (k > 256u ? 2u : 1u) * kLabListBytes; }

hipError_t launch_lab_candidates(const Centroid *cent, uint32_t k, float threshold, bool two_closest, uint8_t *lists, hipStream_t st)
{
    if (k > kLabListMaxK || (two_closest && k < 2u)) return hipErrorInvalidValue;
    // sRGB in the shader's Lab: L 0 .. 100, a -86.2 .. 98.3, b -107.9 .. 94.5 (one cell of margin), moved by the 16 offsets
    // threshold (0 .. 15) / 16 - threshold / 2; clipped to the interior of the grid (its rim cells are unbounded outwards).
    // Cells outside get no list -- count 255, "scan everything": exact for any pixel that lands there all the same, so only
    // speed depends on this box (two thirds of the grid lie outside it).
    const float t = threshold == threshold ? threshold : 0.0f;
    const float o0 = fminf(t * -0.5f, t * 0.4375f), o1 = fmaxf(t * -0.5f, t * 0.4375f);
    auto lo_cell = [](float x, float origin, int n) { const float c = floorf((x - origin) / kLabStep); return (uint32_t)(c < 1.0f ? 1.0f : (c > (float)(n - 2) ? (float)(n - 2) : c)); };
    LabReach reach;
    reach.L0 = lo_cell(-4.0f + o0, kLabL0, kLabNL);   const uint32_t L1 = lo_cell(104.0f + o1, kLabL0, kLabNL);
    reach.a0 = lo_cell(-91.0f + o0, kLabA0, kLabNA);  const uint32_t a1 = lo_cell(103.0f + o1, kLabA0, kLabNA);
    reach.b0 = lo_cell(-112.0f + o0, kLabA0, kLabNA); const uint32_t b1 = lo_cell(99.0f + o1, kLabA0, kLabNA);
    reach.nL = L1 - reach.L0 + 1u; reach.nA = a1 - reach.a0 + 1u; reach.nB = b1 - reach.b0 + 1u;
    const uint32_t halves = k > 256u ? 2u : 1u;
    hipError_t e = hipMemsetAsync(lists, 0xFF, (size_t)halves * kLabCells * kListBytes, st);     // every first record
    if (e != hipSuccess) return e;
    const uint32_t n_waves = ((reach.nL + kCellsPerWave - 1u) / kCellsPerWave) * reach.nA * reach.nB;
    const dim3 grid((n_waves + kBlock / 64 - 1) / (kBlock / 64));
    if (two_closest && halves == 2u) hipLaunchKernelGGL((k_lab_candidates<true, 2>), grid, dim3(kBlock), 0, st, cent, k, reach, lists);
    else if (two_closest) hipLaunchKernelGGL((k_lab_candidates<true, 1>), grid, dim3(kBlock), 0, st, cent, k, reach, lists);
    else if (halves == 2u) hipLaunchKernelGGL((k_lab_candidates<false, 2>), grid, dim3(kBlock), 0, st, cent, k, reach, lists);
    else hipLaunchKernelGGL((k_lab_candidates<false, 1>), grid, dim3(kBlock), 0, st, cent, k, reach, lists);
    return hipGetLastError();
}

// ---- the pass ---------------------------------------------------------------------------------------------------------------
// With the table in L2 the pass is bound by vector issue (profiles/r03e_apply_pmc.txt: ~350 vector instructions per pixel, the
// vector units busy for the whole kernel), so the list walk is written for instruction count -- 13 per candidate instead of 20:
//   * the candidate's index travels in the low byte of its key (one v_perm_b32, which also extracts it from the list word), so the
//     running minimum and runner-up are one v_min_u32 and one v_med3_u32, and the index needs no register of its own;
//   * the walk is unrolled by list WORDS (the wave leaves it after its longest list's last word): byte positions are compile-time
//     constants (the LDS address of byte P's centroid is one sub-dword shift), the reads of a word are in flight together, and
//     bytes behind a list's last entry need no test (they name a centroid that cannot matter, k_lab_candidates).
// (The key itself stays cie94_key's twelve plain operations: written for the register PAIRS a 16-byte LDS read returns -- v_pk_add /
// v_pk_mul / v_pk_fma_f32 on (dL, da) and (db, dC), eight instructions -- it was measured 6 % SLOWER, 644 against 602 us: packed
// fp32 operations issue at half rate on gfx950, tools/valu_rate.hip.)
// The key orders only: near-ties (kmg_math.h; the packed byte costs 2^-15 = 512u of the 2048u slack) are settled with the
// literal distance as everywhere else.
// (dL^2 + dC^2 wC + max(da^2 + db^2 - dC^2, 0) wH) against the centroid at byte offset j16 of the LDS table
// = cie94_key (kmg_math.h).  The table starts at LDS address 0 -- the kernels check --, so the offset is the address: no add per read
typedef const float4 __attribute__((address_space(3))) *LdsFloat4Ptr;
__device__ __forceinline__ float entry_key(uint32_t j16, const PixelTerms &pp)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 c = *(LdsFloat4Ptr)j16;
#else
    (void)j16;
    const float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // (host pass: never called)
#endif
    const float dL = pp.L - c.x, da = pp.a - c.y, db = pp.b - c.z, dC = pp.C - c.w;
    const float dC2 = dC * dC;
    const float t = fmaf(db, db, da * da);
    const float h = fmaxf(t - dC2, 0.0f);
    return fmaf(h, pp.wH, fmaf(dC2, pp.wC, dL * dL));
}

// byte P of list word wd -> that centroid's key with its index in the low byte
template <int P>
__device__ __forceinline__ uint32_t list_entry_key(uint32_t wd, uint32_t lds0, const PixelTerms &pp)
{
    uint32_t j16;                                                   // (byte P) << 4 in one instruction (sub-dword operand select)
    if (P == 0) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(j16) : "v"(wd));
    else if (P == 1) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(j16) : "v"(wd));
    else if (P == 2) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(j16) : "v"(wd));
    else asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(j16) : "v"(wd));
    const float key = entry_key(j16 + lds0, pp);                     // (lds0: a constant that lands in the read's offset field)
    return __builtin_amdgcn_perm(__float_as_uint(key), wd, 0x07060500u | (uint32_t)P);
}

// best <- min(best, kp), second <- the runner-up (best <= second before and after)
__device__ __forceinline__ void min_and_runner_up(uint32_t kp, uint32_t &best, uint32_t &second)
{
    asm("v_med3_u32 %0, %2, %1, %0\n\tv_min_u32 %1, %1, %2" : "+v"(second), "+v"(best) : "v"(kp));
}

// One half's list of one pixel: the smallest and second smallest packed key over it (kept in best / second).  rec0 / rec1 = the
// first record; more than 31 entries: the continuation record is fetched here.
template <int HALVES>
__device__ __forceinline__ void walk_list(const uint8_t *__restrict__ lists, uint32_t cell, uint32_t half, uint4 rec0, uint4 rec1,
                                          const PixelTerms &pp, uint32_t k, uint32_t &best, uint32_t &second)
{
    const uint32_t cnt_raw = rec0.x & 255u;
    const bool over = cnt_raw == 255u;
    const uint32_t cnt = over ? 0u : cnt_raw;
    uint4 more0 = make_uint4(0u, 0u, 0u, 0u), more1 = more0;        // the continuation record of a long list
    if (cnt >= kListBytes) {
        const uint4 *r = reinterpret_cast<const uint4 *>(lists + list_record_offset(HALVES, half, 1u, cell));
        more0 = r[0]; more1 = r[1];
    }
    // byte t of the list's 64 bytes: [count][index 0 .. 62]; a word the list reaches holds entries and, behind them, bytes
    // that name a centroid outside the list (k_lab_candidates), so all four of its bytes are scanned
    const uint32_t rw[16] = {rec0.x, rec0.y, rec0.z, rec0.w, rec1.x, rec1.y, rec1.z, rec1.w,
                             more0.x, more0.y, more0.z, more0.w, more1.x, more1.y, more1.z, more1.w};
    const uint32_t longest = wave_max_u32_dpp(cnt);                  // the wave's longest list
    const uint32_t lds0 = half * 4096u;                              // this half's 256 table entries
#define KMG_LIST_ENTRY(WORD, P) min_and_runner_up(list_entry_key<P>(rw[WORD], lds0, pp), best, second);
    if (longest) {
        if (cnt) { KMG_LIST_ENTRY(0, 1) KMG_LIST_ENTRY(0, 2) KMG_LIST_ENTRY(0, 3) }
    }
#pragma unroll
    for (int wd = 1; wd < 16; ++wd) {
        if ((uint32_t)(wd * 4) > longest) break;
        if ((uint32_t)(wd * 4) <= cnt) { KMG_LIST_ENTRY(wd, 0) KMG_LIST_ENTRY(wd, 1) KMG_LIST_ENTRY(wd, 2) KMG_LIST_ENTRY(wd, 3) }
    }
#undef KMG_LIST_ENTRY
    if (__ballot(over)) {                                           // no list (too long, or a rim cell): every centroid of the half
        const uint32_t n_half = min(k - half * 256u, 256u);
        for (uint32_t j = 0; j < n_half; ++j) {
            if (over) min_and_runner_up((__float_as_uint(entry_key(lds0 + (j << 4), pp)) & ~255u) | j, best, second);
        }
    }
}

template <int HALVES>
__global__ __launch_bounds__(kBlock) void k_dither_lists(const uint32_t *__restrict__ rgba, uint32_t w, uint64_t n,
                                                         uint32_t row0, const Centroid *__restrict__ cent, uint32_t k,
                                                         const float *__restrict__ lut, const uint32_t *__restrict__ pal,
                                                         float threshold, const uint8_t *__restrict__ lists,
                                                         uint32_t *__restrict__ out, int aligned)
{
    extern __shared__ float4 smem4[];
    constexpr uint32_t kpad = 256u * HALVES;                       // every (half, byte) indexes the table: entries k .. are far away
    float4 *s_cent = smem4;
    // the kernel has no static LDS, so its dynamic LDS -- the centroid table first -- starts at LDS address 0 (entry_key)
    if ((uint32_t)reinterpret_cast<uintptr_t>(s_cent) != 0u) __builtin_trap();
    float *s_lut = reinterpret_cast<float *>(smem4 + kpad);
    float *s_off = s_lut + 256;
    const float sentinel_C = chroma(10000.0f, 10000.0f);
    s_lut[threadIdx.x] = lut[threadIdx.x];
    static_assert(kBlock == 256, "one table entry per thread and half");
#pragma unroll
    for (uint32_t h = 0; h < (uint32_t)HALVES; ++h) {
        const uint32_t j = h * 256u + threadIdx.x;
        float4 c = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);       // key ~ 1e36: far above the sentinel's
        if (j < k) { const Centroid ce = cent[j]; c = make_float4(ce.L, ce.a, ce.b, ce.C); }
        s_cent[j] = c;
    }
    if (threadIdx.x < 16) s_off[threadIdx.x] = threshold * (bayer16(threadIdx.x) / 16.0f - 0.5f);
    __syncthreads();
    constexpr uint64_t TILE = (uint64_t)kBlock * 4;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t i0 = tile * TILE + (uint64_t)threadIdx.x * 4;
        uint32_t px[4];
        load4_stream(rgba, i0, n, aligned != 0, px);
        const uint32_t i32 = (uint32_t)i0;                          // n < 2^32
        uint32_t gy = i32 / w, gx = i32 - gy * w;
        gy += row0;
        float pL[4], pa[4], pb[4];
        uint32_t cell[4];
        uint4 rec[4][2];                                            // the four pixels' lists (first half)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t bi = (gx & 3u) + ((gy & 3u) << 2);
            px_to_lab(s_lut, px[q], pL[q], pa[q], pb[q]);
            const float off = s_off[bi];
            pL[q] = pL[q] + off; pa[q] = pa[q] + off; pb[q] = pb[q] + off;   // mix_colors.wgsl:72
            cell[q] = lab_cell_index(pL[q], pa[q], pb[q]);
            const uint4 *r = reinterpret_cast<const uint4 *>(lists + list_record_offset(HALVES, 0u, 0u, cell[q]));
            rec[q][0] = r[0]; rec[q][1] = r[1];
            gx += 1;
            if (gx == w) { gx = 0; gy += 1; }
        }
        uint32_t res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const PixelTerms pt = pixel_terms_fast(pL[q], pa[q], pb[q], chroma(pa[q], pb[q]));
            // mix_colors.wgsl:73-80 starts the scan at the sentinel (10000, 10000, 10000), index k.  Its key is not computed here: a
            // pixel whose smallest key is not below kSentinelFloor -- (10000 - L)^2 alone is above that for every L < 9000, and no
            // centroid of a palette in Lab is 1000 units from a pixel -- goes to the literal scan below, which starts at the
            // sentinel and is exact for any input.
            constexpr uint32_t kSentinelFloor = 0x49742400u;          // 1.0e6f
            uint32_t best = 0x7F7FFF00u, second = 0x7F7FFFFFu;
            uint4 next0 = make_uint4(0u, 0u, 0u, 0u), next1 = next0;
            if (HALVES == 2) {                                       // the second half's record, in flight during the first walk
                const uint4 *r = reinterpret_cast<const uint4 *>(lists + list_record_offset(HALVES, 1u, 0u, cell[q]));
                next0 = r[0]; next1 = r[1];
            }
            walk_list<HALVES>(lists, cell[q], 0u, rec[q][0], rec[q][1], pt, k, best, second);
            uint32_t idx = best & 255u;
            if (HALVES == 2) {
                // the halves apart, then merged; packed keys that differ in their low byte only are a near-tie (decided below)
                uint32_t best1 = 0x7F7FFF00u, second1 = 0x7F7FFFFFu;
                walk_list<HALVES>(lists, cell[q], 1u, next0, next1, pt, k, best1, second1);
                const bool from1 = best1 < best;
                idx = from1 ? 256u + (best1 & 255u) : idx;
                second = min(max(best, best1), min(second, second1));
                best = min(best, best1);
            }
            const float thr = tie_threshold(__uint_as_float(best));
            const bool near = __uint_as_float(second) <= thr || best >= kSentinelFloor || !(pt.L < 9000.0f);
            if (__ballot(near)) {
                // near-tie (kmg_math.h): mix_colors.wgsl:73-80 with the literal distance, the sentinel first, same order.  Rare:
                // the lists are read again, byte by byte
                float lb = cie94_c(pt.L, pt.a, pt.b, pt.C, 10000.0f, 10000.0f, 10000.0f, sentinel_C);
                uint32_t li = k;
                if (near) {
                    for (uint32_t h = 0; h < (uint32_t)HALVES; ++h) {
                        const uint8_t *first = lists + list_record_offset(HALVES, h, 0u, cell[q]);
                        const uint8_t *cont = lists + list_record_offset(HALVES, h, 1u, cell[q]);
                        const uint32_t cnt = first[0];
                        const uint32_t n_half = min(k - h * 256u, 256u);
                        const uint32_t steps = cnt == 255u ? n_half : cnt;
                        for (uint32_t i = 0; i < steps; ++i) {
                            const uint32_t j = h * 256u + (cnt == 255u ? i : (uint32_t)(i < kListBytes - 1u ? first[1u + i] : cont[i - (kListBytes - 1u)]));
                            const float4 c = s_cent[j];
                            if (cie94_key(pt, c.x, c.y, c.z, c.w) <= thr) {
                                const float d = cie94_c(pt.L, pt.a, pt.b, pt.C, c.x, c.y, c.z, c.w);
                                if (d < lb) { lb = d; li = j; }
                            }
                        }
                    }
                    idx = li;
                }
            }
            res[q] = pal[idx];
        }
        store4_stream(out, i0, n, aligned != 0, res);
    }
}

// (Round 6 measured how much of the walk is padding -- tools/dither_list_lengths.py: the lanes' lists hold 4.4 entries on average,
// the longest of a wave 11-12: lanes' words / (64 x the wave's longest) = 0.50 on the benchmark image -- and built the pass with the
// pixels of a 1024-pixel tile sorted by list length through LDS (counting sort by ballots, the walk over lanes of equal length,
// results back through LDS): byte-identical and SLOWER, 0.92 against 0.67 ms -- four workgroup barriers and a second dependent
// memory round trip (the length, then the record) per tile cost more than half a walk saves.  tools/experiments/
// r06_dither_sorted_by_list_length.patch; profiles/r06_dither_list_lengths.txt.)
hipError_t launch_dither_lists(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0, const Centroid *cent, uint32_t k,
                               const float *lut, const uint32_t *pal, float threshold, const uint8_t *lists, uint32_t *out,
                               hipStream_t st)
{
    if (k > kLabListMaxK) return hipErrorInvalidValue;
    const uint64_t n = (uint64_t)w * rows;
    const uint64_t tiles = (n + kBlock * 4 - 1) / (kBlock * 4);
    const uint32_t grid = (uint32_t)(tiles < 8192 ? (tiles ? tiles : 1) : 8192);
    const uint32_t halves = k > 256u ? 2u : 1u;
    const size_t lds = sizeof(float4) * 256 * halves + (256 + 16) * sizeof(float);
    const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? 1 : 0;
    if (halves == 2u)
        hipLaunchKernelGGL(k_dither_lists<2>, dim3(grid), dim3(kBlock), lds, st, rgba, w, n, row0, cent, k, lut, pal, threshold, lists, out,
                           aligned);
    else
        hipLaunchKernelGGL(k_dither_lists<1>, dim3(grid), dim3(kBlock), lds, st, rgba, w, n, row0, cent, k, lut, pal, threshold, lists, out,
                           aligned);
    return hipGetLastError();
}

#ifdef KMG_TOOLS
// tools build: how much of the list walk's issue time is padding.  The walk of k_dither_lists is paid by a wave for its LONGEST list
// (unrolled by words, lanes with shorter lists masked off); this kernel visits the pixels in k_dither_lists' own lane layout and
// adds up, per wave and pixel slot, the words each lane's list has and 64 x the words the wave walks.
// out: [0] sum of the lanes' words, [1] sum over waves of 64 x the longest, [2] pixels, [3] pixels with count 255 (no list),
// [4 + c] pixels whose list has c entries (c < 64).
__global__ __launch_bounds__(kBlock) void k_dither_list_stats(const uint32_t *__restrict__ rgba, uint32_t w, uint64_t n, uint32_t row0,
                                                              const float *__restrict__ lut, float threshold,
                                                              const uint8_t *__restrict__ lists, unsigned long long *__restrict__ out)
{
    __shared__ float s_lut[256], s_off[16];
    __shared__ unsigned long long s_acc[4 + 64];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    if (threadIdx.x < 16) s_off[threadIdx.x] = threshold * (bayer16(threadIdx.x) / 16.0f - 0.5f);
    if (threadIdx.x < 68) s_acc[threadIdx.x] = 0ull;
    __syncthreads();
    constexpr uint64_t TILE = (uint64_t)kBlock * 4;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t i0 = tile * TILE + (uint64_t)threadIdx.x * 4;
        uint32_t gy = (uint32_t)i0 / w, gx = (uint32_t)i0 - gy * w;
        gy += row0;
        for (int q = 0; q < 4; ++q) {
            const bool in = i0 + (uint64_t)q < n;
            uint32_t cnt = 0;
            if (in) {
                float L, a, b;
                px_to_lab(s_lut, rgba[i0 + q], L, a, b);
                const float off = s_off[(gx & 3u) + ((gy & 3u) << 2)];
                cnt = lists[list_record_offset(1u, 0u, 0u, lab_cell_index(L + off, a + off, b + off))];
            }
            const bool over = cnt == 255u;
            const uint32_t c = over ? 0u : cnt;
            const uint32_t words = c ? 1u + c / 4u : 0u;
            const uint32_t longest = wave_max_u32_dpp(c);
            const uint32_t wave_words = longest ? 1u + longest / 4u : 0u;
            const uint32_t sum = wave_add_u32(in ? words : 0u);
            if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s_acc[0], (unsigned long long)sum); atomicAdd(&s_acc[1], 64ull * wave_words); }
            if (in) { atomicAdd(&s_acc[2], 1ull); if (over) atomicAdd(&s_acc[3], 1ull); else atomicAdd(&s_acc[4 + (c < 63u ? c : 63u)], 1ull); }
            gx += 1;
            if (gx == w) { gx = 0; gy += 1; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 68 && s_acc[threadIdx.x]) atomicAdd(out + threadIdx.x, s_acc[threadIdx.x]);
}

hipError_t launch_dither_list_stats(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0, const float *lut, float threshold,
                                    const uint8_t *lists, unsigned long long *out68, hipStream_t st)
{
    const uint64_t n = (uint64_t)w * rows;
    const uint64_t tiles = (n + kBlock * 4 - 1) / (kBlock * 4);
    hipLaunchKernelGGL(k_dither_list_stats, dim3((uint32_t)(tiles < 4096 ? tiles : 4096)), dim3(kBlock), 0, st, rgba, w, n, row0, lut, threshold, lists, out68);
    return hipGetLastError();
}
#endif

// ---- the meld pass ----------------------------------------------------------------------------------------------------------
// mix_colors.wgsl:29-48 keeps, over an ordered scan with `<`, the two closest centroids of a pixel under the literal distance:
// the two smallest in (distance, index) order, the sentinel (10000, 10000, 10000) filling a slot no centroid takes.  Here the
// walk keeps the THREE smallest packed keys (index in the low byte): when the first two are separated from their successors by
// more than the tie slack, key order is literal order and nothing outside the list is closer (k_lab_candidates<true>), so the
// two winners are known after 13 instructions per candidate instead of the ~45 of a literal distance; their two literal
// distances (the values mix_colors.wgsl:86-89 uses) are then computed once.  Any pixel that fails that test -- or whose second
// key is not far below the sentinel's -- is re-scanned literally over its list, in order, exactly as the reference does.
__device__ __forceinline__ void three_smallest(uint32_t kp, uint32_t &b, uint32_t &s, uint32_t &t)
{
    asm("v_med3_u32 %0, %3, %1, %0\n\tv_med3_u32 %1, %3, %2, %1\n\tv_min_u32 %2, %2, %3" : "+v"(t), "+v"(s), "+v"(b) : "v"(kp));
}

// one half's list of one pixel: its three smallest packed keys (walk_list with three_smallest)
template <int HALVES>
__device__ __forceinline__ void walk_list3(const uint8_t *__restrict__ lists, uint32_t cell, uint32_t half, uint4 rec0, uint4 rec1,
                                           const PixelTerms &pp, uint32_t k, uint32_t &k1, uint32_t &k2, uint32_t &k3)
{
    const uint32_t cnt_raw = rec0.x & 255u;
    const bool over = cnt_raw == 255u;
    const uint32_t cnt = over ? 0u : cnt_raw;
    uint4 more0 = make_uint4(0u, 0u, 0u, 0u), more1 = more0;
    if (cnt >= kListBytes) {
        const uint4 *r = reinterpret_cast<const uint4 *>(lists + list_record_offset(HALVES, half, 1u, cell));
        more0 = r[0]; more1 = r[1];
    }
    const uint32_t rw[16] = {rec0.x, rec0.y, rec0.z, rec0.w, rec1.x, rec1.y, rec1.z, rec1.w,
                             more0.x, more0.y, more0.z, more0.w, more1.x, more1.y, more1.z, more1.w};
    const uint32_t longest = wave_max_u32_dpp(cnt);
    const uint32_t lds0 = half * 4096u;
#define KMG_LIST_ENTRY(WORD, P) three_smallest(list_entry_key<P>(rw[WORD], lds0, pp), k1, k2, k3);
    if (longest) {
        if (cnt) { KMG_LIST_ENTRY(0, 1) KMG_LIST_ENTRY(0, 2) KMG_LIST_ENTRY(0, 3) }
    }
#pragma unroll
    for (int wd = 1; wd < 16; ++wd) {
        if ((uint32_t)(wd * 4) > longest) break;
        if ((uint32_t)(wd * 4) <= cnt) { KMG_LIST_ENTRY(wd, 0) KMG_LIST_ENTRY(wd, 1) KMG_LIST_ENTRY(wd, 2) KMG_LIST_ENTRY(wd, 3) }
    }
#undef KMG_LIST_ENTRY
    if (__ballot(over)) {
        const uint32_t n_half = min(k - half * 256u, 256u);
        for (uint32_t j = 0; j < n_half; ++j) {
            if (over) three_smallest((__float_as_uint(entry_key(lds0 + (j << 4), pp)) & ~255u) | j, k1, k2, k3);
        }
    }
}

template <int HALVES>
__global__ __launch_bounds__(kBlock) void k_meld_lists(const uint32_t *__restrict__ rgba, uint64_t n,
                                                       const Centroid *__restrict__ cent, uint32_t k,
                                                       const float *__restrict__ lut, const uint8_t *__restrict__ lists,
                                                       uint32_t *__restrict__ out, int aligned)
{
    extern __shared__ float4 smem4[];
    float4 *s_cent = smem4;
    if ((uint32_t)reinterpret_cast<uintptr_t>(s_cent) != 0u) __builtin_trap();   // (entry_key reads the table at LDS address 0)
    float *s_lut = reinterpret_cast<float *>(smem4 + 256 * HALVES);
    float *s_thr = s_lut + 256;                                    // the thresholds of the sRGB8 encode (kmg_device.h)
    s_lut[threadIdx.x] = lut[threadIdx.x];
    s_thr[threadIdx.x] = lut[256 + threadIdx.x];
    if (threadIdx.x == 0) s_thr[256] = 3.0e38f;
#pragma unroll
    for (uint32_t h = 0; h < (uint32_t)HALVES; ++h) {
        const uint32_t j = h * 256u + threadIdx.x;
        float4 c = make_float4(1.0e18f, 0.0f, 0.0f, 0.0f);
        if (j < k) { const Centroid ce = cent[j]; c = make_float4(ce.L, ce.a, ce.b, ce.C); }
        s_cent[j] = c;
    }
    __syncthreads();
    const float sentinel_C = chroma(10000.0f, 10000.0f);
    constexpr uint64_t TILE = (uint64_t)kBlock * 4;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t i0 = tile * TILE + (uint64_t)threadIdx.x * 4;
        uint32_t px[4];
        load4_stream(rgba, i0, n, aligned != 0, px);
        float pL[4], pa[4], pb[4];
        uint32_t cell[4];
        uint4 rec[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            px_to_lab(s_lut, px[q], pL[q], pa[q], pb[q]);
            cell[q] = lab_cell_index(pL[q], pa[q], pb[q]);
            const uint4 *r = reinterpret_cast<const uint4 *>(lists + list_record_offset(HALVES, 0u, 0u, cell[q]));
            rec[q][0] = r[0]; rec[q][1] = r[1];
        }
        uint32_t res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float L = pL[q], a = pa[q], b = pb[q];
            const PixelTerms pt = pixel_terms_fast(L, a, b, chroma(a, b));
            constexpr uint32_t kSentinelFloor = 0x49742400u;          // 1.0e6f (see k_dither_lists)
            uint32_t k1 = 0x7F7FFF00u, k2 = 0x7F7FFF00u, k3 = 0x7F7FFFFFu;
            uint4 next0 = make_uint4(0u, 0u, 0u, 0u), next1 = next0;
            if (HALVES == 2) {
                const uint4 *r = reinterpret_cast<const uint4 *>(lists + list_record_offset(HALVES, 1u, 0u, cell[q]));
                next0 = r[0]; next1 = r[1];
            }
            walk_list3<HALVES>(lists, cell[q], 0u, rec[q][0], rec[q][1], pt, k, k1, k2, k3);
            uint32_t i1 = k1 & 255u, i2 = k2 & 255u;                  // the winners' table entries
            if (HALVES == 2) {
                // the three smallest of the two sorted triples; a key of the second half carries index - 256 in its low byte
                uint32_t c1 = 0x7F7FFF00u, c2 = 0x7F7FFF00u, c3 = 0x7F7FFFFFu;
                walk_list3<HALVES>(lists, cell[q], 1u, next0, next1, pt, k, c1, c2, c3);
                const uint32_t m1 = min(k1, c1), m2 = min(min(k2, c2), max(k1, c1));
                const uint32_t m3 = min(min(k3, c3), min(max(k2, c1), max(k1, c2)));
                // (keys of the two halves that are EQUAL are a near-tie: the literal scan below decides, whatever these say)
                i1 = c1 < k1 ? 256u + (c1 & 255u) : (k1 & 255u);
                i2 = (m2 == c1 || m2 == c2) ? 256u + (m2 & 255u) : (m2 & 255u);
                k1 = m1; k2 = m2; k3 = m3;
            }
            // the two winners, unless something is within the tie slack of one of them
            float cL, ca, cb, cC, sL, sa, sb, sC;                     // (.C: the chroma kept in the table -- cie94_c, kmg_math.h)
            {
                const float4 c1 = s_cent[i1], c2 = s_cent[i2];
                cL = c1.x; ca = c1.y; cb = c1.z; cC = c1.w; sL = c2.x; sa = c2.y; sb = c2.z; sC = c2.w;
            }
            const bool near = __uint_as_float(k2) <= tie_threshold(__uint_as_float(k1)) ||
                              __uint_as_float(k3) <= tie_threshold(__uint_as_float(k2)) || k2 >= kSentinelFloor || !(pt.L < 9000.0f);
            if (__ballot(near)) {
                // mix_colors.wgsl:30-41, literally, over the lists in index order (k_meld of kmg_kernels.hip does the same over mask
                // words).  Rare: the lists are read again, byte by byte
                float xL = 10000.0f, xa = 10000.0f, xb = 10000.0f, xC = sentinel_C, yL = 10000.0f, ya = 10000.0f, yb = 10000.0f, yC = sentinel_C;
                float d_closest = cie94_c(L, a, b, pt.C, xL, xa, xb, xC), d_second = d_closest;
                if (near) {
                    for (uint32_t h = 0; h < (uint32_t)HALVES; ++h) {
                        const uint8_t *first = lists + list_record_offset(HALVES, h, 0u, cell[q]);
                        const uint8_t *cont = lists + list_record_offset(HALVES, h, 1u, cell[q]);
                        const uint32_t cnt = first[0];
                        const uint32_t steps = cnt == 255u ? min(k - h * 256u, 256u) : cnt;
                        for (uint32_t i = 0; i < steps; ++i) {
                            const uint32_t j = h * 256u + (cnt == 255u ? i : (uint32_t)(i < kListBytes - 1u ? first[1u + i] : cont[i - (kListBytes - 1u)]));
                            const float4 c = s_cent[j];
                            const float d = cie94_c(L, a, b, pt.C, c.x, c.y, c.z, c.w);
                            if (d < d_closest) {
                                yL = xL; ya = xa; yb = xb; yC = xC; d_second = d_closest;
                                xL = c.x; xa = c.y; xb = c.z; xC = c.w; d_closest = d;
                            } else if (d < d_second) {
                                yL = c.x; ya = c.y; yb = c.z; yC = c.w; d_second = d;
                            }
                        }
                    }
                    cL = xL; ca = xa; cb = xb; cC = xC; sL = yL; sa = ya; sb = yb; sC = yC;
                }
            }
            // :86-89
            const float factor = cie94_c(L, a, b, pt.C, sL, sa, sb, sC) / cie94_c(cL, ca, cb, cC, sL, sa, sb, sC);
            const float oL = factor * cL + (1.0f - factor) * sL;
            const float oa = factor * ca + (1.0f - factor) * sa;
            const float ob = factor * cb + (1.0f - factor) * sb;
            res[q] = lab_to_rgba8_dev<true>(oL, oa, ob, s_thr);
        }
        store4_stream(out, i0, n, aligned != 0, res);
    }
}

hipError_t launch_meld_lists(const uint32_t *rgba, uint64_t n, const Centroid *cent, uint32_t k, const float *lut,
                             const uint8_t *lists, uint32_t *out, hipStream_t st)
{
    if (k < 2u || k > kLabListMaxK) return hipErrorInvalidValue;
    const uint64_t tiles = (n + kBlock * 4 - 1) / (kBlock * 4);
    const uint32_t grid = (uint32_t)(tiles < 8192 ? (tiles ? tiles : 1) : 8192);
    const uint32_t halves = k > 256u ? 2u : 1u;
    const size_t lds = sizeof(float4) * 256 * halves + (256 + 257) * sizeof(float);
    const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? 1 : 0;
    if (halves == 2u) hipLaunchKernelGGL(k_meld_lists<2>, dim3(grid), dim3(kBlock), lds, st, rgba, n, cent, k, lut, lists, out, aligned);
    else hipLaunchKernelGGL(k_meld_lists<1>, dim3(grid), dim3(kBlock), lds, st, rgba, n, cent, k, lut, lists, out, aligned);
    return hipGetLastError();
}

// ---- test support -----------------------------------------------------------------------------------------------------------
// violations += #(colour, Bayer index) whose brute-force dither arg-min (literal distance, sentinel included) differs from the
// arg-min over the list of its Lab cell; one workgroup per RGB cell as in the other exhaustive checks
__global__ __launch_bounds__(kBlock) void k_check_lab_lists(const Centroid *__restrict__ cent, uint32_t k,
                                                            const uint8_t *__restrict__ lists, const float *__restrict__ lut,
                                                            float threshold, unsigned long long *__restrict__ violations)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t rgb_cell = blockIdx.x;
    const float sentinel_C = chroma(10000.0f, 10000.0f);
    unsigned long long bad = 0;
    for (uint32_t c = threadIdx.x; c < kCellColours; c += kBlock) {
        float L0, a0, b0;
        colour_to_lab(s_lut, rgb_cell * kCellColours + c, L0, a0, b0);
        for (uint32_t bi = 0; bi < 16u; ++bi) {
            const float off = threshold * (bayer16(bi) / 16.0f - 0.5f);
            const PixelTerms pt = pixel_terms(L0 + off, a0 + off, b0 + off);
            const uint32_t cell = lab_cell_index(pt.L, pt.a, pt.b);
            const uint32_t halves = k > 256u ? 2u : 1u;
            const float start = cie94_c(pt.L, pt.a, pt.b, pt.C, 10000.0f, 10000.0f, 10000.0f, sentinel_C);
            float best = start, bestp = start;
            uint32_t idx = k, idxp = k;
            for (uint32_t j = 0; j < k; ++j) {
                const Centroid ce = cent[j];
                const float d = cie94_c(pt.L, pt.a, pt.b, pt.C, ce.L, ce.a, ce.b, ce.C);
                if (d < best) { best = d; idx = j; }
            }
            for (uint32_t h = 0; h < halves; ++h) {
                const uint8_t *rec = lists + list_record_offset(halves, h, 0u, cell), *more = lists + list_record_offset(halves, h, 1u, cell);
                const uint32_t cnt = rec[0];
                const uint32_t steps = cnt == 255u ? min(k - h * 256u, 256u) : cnt;    // 255: the pass scans the whole half
                for (uint32_t i = 0; i < steps; ++i) {
                    const uint32_t j = h * 256u + (cnt == 255u ? i : (uint32_t)(i < kListBytes - 1u ? rec[1u + i] : more[i - (kListBytes - 1u)]));
                    const Centroid ce = cent[j];
                    const float d = cie94_c(pt.L, pt.a, pt.b, pt.C, ce.L, ce.a, ce.b, ce.C);
                    if (d < bestp) { bestp = d; idxp = j; }
                }
            }
            bad += idx != idxp;
        }
    }
    if (bad) atomicAdd(violations, bad);
}

// violations += #colours whose two closest centroids (mix_colors.wgsl:29-48, literal distance, sentinel included) differ
// between the scan of all centroids and the scan of the list of the colour's Lab cell
__global__ __launch_bounds__(kBlock) void k_check_lab_lists_two(const Centroid *__restrict__ cent, uint32_t k,
                                                                const uint8_t *__restrict__ lists, const float *__restrict__ lut,
                                                                unsigned long long *__restrict__ violations)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t rgb_cell = blockIdx.x;
    unsigned long long bad = 0;
    for (uint32_t c = threadIdx.x; c < kCellColours; c += kBlock) {
        float L, a, b;
        colour_to_lab(s_lut, rgb_cell * kCellColours + c, L, a, b);
        const uint32_t cell = lab_cell_index(L, a, b);
        const uint32_t halves = k > 256u ? 2u : 1u;
        const float d0 = cie94(L, a, b, 10000.0f, 10000.0f, 10000.0f);
        float dc = d0, ds = d0, pc = d0, ps = d0;
        uint32_t ic = k, is = k, jc = k, js = k;
        for (uint32_t j = 0; j < k; ++j) {
            const Centroid ce = cent[j];
            const float d = cie94(L, a, b, ce.L, ce.a, ce.b);
            if (d < dc) { ds = dc; is = ic; dc = d; ic = j; } else if (d < ds) { ds = d; is = j; }
        }
        for (uint32_t h = 0; h < halves; ++h) {
            const uint8_t *rec = lists + list_record_offset(halves, h, 0u, cell), *more = lists + list_record_offset(halves, h, 1u, cell);
            const uint32_t cnt = rec[0];
            const uint32_t steps = cnt == 255u ? min(k - h * 256u, 256u) : cnt;        // 255: the pass scans the whole half
            for (uint32_t i = 0; i < steps; ++i) {
                const uint32_t j = h * 256u + (cnt == 255u ? i : (uint32_t)(i < kListBytes - 1u ? rec[1u + i] : more[i - (kListBytes - 1u)]));
                const Centroid ce = cent[j];
                const float d = cie94(L, a, b, ce.L, ce.a, ce.b);
                if (d < pc) { ps = pc; js = jc; pc = d; jc = j; } else if (d < ps) { ps = d; js = j; }
            }
        }
        bad += (ic != jc) || (is != js);
    }
    if (bad) atomicAdd(violations, bad);
}

hipError_t launch_check_lab_lists_two(const Centroid *cent, uint32_t k, const uint8_t *lists, const float *lut,
                                      unsigned long long *violations, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_lab_lists_two, dim3(kCells), dim3(kBlock), 0, st, cent, k, lists, lut, violations);
    return hipGetLastError();
}

hipError_t launch_check_lab_lists(const Centroid *cent, uint32_t k, const uint8_t *lists, const float *lut, float threshold,
                                  unsigned long long *violations, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_lab_lists, dim3(kCells), dim3(kBlock), 0, st, cent, k, lists, lut, threshold, violations);
    return hipGetLastError();
}

}  // namespace kmg
