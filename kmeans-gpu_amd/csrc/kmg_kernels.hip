// kmg_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Lloyd hot path.
//
// Data layout in HBM
//   pixels     : RGBA8, 4 B/px, row-major, read with 16-byte loads (4 px), 8 px per thread
//   labels     : u32 per pixel (find_centroid.wgsl:43 writes an r32uint texel per pixel)
//   centroids  : k x (L, a, b, C) f32 -- staged into LDS per workgroup, broadcast to the waves
//   partials   : [workgroup][k][4] int64 -- per-workgroup exact sums, no float atomics
//   acc        : [k][4] int64
// Lab is never materialised: every pass re-derives it from the 4-byte pixel (the sRGB decode is a
// 256-entry LDS table), so a Lloyd iteration moves 4 B/px in and 4 B/px out.
//
// Compile with -ffp-contract=off: arithmetic must match kmg_math.h operation for operation.

#include "kmg_internal.h"
#include <mutex>

#include "kmg_kernels.h"

#include <stdlib.h>
#include "kmg_device.h"
#include "kmg_table_dev.h"       // lane_value, wave_min, wave_max_u32_dpp

namespace kmg {



// ------------------------------------------------------------------------------------------
// RGBA8 -> Lab  (rgb_to_lab.wgsl:66-80).  Only used by tests / callers that want Lab itself.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_rgb_to_lab(const uint32_t *__restrict__ rgba, uint64_t n,
                                                       const float *__restrict__ lut,
                                                       float *__restrict__ lab3)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float L, a, b;
        px_to_lab(s_lut, rgba[i], L, a, b);
        lab3[3 * i] = L; lab3[3 * i + 1] = a; lab3[3 * i + 2] = b;
    }
}

hipError_t launch_rgb_to_lab(const uint32_t *rgba, uint64_t n, const float *lut, float *lab3,
                             hipStream_t st)
{
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    uint32_t grid = (uint32_t)(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
    hipLaunchKernelGGL(k_rgb_to_lab, dim3(grid), dim3(kBlock), 0, st, rgba, n, lut, lab3);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Arg-min scan shared by the assign and the output kernels.
//
// The centroid table lives in LDS as (L, a, b, C) float4 entries and is read with ONE
// ds_read_b128 per centroid per wave (all lanes read the same address -> broadcast), so the
// centroid operands arrive in VGPRs.  (Scalar loads would be free, but on gfx950 a VALU
// instruction with an SGPR source issues at half rate -- tools/valu_rate.hip.)
//
// CHUNKED: centroids are visited four at a time; only the chunk minimum is compared against the
// running best (v_min3 + v_min + v_cmp + 2 v_cndmask per 4 pairs instead of 3 per pair) and the
// winning index inside the chunk is recovered afterwards by re-evaluating that one chunk.
// Strict '<' between chunk minima keeps the earliest chunk, the recovery takes the first entry
// that attains the minimum: exactly find_centroid.wgsl's "first minimum wins".
// The table is padded to a multiple of 4 with entries no pixel can be closest to.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;

// Near-tie repair (kmg_math.h): the scan orders by the key and remembers the second smallest key; a pixel whose
// second smallest key is within the tie threshold of the smallest is decided again by the LITERAL distance among
// the centroids inside the threshold, first minimum winning -- find_centroid.wgsl:32-41 / mix_colors.wgsl:73-80
// exactly.  SENTINEL (dither): the running minimum starts at the distance to vec3(10000.0), index k.
template <int PPT, bool CHUNKED, bool SENTINEL>
__device__ __forceinline__ void argmin_scan(const PixelTerms (&pt)[PPT], float (&best)[PPT],
                                            uint32_t (&idx)[PPT], const float4 *s_cent, uint32_t k,
                                            uint32_t kpad)
{
    float second[PPT];                                   // best <= second: the two smallest keys met so far
#pragma unroll
    for (int p = 0; p < PPT; ++p) second[p] = 3.0e38f;
    if (!CHUNKED) {
#pragma unroll 2
        for (uint32_t j = 0; j < k; ++j) {
            const float4 c = s_cent[j];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                float d = cie94_key(pt[p], c.x, c.y, c.z, c.w);
                bool lt = d < best[p];
                if (kLiteralArgmin) second[p] = __builtin_amdgcn_fmed3f(d, best[p], second[p]);
                best[p] = lt ? d : best[p];
                idx[p] = lt ? j : idx[p];
            }
        }
    } else {
        uint32_t chunk[PPT];
#pragma unroll
        for (int p = 0; p < PPT; ++p) chunk[p] = kNoChunk;
        for (uint32_t j = 0; j < kpad; j += 4) {
            const float4 c0 = s_cent[j], c1 = s_cent[j + 1], c2 = s_cent[j + 2], c3 = s_cent[j + 3];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                float d0 = cie94_key(pt[p], c0.x, c0.y, c0.z, c0.w);
                float d1 = cie94_key(pt[p], c1.x, c1.y, c1.z, c1.w);
                float d2 = cie94_key(pt[p], c2.x, c2.y, c2.z, c2.w);
                float d3 = cie94_key(pt[p], c3.x, c3.y, c3.z, c3.w);
                float m = fminf(fminf(fminf(d0, d1), d2), d3);
                bool lt = m < best[p];
                // second smallest chunk minimum; the other keys of the winning chunk are looked at below
                if (kLiteralArgmin) second[p] = __builtin_amdgcn_fmed3f(m, best[p], second[p]);
                best[p] = lt ? m : best[p];
                chunk[p] = lt ? j : chunk[p];
            }
        }
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            if (chunk[p] != kNoChunk) {
                uint32_t found = chunk[p] + 3;
                float d4[4];
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    const float4 c = s_cent[chunk[p] + q];
                    d4[q] = cie94_key(pt[p], c.x, c.y, c.z, c.w);
                    if (q < 3) found = (d4[q] <= best[p]) ? chunk[p] + q : found;
                }
                idx[p] = found;
                if (kLiteralArgmin) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        second[p] = fminf(second[p], chunk[p] + q == found ? 3.0e38f : d4[q]);
                }
            }
        }
    }
    if (kLiteralArgmin) {
        float thr[PPT];
        bool near_any = false;
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            thr[p] = tie_threshold(best[p]);
            near_any = near_any || second[p] <= thr[p];
        }
        if (__ballot(near_any)) {
            // One near-tie pixel at a time, the WAVE working on it: the pixel's terms go to every lane, lane l takes the
            // centroids l, l + 64, ... -- the literal distance where the key is inside the threshold, smallest index first --
            // and the wave reduces to (smallest distance, lowest index among equals) = what the reference's ordered scan with
            // strict `<` returns.  The cost follows the number of such pixels (a few per thousand): the loop this replaces
            // walked the whole table with all lanes for every wave that held one -- 40 % of the vector work of the kernel at
            // 4096^2, k = 16 (512 pixels per wave: nearly every wave), and on a small image the launch's duration (46 us at
            // k = 256 for the wave that met one; profiles/NOTES.md, round 4).
            const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                unsigned long long todo = __ballot(second[p] <= thr[p]);
                while (todo) {
                    const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    PixelTerms q;
                    q.L = lane_value(pt[p].L, src); q.a = lane_value(pt[p].a, src); q.b = lane_value(pt[p].b, src);
                    q.C = lane_value(pt[p].C, src); q.wC = lane_value(pt[p].wC, src); q.wH = lane_value(pt[p].wH, src);
                    const float t = lane_value(thr[p], src);
                    float my_d = 3.0e38f;
                    uint32_t my_j = 0xFFFFFFFFu;
                    for (uint32_t j = lane; j < k; j += 64u) {
                        const float4 c = s_cent[j];
                        if (cie94_key(q, c.x, c.y, c.z, c.w) <= t) {
                            const float d = cie94_c(q.L, q.a, q.b, q.C, c.x, c.y, c.z, c.w);
                            if (d < my_d) { my_d = d; my_j = j; }
                        }
                    }
                    const float wd = wave_min(my_d);
                    const uint32_t wj = ~wave_max_u32_dpp(~(my_d == wd ? my_j : 0xFFFFFFFFu));
                    // (find_centroid.wgsl:29-30 / mix_colors.wgsl:73-75: what the scan starts from)
                    const float start = SENTINEL ? cie94_c(q.L, q.a, q.b, q.C, 10000.0f, 10000.0f, 10000.0f, chroma(10000.0f, 10000.0f)) : 100000.0f;
                    const uint32_t res = wd < start ? wj : (SENTINEL ? k : 0u);
                    if (lane == src) idx[p] = res;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// assign (+ accumulate): find_centroid.wgsl:15-44 fused with the masked sums of
// choose_centroid.wgsl:97-104.  One workgroup = 4 waves; each thread owns PPT pixels (groups of
// 4 consecutive pixels = one 16-byte load, the groups of a tile interleaved so every load
// instruction is fully coalesced).  Sums go to per-workgroup int64 bins in LDS (ds_add_u64),
// flushed once per workgroup into the partial-sum slab.
// PPT = 8 on large images; 4 / 2 / 1 (groups of PPT consecutive pixels) on small ones (assign_ppt): the reference's default
// call shrinks every image to <= 256 x 256 (structures.rs:67-89), and 65 536 pixels at 8 per thread are 32 workgroups for 256
// CUs -- k_assign took 97 us per Lloyd iteration of such an image at k = 256, 18 x that per default call.
// LDS: [centroids kpad x 16 B][bins k x 32 B (ACCUM)][sRGB table 1 KiB]
// ------------------------------------------------------------------------------------------
// LOOP (small images, kmg_lloyd_run): ONE launch per Lloyd iteration (modules.rs:769-800: update, then re-assign).  Every
// workgroup first does the update itself -- choose_centroid.wgsl:180-206 `pick` for all clusters, from the previous launch's sums
// in loop.acc_in, on its LDS copy of the centroids (workgroup 0 also writes them to loop.cent_out and the convergence count to
// loop.n_converged) -- then assigns, and adds its bins to loop.acc_out with atomics (integers: any order); workgroup 0 clears
// loop.acc_clear for the launch after.  Three sum buffers and two centroid buffers in rotation, so that no launch reads what
// it or a concurrent workgroup writes: the launch boundary is the only synchronisation.
struct AssignLoop {
    const int64_t *acc_in = nullptr;     // sums of the previous launch's assignment (do_update)
    int64_t *acc_out = nullptr;          // zero on entry
    int64_t *acc_clear = nullptr;        // cleared for the next launch
    Centroid *cent_out = nullptr;        // the updated centroids (do_update); the launch reads `cent`
    uint32_t *n_converged = nullptr;
    float convergence = 0.0f;
    int do_update = 0;
};

template <int PPT, bool ACCUM, bool CHUNKED, bool LOOP = false>
__global__ __launch_bounds__(kBlock) void k_assign(const uint32_t *__restrict__ rgba, uint64_t n,
                                                   const Centroid *__restrict__ cent, uint32_t k,
                                                   const float *__restrict__ lut,
                                                   uint32_t *__restrict__ labels,
                                                   int64_t *__restrict__ partials, int aligned, AssignLoop loop)
{
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 3u) & ~3u;
    float4 *s_cent = smem4;
    unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem4 + kpad);
    float *s_lut = reinterpret_cast<float *>(bins + (ACCUM ? 4ull * k : 0ull));

    s_lut[threadIdx.x] = lut[threadIdx.x];
    stage_centroids(s_cent, cent, k, kpad);
    if (LOOP && loop.do_update) {
        __shared__ uint32_t s_conv;
        const long long *sums = reinterpret_cast<const long long *>(bins);
        for (uint32_t i = threadIdx.x; i < 4 * k; i += kBlock) bins[i] = (unsigned long long)loop.acc_in[i];
        if (threadIdx.x == 0) s_conv = 0u;
        __syncthreads();
        // (update_centroids of kmg_device.h, operation for operation, on the LDS copy)
        uint32_t mine = 0;
        for (uint32_t c = threadIdx.x; c < k; c += kBlock) {
            const float4 prev = s_cent[c];
            float4 now = prev;
            const long long count = sums[4ull * c + 3];
            if (count > 0) {                                         // choose_centroid.wgsl:185
                float nw[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double mean = ((double)sums[4ull * c + j] / (double)count) * (1.0 / 1048576.0);
                    nw[j] = (float)mean;                             // :186
                }
                now = make_float4(nw[0], nw[1], nw[2], chroma(nw[1], nw[2]));
                if (cie94(nw[0], nw[1], nw[2], prev.x, prev.y, prev.z) < loop.convergence) mine += 1;   // :191
            }                                                        // :192-194 empty: unchanged, 0
            s_cent[c] = now;
            if (blockIdx.x == 0) { Centroid o; o.L = now.x; o.a = now.y; o.b = now.z; o.C = now.w; loop.cent_out[c] = o; }
        }
        if (mine) atomicAdd(&s_conv, mine);
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) *loop.n_converged = s_conv;   // :196-202
    }
    if (LOOP && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < 4 * k; i += kBlock) loop.acc_clear[i] = 0;
    if (ACCUM)
        for (uint32_t i = threadIdx.x; i < 4 * k; i += kBlock) bins[i] = 0ull;
    __syncthreads();

    constexpr int G = PPT < 4 ? PPT : 4;                   // consecutive pixels per group
    constexpr int GROUPS = PPT / G;
    constexpr uint64_t TILE = (uint64_t)kBlock * PPT;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint64_t i0[GROUPS];
        float L[PPT], A[PPT], B[PPT];
        PixelTerms pt[PPT];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            i0[g] = tile * TILE + (uint64_t)g * (kBlock * G) + (uint64_t)threadIdx.x * G;
            uint32_t px[4];
            if (G == 4) {
                load4(rgba, i0[g], n, aligned != 0, px);
            } else {
#pragma unroll
                for (int q = 0; q < G; ++q) px[q] = (i0[g] + q < n) ? rgba[i0[g] + q] : 0u;
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int p = g * G + q;
                px_to_lab(s_lut, px[q], L[p], A[p], B[p]);
                pt[p] = pixel_terms(L[p], A[p], B[p]);
            }
        }

        // find_centroid.wgsl:29-41: min_distance = 100000.0 (squared here), found_index = 0,
        // strict '<' so the first minimum wins.
        float best[PPT];
        uint32_t idx[PPT];
#pragma unroll
        for (int p = 0; p < PPT; ++p) { best[p] = 1.0e10f; idx[p] = 0u; }
        argmin_scan<PPT, CHUNKED, false>(pt, best, idx, s_cent, k, kpad);

#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (labels) {
                if (G == 4) {
                    uint32_t v[4] = {idx[g * 4], idx[g * 4 + 1], idx[g * 4 + 2], idx[g * 4 + 3]};
                    store4(labels, i0[g], n, aligned != 0, v);
                } else {
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (i0[g] + q < n) labels[i0[g] + q] = idx[g * G + q];
                }
            }
            if (ACCUM) {
                // the consecutive pixels of a group often share a label (always, in flat regions of a
                // real image): merge such runs in registers and touch the LDS bins once per run
                long long rs[4] = {0, 0, 0, 0};
                uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int p = g * G + q;
                    if (i0[g] + q < n) {
                        if (idx[p] != cur) {
                            if (cur != 0xFFFFFFFFu) {
                                unsigned long long *bin = bins + 4ull * cur;
                                for (int c = 0; c < 4; ++c) atomicAdd(bin + c, (unsigned long long)rs[c]);
                            }
                            cur = idx[p];
                            rs[0] = rs[1] = rs[2] = rs[3] = 0;
                        }
                        rs[0] += (long long)lab_fix(L[p]);
                        rs[1] += (long long)lab_fix(A[p]);
                        rs[2] += (long long)lab_fix(B[p]);
                        rs[3] += 1;
                    }
                }
                if (cur != 0xFFFFFFFFu) {
                    unsigned long long *bin = bins + 4ull * cur;
                    for (int c = 0; c < 4; ++c) atomicAdd(bin + c, (unsigned long long)rs[c]);
                }
            }
        }
    }

    if (ACCUM && LOOP) {
        __syncthreads();
        unsigned long long *to = reinterpret_cast<unsigned long long *>(loop.acc_out);
        for (uint32_t i = threadIdx.x; i < 4 * k; i += kBlock) {
            const unsigned long long v = bins[i];
            if (v) atomicAdd(to + i, v);                             // (only the clusters this workgroup met)
        }
    } else if (ACCUM) {
        __syncthreads();
        unsigned long long *row = reinterpret_cast<unsigned long long *>(partials) +
                                  (uint64_t)blockIdx.x * 4ull * k;
        for (uint32_t i = threadIdx.x; i < 4 * k; i += kBlock) row[i] = bins[i];
    }
}

constexpr int kAssignPPT = 8;   // pixels per thread of the assign / output kernels on large images

// pixels per thread of k_assign: 8 once the image fills the device at that (4 waves per SIMD = 2^21 pixels), fewer below
// (measured, MI355X, k = 256: tools/default_reduce_probe.py and profiles/NOTES.md round 4)
static int assign_ppt(uint64_t n)
{
    static const int forced = tools_env_int(KMG_TOOLS_ENV("KMG_ASSIGN_PPT"), 0);      // tools build only
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    return n >= (1ull << 21) ? 8 : (n >= (1ull << 20) ? 4 : (n >= (1ull << 19) ? 2 : 1));
}

const DeviceInfo &device_info()
{
    constexpr int kMaxOrdinals = 64;
    static DeviceInfo info[kMaxOrdinals];
    static std::once_flag once[kMaxOrdinals];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxOrdinals) dev = 0;
    std::call_once(once[dev], [dev] {
        int cus = 0, lds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;   // MI355X
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds <= 0) lds = 65536;
        info[dev].cus = (uint32_t)cus;
        info[dev].lds_max = (size_t)lds;
    });
    return info[dev];
}

uint32_t assign_grid(uint64_t n)
{
    const uint64_t tile = (uint64_t)kBlock * (uint64_t)assign_ppt(n);
    uint64_t tiles = (n + tile - 1) / tile;
    if (tiles < 1) tiles = 1;
    return (uint32_t)(tiles < 2048 ? tiles : 2048);
}

hipError_t launch_assign(const uint32_t *rgba, uint64_t n, const Centroid *cent, uint32_t k,
                         const float *lut, uint32_t *labels, int64_t *partials, hipStream_t st)
{
    const uint32_t grid = assign_grid(n);
    const uint32_t kpad = (k + 3u) & ~3u;
    const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 &&
                         (labels == nullptr || (reinterpret_cast<uintptr_t>(labels) & 15u) == 0))
                            ? 1 : 0;
    const bool chunked = k >= 32;   // the recovery step costs ~4 extra pairs per pixel
    size_t lds = sizeof(float4) * kpad + 256 * sizeof(float);
    if (partials) lds += sizeof(unsigned long long) * 4ull * k;
#define KMG_ASSIGN(P, A, C)                                                                                    \
    hipLaunchKernelGGL((k_assign<P, A, C>), dim3(grid), dim3(kBlock), lds, st, rgba, n, cent, k, lut, labels, partials, aligned, AssignLoop())
#define KMG_ASSIGN_P(P)                                                                                        \
    do {                                                                                                       \
        if (partials) { if (chunked) KMG_ASSIGN(P, true, true); else KMG_ASSIGN(P, true, false); }             \
        else          { if (chunked) KMG_ASSIGN(P, false, true); else KMG_ASSIGN(P, false, false); }           \
    } while (0)
    switch (assign_ppt(n)) {
    case 1: KMG_ASSIGN_P(1); break;
    case 2: KMG_ASSIGN_P(2); break;
    case 4: KMG_ASSIGN_P(4); break;
    default: KMG_ASSIGN_P(kAssignPPT); break;
    }
#undef KMG_ASSIGN_P
#undef KMG_ASSIGN
    return hipGetLastError();
}

bool assign_loop_fits(uint64_t n) { return assign_ppt(n) == 1; }

size_t assign_loop_scratch_bytes(uint32_t k) { return sizeof(int64_t) * 12ull * k + sizeof(Centroid) * k; }

hipError_t launch_assign_loop(const uint32_t *rgba, uint64_t n, const Centroid *cent, Centroid *cent_out, uint32_t k, const float *lut,
                              uint32_t *labels, const int64_t *acc_in, int64_t *acc_out, int64_t *acc_clear, int do_update,
                              float convergence, uint32_t *n_converged, hipStream_t st)
{
    const uint64_t tiles = (n + kBlock - 1) / kBlock;
    const uint32_t grid = (uint32_t)(tiles < 2048 ? (tiles ? tiles : 1) : 2048);
    const uint32_t kpad = (k + 3u) & ~3u;
    const int aligned = 0;                                  // (one pixel per thread: scalar loads and stores)
    const size_t lds = sizeof(float4) * kpad + 256 * sizeof(float) + sizeof(unsigned long long) * 4ull * k;
    AssignLoop loop;
    loop.acc_in = acc_in; loop.acc_out = acc_out; loop.acc_clear = acc_clear; loop.cent_out = cent_out;
    loop.n_converged = n_converged; loop.convergence = convergence; loop.do_update = do_update;
    int64_t *no_partials = nullptr;
    if (k >= 32)
        hipLaunchKernelGGL((k_assign<1, true, true, true>), dim3(grid), dim3(kBlock), lds, st, rgba, n, cent, k, lut, labels, no_partials, aligned, loop);
    else
        hipLaunchKernelGGL((k_assign<1, true, false, true>), dim3(grid), dim3(kBlock), lds, st, rgba, n, cent, k, lut, labels, no_partials, aligned, loop);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// acc[c][0..3] = sum over workgroup rows of partials[row][c][0..3].  One workgroup per cluster.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_reduce_partials(const int64_t *__restrict__ partials,
                                                            uint32_t rows, uint32_t k,
                                                            int64_t *__restrict__ acc)
{
    __shared__ long long s[4][kBlock / 64];
    const uint32_t c = blockIdx.x;
    long long v[4] = {0, 0, 0, 0};
    for (uint32_t r = threadIdx.x; r < rows; r += kBlock) {
        const long long *src = reinterpret_cast<const long long *>(partials) + ((uint64_t)r * k + c) * 4ull;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += src[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_down(v[j], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int j = 0; j < 4; ++j) s[j][wave] = v[j];
    __syncthreads();
    if (threadIdx.x < 4) {
        long long t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += s[threadIdx.x][w];
        acc[4ull * c + threadIdx.x] = t;
    }
}

hipError_t launch_reduce_partials(const int64_t *partials, uint32_t rows, uint32_t k, int64_t *acc,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(k_reduce_partials, dim3(k), dim3(kBlock), 0, st, partials, rows, k, acc);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// choose_centroid.wgsl:180-206 `pick`, all clusters in one launch.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_update(const int64_t *__restrict__ acc, uint32_t k,
                                                   float convergence, Centroid *__restrict__ cent,
                                                   uint32_t *__restrict__ n_converged)
{
    __shared__ uint32_t s_count;
    update_centroids(acc, k, convergence, cent, n_converged, &s_count, kBlock);
}

hipError_t launch_update(const int64_t *acc, uint32_t k, float convergence, Centroid *cent,
                         uint32_t *n_converged, hipStream_t st)
{
    hipLaunchKernelGGL(k_update, dim3(1), dim3(kBlock), 0, st, acc, k, convergence, cent, n_converged);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Small slabs (a small image, few clusters: rows x k x 4 <= kReduceUpdateValues): the reduction of the partial rows and the
// update in ONE launch of one workgroup -- thread t sums value t % 4k of the rows t / 4k, t / 4k + groups, ...; the sums stay
// in LDS for update_centroids.  On a 256-pixel working image an iteration is launch-bound (assign 4.4 + reduce 2.5 + update
// 2.2 us at k = 8): one launch less per iteration.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kReduceUpdateBlock = 1024;
constexpr uint32_t kReduceUpdateValues = 32768;

bool reduce_update_fits(uint32_t rows, uint32_t k) { return 4u * k <= kReduceUpdateBlock && (uint64_t)rows * 4u * k <= kReduceUpdateValues; }

__global__ __launch_bounds__(kReduceUpdateBlock) void k_reduce_update(const int64_t *__restrict__ partials, uint32_t rows, uint32_t k,
                                                                      int64_t *__restrict__ acc, int do_update, float convergence,
                                                                      Centroid *__restrict__ cent, uint32_t *__restrict__ n_converged)
{
    __shared__ long long s_part[kReduceUpdateBlock];
    __shared__ long long s_acc[kReduceUpdateBlock];
    __shared__ uint32_t s_count;
    const uint32_t vals = 4u * k, groups = kReduceUpdateBlock / vals;
    const uint32_t v = threadIdx.x % vals, g = threadIdx.x / vals;
    long long sum = 0;
    if (g < groups)
        for (uint32_t r = g; r < rows; r += groups) sum += partials[(uint64_t)r * vals + v];
    s_part[threadIdx.x] = g < groups ? sum : 0;
    __syncthreads();
    if (threadIdx.x < vals) {
        long long t = 0;
        for (uint32_t q = 0; q < groups; ++q) t += s_part[q * vals + threadIdx.x];
        s_acc[threadIdx.x] = t;
        acc[threadIdx.x] = t;
    }
    __syncthreads();
    if (do_update) update_centroids(reinterpret_cast<const int64_t *>(s_acc), k, convergence, cent, n_converged, &s_count, kReduceUpdateBlock);
}

hipError_t launch_reduce_update(const int64_t *partials, uint32_t rows, uint32_t k, int64_t *acc, int do_update, float convergence,
                                Centroid *cent, uint32_t *n_converged, hipStream_t st)
{
    hipLaunchKernelGGL(k_reduce_update, dim3(1), dim3(kReduceUpdateBlock), 0, st, partials, rows, k, acc, do_update, convergence,
                       cent, n_converged);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Farthest-point init (plus_plus_init.wgsl, kmeans++_calc_diff.wgsl).
// The arg-max tie rule of the reference's scan (earliest maximum inside a thread's 16 pixels,
// latest thread across threads) is encoded in a 64-bit key whose maximum is the winner:
//   [ distance bits : 32 ][ index / 16 : 28 ][ 15 - index % 16 : 4 ]
// A key whose distance is 0 means "every distance was 0" -> Candidate(0, 0.0) -> index 0.
// ------------------------------------------------------------------------------------------
__global__ void k_init_first(const uint32_t *__restrict__ rgba, uint64_t index,
                             const float *__restrict__ lut, Centroid *__restrict__ cent,
                             unsigned long long *__restrict__ key)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // plus_plus_init.wgsl:161-168 `initial`
        uint32_t px = rgba[index];
        float L, a, b;
        linear100_to_lab(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u], L, a, b);
        Centroid c; c.L = L; c.a = a; c.b = b; c.C = chroma(a, b);
        cent[0] = c;
        *key = 0ull;
    }
}

hipError_t launch_init_first(const uint32_t *rgba, uint64_t index, const float *lut, Centroid *cent,
                             unsigned long long *key, hipStream_t st)
{
    hipLaunchKernelGGL(k_init_first, dim3(1), dim3(64), 0, st, rgba, index, lut, cent, key);
    return hipGetLastError();
}

// PICK (a whole image on one device): one launch per pass, no atomics.  `key` is then an array of 2 x kInitSlots slots: launch j
// first picks centroid j - 1 = the pixel named by the largest slot of launch j - 1 -- plus_plus_init.wgsl:172-181 `pick`, by
// EVERY workgroup, the same arithmetic on the same pixel -- then runs pass j and leaves its workgroups' keys in the other slot
// set.  (The reference's default call initialises on a <= 256 x 256 image: 255 passes at k = 256, each 4.9 us + a 2 us pick
// launch; 171 workgroups' atomicMax on one address retire at ~13 ns each, 2 us at the end of every pass.)
// The last centroid is picked by k_init_pick_slots.
constexpr uint32_t kInitSlots = 2048;          // >= the grid of k_init_pass

__device__ __forceinline__ uint32_t init_key_index(unsigned long long kk)
{
    uint32_t index = 0;                                       // Candidate(0, 0.0) when every distance is 0
    if ((kk >> 32) != 0ull) {
        const uint32_t low = (uint32_t)kk;
        index = (low & ~15u) | (15u - (low & 15u));
    }
    return index;
}

// largest of n_slots keys, in thread 0 (s_key: kBlock / 64 words of LDS; ends with a barrier)
__device__ __forceinline__ unsigned long long init_max_slot(const unsigned long long *__restrict__ slots, uint32_t n_slots,
                                                            unsigned long long *s_key)
{
    unsigned long long m = 0ull;
    for (uint32_t q = threadIdx.x; q < n_slots; q += kBlock) { const unsigned long long v = slots[q]; m = v > m ? v : m; }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < kBlock / 64; ++w) m = s_key[w] > m ? s_key[w] : m;
    return m;
}

template <bool PICK>
__global__ __launch_bounds__(kBlock) void k_init_pass(const uint32_t *__restrict__ rgba, uint64_t n,
                                                      const float *__restrict__ lut,
                                                      Centroid *__restrict__ cent, uint32_t j,
                                                      float *__restrict__ dist,
                                                      unsigned long long *__restrict__ key,
                                                      uint64_t first_index)
{
    // first_index: image-wide linear index of this band's first pixel (0 for a whole image), so the
    // tie rule of the key is the image's, not the band's
    __shared__ float s_lut[256];
    __shared__ unsigned long long s_key[kBlock / 64];
    __shared__ Centroid s_c;
    s_lut[threadIdx.x] = lut[threadIdx.x];
    // (this thread's first pixel and its running distance do not depend on the new centroid: requested before the pick, whose
    // chain of two dependent loads and a Lab conversion they then overlap -- a pass is a chain of round trips, 4.6 -> ~4 us)
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t i_first = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t px_first = 0u;
    float d_first = 1000000.0f;
    if (PICK && i_first < n) {
        px_first = rgba[i_first];
        if (j != 1) d_first = dist[i_first];
    }
    float L_first = 0.0f, a_first = 0.0f, b_first = 0.0f;
    if (PICK && j >= 2u) {
        const unsigned long long kk = init_max_slot(key + ((j - 1u) & 1u) * kInitSlots, gridDim.x, s_key);   // (a barrier inside: s_lut is there)
        if (threadIdx.x == 0) {
            const uint32_t px = rgba[init_key_index(kk)];
            float L, a, b;
            linear100_to_lab(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u], L, a, b);
            Centroid o; o.L = L; o.a = a; o.b = b; o.C = chroma(a, b);
            s_c = o;
            if (blockIdx.x == 0) cent[j - 1u] = o;
        }
        px_to_lab(s_lut, px_first, L_first, a_first, b_first);
    }
    __syncthreads();
    if (PICK && j < 2u) px_to_lab(s_lut, px_first, L_first, a_first, b_first);
    const Centroid c = (PICK && j >= 2u) ? s_c : cent[j - 1];
    unsigned long long best = 0ull;
    for (uint64_t i = i_first; i < n; i += stride) {
        float L, a, b, before;
        if (PICK && i == i_first) {
            L = L_first; a = a_first; b = b_first; before = d_first;
        } else {
            px_to_lab(s_lut, rgba[i], L, a, b);
            before = j == 1 ? 1000000.0f : dist[i];
        }
        // kmeans++_calc_diff.wgsl:26-30; the running minimum equals the recomputed one
        float d = cie94(L, a, b, c.L, c.a, c.b);
        float m = fminf(before, d);
        dist[i] = m;
        const uint64_t gi = first_index + i;
        unsigned long long kk = ((unsigned long long)float_to_bits(m) << 32) |
                                (unsigned long long)(((uint32_t)(gi >> 4) << 4) | (15u - (uint32_t)(gi & 15u)));
        best = kk > best ? kk : best;
    }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(best, off, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) best = s_key[w] > best ? s_key[w] : best;
        if (PICK) key[(j & 1u) * kInitSlots + blockIdx.x] = best;
        else atomicMax(key, best);
    }
}

// the last centroid of a PICK initialisation: from the slots of pass j
__global__ __launch_bounds__(kBlock) void k_init_pick_slots(const uint32_t *__restrict__ rgba, const float *__restrict__ lut,
                                                            const unsigned long long *__restrict__ slots, uint32_t n_slots,
                                                            Centroid *__restrict__ cent, uint32_t j)
{
    __shared__ unsigned long long s_key[kBlock / 64];
    const unsigned long long kk = init_max_slot(slots + (j & 1u) * kInitSlots, n_slots, s_key);
    if (threadIdx.x == 0) {
        const uint32_t px = rgba[init_key_index(kk)];
        float L, a, b;
        linear100_to_lab(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u], L, a, b);
        Centroid o; o.L = L; o.a = a; o.b = b; o.C = chroma(a, b);
        cent[j] = o;
    }
}

// ------------------------------------------------------------------------------------------
// Several centroids per launch, exactly (whole image on one device; round 5).
//
// A pass only LOWERS running distances, so every pixel's key can only decrease.  Let k1 > k2 > ... be the largest keys after
// the sweep that applied centroids c_0 .. c_{t-1}; c_t = pixel(k1) (plus_plus_init.wgsl:172-181).  If cie94(pixel(k2), c_t) >=
// dist(k2), the sweep against c_t leaves k2 untouched while every other key stays below it or drops: pixel(k2) IS c_{t+1} --
// without that sweep.  Likewise pixel(k3) against c_t and c_{t+1}, and so on until the first failure.  (A key whose distance
// is 0 is never accepted this way: that case is Candidate(0, 0.0) = pixel 0, init_key_index.)  One launch therefore
//   [picks up to kInitMulti centroids from the top keys the previous launch left] -> [one sweep: dist = min(dist, all of them)]
// Farthest points are far from each other: the reference's default call (<= 256 x 256 pixels, k = 256) takes ~90 launches
// instead of 255 (measured acceptance: profiles/NOTES.md round 5), each a chain of round trips whatever it computes.
// slots: [2][grid][kInitMulti] candidates (key + Lab of its pixel), one row per WORKGROUP of launch parity L & 1 (sorted,
// largest first; key 0 = none);
// count[2]: centroids chosen after launch parity L & 1.  The host enqueues launches in chunks and reads count between them.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kInitMulti = 4;

// a candidate: its key and the Lab of its pixel (travels with the key, so that a pick fetches no pixel)
struct alignas(16) InitCand { unsigned long long key; float L, a, b; uint32_t pad; };
static_assert(sizeof(InitCand) == 32, "InitCand layout");

struct TopList {
    unsigned long long key[kInitMulti];
    float L[kInitMulti], a[kInitMulti], b[kInitMulti];
};

__device__ __forceinline__ void top_clear(TopList &t)
{
#pragma unroll
    for (uint32_t i = 0; i < kInitMulti; ++i) { t.key[i] = 0ull; t.L[i] = 0.0f; t.a[i] = 0.0f; t.b[i] = 0.0f; }
}

__device__ __forceinline__ void top_insert(TopList &t, unsigned long long x, float L, float a, float b)
{
    if (x > t.key[kInitMulti - 1]) {
        t.key[kInitMulti - 1] = x; t.L[kInitMulti - 1] = L; t.a[kInitMulti - 1] = a; t.b[kInitMulti - 1] = b;
#pragma unroll
        for (int i = (int)kInitMulti - 1; i > 0; --i)
            if (t.key[i] > t.key[i - 1]) {
                const unsigned long long u = t.key[i]; t.key[i] = t.key[i - 1]; t.key[i - 1] = u;
                float f;
                f = t.L[i]; t.L[i] = t.L[i - 1]; t.L[i - 1] = f;
                f = t.a[i]; t.a[i] = t.a[i - 1]; t.a[i - 1] = f;
                f = t.b[i]; t.b[i] = t.b[i - 1]; t.b[i - 1] = f;
            }
    }
}

// the largest head of the lanes' sorted lists: every lane gets the key, the lane that holds it writes the candidate to
// *out and drops it from its list (keys are distinct or 0; key 0: lane 0 writes an empty candidate)
__device__ __forceinline__ void wave_pop_max(TopList &t, InitCand *out, uint32_t lane)
{
    const uint32_t hi = (uint32_t)(t.key[0] >> 32), lo = (uint32_t)t.key[0];
    const uint32_t mhi = wave_max_u32_dpp(hi);
    const uint32_t mlo = wave_max_u32_dpp(hi == mhi ? lo : 0u);
    const unsigned long long g = ((unsigned long long)mhi << 32) | mlo;
    if (g == 0ull) {
        if (lane == 0u) { InitCand o; o.key = 0ull; o.L = 0.0f; o.a = 0.0f; o.b = 0.0f; o.pad = 0u; *out = o; }
    } else if (t.key[0] == g) {
        InitCand o; o.key = g; o.L = t.L[0]; o.a = t.a[0]; o.b = t.b[0]; o.pad = 0u;
        *out = o;
#pragma unroll
        for (uint32_t i = 0; i + 1 < kInitMulti; ++i) { t.key[i] = t.key[i + 1]; t.L[i] = t.L[i + 1]; t.a[i] = t.a[i + 1]; t.b[i] = t.b[i + 1]; }
        t.key[kInitMulti - 1] = 0ull;
    }
}

__global__ __launch_bounds__(kBlock) void k_init_multi(const uint32_t *__restrict__ rgba, uint64_t n, const float *__restrict__ lut,
                                                       Centroid *__restrict__ cent, uint32_t k, uint32_t launch, float *__restrict__ dist,
                                                       InitCand *__restrict__ slots, uint32_t *__restrict__ count)
{
    constexpr uint32_t kWaves = kBlock / 64;
    __shared__ float s_lut[256];
    __shared__ InitCand s_top[kWaves * kInitMulti];
    __shared__ InitCand s_best[kInitMulti];
    __shared__ uint32_t s_fail;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t n_rows = gridDim.x;                              // rows of a slot set: one per workgroup of a launch
    s_lut[threadIdx.x] = lut[threadIdx.x];
    if (threadIdx.x == 0) s_fail = 0u;
    // (this thread's first pixel and its running distance do not depend on the picks: requested before them)
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t i_first = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t px_first = 0u;
    float d_first = 1000000.0f;                                     // kmeans++_calc_diff.wgsl:26-30
    if (i_first < n) {
        px_first = rgba[i_first];
        if (launch != 1u) d_first = dist[i_first];
    }
    uint32_t n_new = 1u;                                            // centroids this launch sweeps against
    float L_first = 0.0f, a_first = 0.0f, b_first = 0.0f;
    if (launch == 1u) {
        // centroid 0 is there (k_init_first): nothing to pick
        if (threadIdx.x == 0) {
            const Centroid c0 = cent[0];
            InitCand o; o.key = 0ull; o.L = c0.L; o.a = c0.a; o.b = c0.b; o.pad = 0u;
            s_best[0] = o;
            if (blockIdx.x == 0) count[1] = 1u;
        }
        __syncthreads();
        px_to_lab(s_lut, px_first, L_first, a_first, b_first);
    } else {
        // the kInitMulti largest candidates of the previous launch: every thread merges its share of the rows, every wave
        // its threads', thread 0..15 ranks the waves' 16.  The first rows of a thread are requested whole and together with
        // the count -- one round trip; a row is sorted, so further rows (large grids) are read only when their head counts.
        const InitCand *prev = slots + (uint64_t)((launch - 1u) & 1u) * n_rows * kInitMulti;
        TopList t;
        top_clear(t);
        {
            // thread r takes row r (sorted: it IS the thread's list) -- requested together with the count: one round trip
            InitCand row[kInitMulti];
#pragma unroll
            for (uint32_t e = 0; e < kInitMulti; ++e) {
                row[e].key = 0ull; row[e].L = 0.0f; row[e].a = 0.0f; row[e].b = 0.0f;
                if (threadIdx.x < n_rows) row[e] = prev[(uint64_t)threadIdx.x * kInitMulti + e];
            }
            const uint32_t have_ = count[(launch - 1u) & 1u];
            if (have_ >= k) {                                       // the table is complete: an empty launch of a chunk
                if (blockIdx.x == 0 && threadIdx.x == 0) count[launch & 1u] = have_;
                return;
            }
#pragma unroll
            for (uint32_t e = 0; e < kInitMulti; ++e) { t.key[e] = row[e].key; t.L[e] = row[e].L; t.a[e] = row[e].a; t.b[e] = row[e].b; }
        }
        const uint32_t have = count[(launch - 1u) & 1u];
        // (more workgroups than threads -- images beyond 65 536 pixels: further rows are read only when their head counts)
        for (uint32_t r = threadIdx.x + kBlock; r < n_rows; r += kBlock) {
            const InitCand *rp = prev + (uint64_t)r * kInitMulti;
            if (rp[0].key > t.key[kInitMulti - 1])
                for (uint32_t e = 0; e < kInitMulti; ++e) { const InitCand cnd = rp[e]; top_insert(t, cnd.key, cnd.L, cnd.a, cnd.b); }
        }
#pragma unroll
        for (uint32_t r = 0; r < kInitMulti; ++r) wave_pop_max(t, &s_top[wv * kInitMulti + r], lane);
        __syncthreads();
        px_to_lab(s_lut, px_first, L_first, a_first, b_first);      // (own pixel: independent of the picks)
        if (threadIdx.x < kWaves * kInitMulti) {
            const InitCand mine = s_top[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < kWaves * kInitMulti; ++q) {
                const unsigned long long o = s_top[q].key;
                rank += (o > mine.key || (o == mine.key && q < threadIdx.x)) ? 1u : 0u;
            }
            if (rank < kInitMulti) {
                InitCand o = mine;
                if ((uint32_t)(o.key >> 32) == 0u && rank == 0u) {
                    // every distance is 0: Candidate(0, 0.0) = pixel 0 (plus_plus_init.wgsl:62-68, 172-181)
                    px_to_lab(s_lut, rgba[0], o.L, o.a, o.b);
                }
                s_best[rank] = o;
            }
        }
        __syncthreads();
        // the tests: thread (a, b), b < a -- does the sweep against candidate b leave candidate a's distance alone?
        if (threadIdx.x < kInitMulti * kInitMulti) {
            const uint32_t a = threadIdx.x / kInitMulti, b = threadIdx.x % kInitMulti;
            if (b < a) {
                const InitCand pa = s_best[a], cb = s_best[b];
                const float have_d = bits_to_float((uint32_t)(pa.key >> 32));
                if (!(cie94(pa.L, pa.a, pa.b, cb.L, cb.a, cb.b) >= have_d)) atomicOr(&s_fail, 1u << a);
            }
        }
        __syncthreads();
        const uint32_t fail = s_fail;
        if ((uint32_t)(s_best[0].key >> 32) != 0u) {                // (all distances zero: one pick)
            while (n_new < kInitMulti && have + n_new < k && (uint32_t)(s_best[n_new].key >> 32) != 0u && !((fail >> n_new) & 1u)) ++n_new;
        }
        if (blockIdx.x == 0 && threadIdx.x < n_new) {
            const InitCand o = s_best[threadIdx.x];
            Centroid c; c.L = o.L; c.a = o.a; c.b = o.b; c.C = chroma(o.a, o.b);
            cent[have + threadIdx.x] = c;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) count[launch & 1u] = have + n_new;
        if (have + n_new >= k) return;                              // the last centroids: no distances are needed any more
    }
    float cL[kInitMulti], ca[kInitMulti], cb_[kInitMulti];
#pragma unroll
    for (uint32_t q = 0; q < kInitMulti; ++q) { const InitCand o = s_best[q < n_new ? q : 0u]; cL[q] = o.L; ca[q] = o.a; cb_[q] = o.b; }
    TopList t;
    top_clear(t);
    for (uint64_t i = i_first; i < n; i += stride) {
        float L, a, b, before;
        if (i == i_first) { L = L_first; a = a_first; b = b_first; before = d_first; }
        else { px_to_lab(s_lut, rgba[i], L, a, b); before = launch == 1u ? 1000000.0f : dist[i]; }
        // (all kInitMulti distances, unconditionally -- entries beyond n_new repeat the first: four independent chains that
        // the scheduler interleaves cost little more than one, a branch per centroid serialises them)
        float d4[kInitMulti];
#pragma unroll
        for (uint32_t q = 0; q < kInitMulti; ++q) d4[q] = cie94(L, a, b, cL[q], ca[q], cb_[q]);
        const float m = fminf(fminf(before, d4[0]), fminf(fminf(d4[1], d4[2]), d4[3]));
        dist[i] = m;
        const unsigned long long kk = ((unsigned long long)float_to_bits(m) << 32) |
                                      (unsigned long long)(((uint32_t)(i >> 4) << 4) | (15u - (uint32_t)(i & 15u)));
        top_insert(t, kk, L, a, b);
    }
    // the workgroup's row: its waves' top candidates, ranked by 16 threads
    __syncthreads();                                                // (s_top / s_best of the picks are read no more)
#pragma unroll
    for (uint32_t r = 0; r < kInitMulti; ++r) wave_pop_max(t, &s_top[wv * kInitMulti + r], lane);
    __syncthreads();
    if (threadIdx.x < kWaves * kInitMulti) {
        const InitCand mine = s_top[threadIdx.x];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < kWaves * kInitMulti; ++q) {
            const unsigned long long o = s_top[q].key;
            rank += (o > mine.key || (o == mine.key && q < threadIdx.x)) ? 1u : 0u;
        }
        if (rank < kInitMulti) slots[((uint64_t)(launch & 1u) * n_rows + blockIdx.x) * kInitMulti + rank] = mine;
    }
}

static uint32_t init_pass_grid(uint64_t n)
{
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    return (uint32_t)(blocks < kInitSlots ? (blocks ? blocks : 1) : kInitSlots);
}

size_t init_slots_bytes() { return sizeof(unsigned long long) * 2u * kInitSlots; }

// Every workgroup of a launch merges ALL rows of the previous one (no device-wide step in between), so the grid stays small: at
// most one row per thread of the merge -- 256 workgroups, the image's pixels strided over their 65 536 threads.  (With the 2048
// workgroups of the single-pick pass the merge alone moved 2048^2 x 128 B = 537 MB per launch: 92 us at 524 288 pixels.)
static uint32_t init_multi_grid(uint64_t n)
{
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    return (uint32_t)(blocks < (uint64_t)kBlock ? (blocks ? blocks : 1) : (uint64_t)kBlock);
}

static size_t init_multi_cands(uint64_t n) { return 2ull * init_multi_grid(n) * kInitMulti; }

size_t init_multi_bytes(uint64_t n) { return sizeof(InitCand) * init_multi_cands(n) + 2u * sizeof(uint32_t); }

hipError_t launch_init_multi(const uint32_t *rgba, uint64_t n, const float *lut, Centroid *cent, uint32_t k, uint32_t launch,
                             float *dist, void *scratch, hipStream_t st)
{
    InitCand *slots = static_cast<InitCand *>(scratch);
    uint32_t *count = reinterpret_cast<uint32_t *>(slots + init_multi_cands(n));
    hipLaunchKernelGGL(k_init_multi, dim3(init_multi_grid(n)), dim3(kBlock), 0, st, rgba, n, lut, cent, k, launch, dist, slots, count);
    return hipGetLastError();
}

const uint32_t *init_multi_count(const void *scratch, uint64_t n, uint32_t launch)
{
    const InitCand *slots = static_cast<const InitCand *>(scratch);
    return reinterpret_cast<const uint32_t *>(slots + init_multi_cands(n)) + (launch & 1u);
}

hipError_t launch_init_pick_slots(const uint32_t *rgba, uint64_t n, const float *lut, const unsigned long long *slots, Centroid *cent,
                                  uint32_t j, hipStream_t st)
{
    hipLaunchKernelGGL(k_init_pick_slots, dim3(1), dim3(kBlock), 0, st, rgba, lut, slots, init_pass_grid(n), cent, j);
    return hipGetLastError();
}

hipError_t launch_init_pass(const uint32_t *rgba, uint64_t n, const float *lut, Centroid *cent,
                            uint32_t j, float *dist, unsigned long long *key, uint64_t first_index, hipStream_t st, bool pick)
{
    const uint32_t grid = init_pass_grid(n);
    if (pick) hipLaunchKernelGGL(k_init_pass<true>, dim3(grid), dim3(kBlock), 0, st, rgba, n, lut, cent, j, dist, key, first_index);
    else hipLaunchKernelGGL(k_init_pass<false>, dim3(grid), dim3(kBlock), 0, st, rgba, n, lut, cent, j, dist, key, first_index);
    return hipGetLastError();
}

// Sharded init: the (all-reduced, max) key names one pixel of the whole image; the band that owns it
// publishes its colour as {rgba, 1}, every other band {0, 0}, so a sum all-reduce delivers it everywhere.
__global__ void k_init_pick_band(const uint32_t *__restrict__ rgba, uint64_t n, uint64_t first_index,
                                 const unsigned long long *__restrict__ key, uint32_t *__restrict__ colour2)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long kk = *key;
        uint64_t index = 0;                                   // Candidate(0, 0.0) when every distance is 0
        if ((kk >> 32) != 0ull) {
            const uint32_t low = (uint32_t)kk;
            index = (low & ~15u) | (15u - (low & 15u));
        }
        const bool mine = index >= first_index && index < first_index + n;
        colour2[0] = mine ? rgba[index - first_index] : 0u;
        colour2[1] = mine ? 1u : 0u;
    }
}

hipError_t launch_init_pick_band(const uint32_t *rgba, uint64_t n, uint64_t first_index,
                                 const unsigned long long *key, uint32_t *colour2, hipStream_t st)
{
    hipLaunchKernelGGL(k_init_pick_band, dim3(1), dim3(64), 0, st, rgba, n, first_index, key, colour2);
    return hipGetLastError();
}

// centroid j <- shader Lab of one RGBA8 colour held in device memory
__global__ void k_set_centroid_rgba(const uint32_t *__restrict__ colour, const float *__restrict__ lut,
                                    Centroid *__restrict__ cent, uint32_t j)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t px = colour[0];
        float L, a, b;
        linear100_to_lab(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u], L, a, b);
        Centroid c; c.L = L; c.a = a; c.b = b; c.C = chroma(a, b);
        cent[j] = c;
    }
}

hipError_t launch_set_centroid_rgba(const uint32_t *colour, const float *lut, Centroid *cent, uint32_t j,
                                    hipStream_t st)
{
    hipLaunchKernelGGL(k_set_centroid_rgba, dim3(1), dim3(64), 0, st, colour, lut, cent, j);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Bilinear shrink (resize.wgsl:7-18 + the sampler of structures.rs:121-131): uv = gid / dims,
// clamp-to-edge, linear filter with exact f32 weights, x first then y, rgba8unorm store.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_resize(const uint32_t *__restrict__ rgba, uint32_t w,
                                                   uint32_t h, uint32_t nw, uint32_t nh,
                                                   uint32_t *__restrict__ out, uint32_t src_row0,
                                                   uint32_t out_row0, uint32_t out_rows)
{
    // `rgba` starts at image row src_row0 and `out` at output row out_row0 (a row band of a sharded image, launch_resize_band);
    // the whole image: 0, 0, nh
    const uint64_t total = (uint64_t)nw * out_rows;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const uint32_t gx = (uint32_t)(i % nw), gy = (uint32_t)(i / nw) + out_row0;
        const float u = (float)gx / (float)nw, v = (float)gy / (float)nh;
        const float tx = u * (float)w - 0.5f, ty = v * (float)h - 0.5f;
        const float fx0 = floorf(tx), fy0 = floorf(ty);
        const float wx = tx - fx0, wy = ty - fy0;
        long long x0 = (long long)fx0, y0 = (long long)fy0;
        long long x1 = x0 + 1, y1 = y0 + 1;
        x0 = x0 < 0 ? 0 : (x0 > (long long)w - 1 ? (long long)w - 1 : x0);
        x1 = x1 < 0 ? 0 : (x1 > (long long)w - 1 ? (long long)w - 1 : x1);
        y0 = y0 < 0 ? 0 : (y0 > (long long)h - 1 ? (long long)h - 1 : y0);
        y1 = y1 < 0 ? 0 : (y1 > (long long)h - 1 ? (long long)h - 1 : y1);
        y0 -= src_row0; y1 -= src_row0;
        const uint32_t p00 = rgba[(uint64_t)y0 * w + x0], p10 = rgba[(uint64_t)y0 * w + x1];
        const uint32_t p01 = rgba[(uint64_t)y1 * w + x0], p11 = rgba[(uint64_t)y1 * w + x1];
        uint32_t o = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const float t00 = (float)((p00 >> (8 * ch)) & 255u) / 255.0f;
            const float t10 = (float)((p10 >> (8 * ch)) & 255u) / 255.0f;
            const float t01 = (float)((p01 >> (8 * ch)) & 255u) / 255.0f;
            const float t11 = (float)((p11 >> (8 * ch)) & 255u) / 255.0f;
            const float top = fmaf(wx, t10 - t00, t00);
            const float bot = fmaf(wx, t11 - t01, t01);
            o |= unorm8(fmaf(wy, bot - top, top)) << (8 * ch);
        }
        out[i] = o;
    }
}

hipError_t launch_resize(const uint32_t *rgba, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh,
                         uint32_t *out, hipStream_t st)
{
    return launch_resize_band(rgba, w, h, 0, nw, nh, 0, nh, out, st);
}

// source row of the first sample of output row gy (resize.wgsl:15-16 with clamp-to-edge): the same binary32 operations as the
// kernel, so that a host can tell which band of a sharded image an output row belongs to
uint32_t resize_source_row(uint32_t gy, uint32_t h, uint32_t nh)
{
    const float v = (float)gy / (float)nh;
    const float ty = v * (float)h - 0.5f;
    long long y0 = (long long)floorf(ty);
    y0 = y0 < 0 ? 0 : (y0 > (long long)h - 1 ? (long long)h - 1 : y0);
    return (uint32_t)y0;
}

hipError_t launch_resize_band(const uint32_t *band, uint32_t w, uint32_t h, uint32_t src_row0, uint32_t nw, uint32_t nh,
                              uint32_t out_row0, uint32_t out_rows, uint32_t *out, hipStream_t st)
{
    if (out_rows == 0) return hipSuccess;
    uint64_t blocks = ((uint64_t)nw * out_rows + kBlock - 1) / kBlock;
    uint32_t grid = (uint32_t)(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
    hipLaunchKernelGGL(k_resize, dim3(grid), dim3(kBlock), 0, st, band, w, h, nw, nh, out, src_row0, out_row0, out_rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Output pass at full resolution: find_centroid + swap + lab_to_rgb (replace), or
// mix_colors.wgsl main_dither + lab_to_rgb (dither).  Only k (+1 sentinel) distinct output colours
// exist, so the Lab->sRGB8 conversion is done once per centroid on the host (pal[]).
// ------------------------------------------------------------------------------------------
__constant__ const float c_bayer[16] = {0, 8, 2, 10, 12, 4, 14, 6, 3, 11, 1, 9, 15, 7, 13, 5};

template <int PPT, bool DITHER, bool CHUNKED>
__global__ __launch_bounds__(kBlock) void k_apply(const uint32_t *__restrict__ rgba, uint32_t w,
                                                  uint64_t n, uint32_t row0,
                                                  const Centroid *__restrict__ cent, uint32_t k,
                                                  const float *__restrict__ lut,
                                                  const uint32_t *__restrict__ pal, float threshold,
                                                  uint32_t *__restrict__ out, int aligned)
{
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 3u) & ~3u;
    float4 *s_cent = smem4;
    float *s_lut = reinterpret_cast<float *>(smem4 + kpad);
    float *s_off = s_lut + 256;

    s_lut[threadIdx.x] = lut[threadIdx.x];
    stage_centroids(s_cent, cent, k, kpad);
    if (DITHER && threadIdx.x < 16) {
        // mix_colors.wgsl:21-27,70,72: threshold * (M[x%4 + 4*(y%4)] / 16 - 0.5)
        float iv = c_bayer[threadIdx.x] / 16.0f - 0.5f;
        s_off[threadIdx.x] = threshold * iv;
    }
    __syncthreads();

    constexpr int GROUPS = PPT / 4;
    constexpr uint64_t TILE = (uint64_t)kBlock * PPT;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint64_t i0[GROUPS];
        PixelTerms pt[PPT];
        float best[PPT];
        uint32_t idx[PPT];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            i0[g] = tile * TILE + (uint64_t)g * (kBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4(rgba, i0[g], n, aligned != 0, px);
            // image coordinates of the group's first pixel (n < 2^32): one 32-bit divide per 4 pixels
            uint32_t gx = 0, gy = 0;
            if (DITHER) {
                const uint32_t i32 = (uint32_t)i0[g];
                gy = i32 / w;
                gx = i32 - gy * w;
                gy += row0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = g * 4 + q;
                float L, a, b;
                px_to_lab(s_lut, px[q], L, a, b);
                if (DITHER) {
                    const float off = s_off[(gx & 3u) + ((gy & 3u) << 2)];
                    L = L + off; a = a + off; b = b + off;         // :72
                    gx += 1;
                    if (gx == w) { gx = 0; gy += 1; }
                }
                pt[p] = pixel_terms(L, a, b);
                if (DITHER) {
                    // :73 closest = vec3(10000.0): the running minimum starts at the sentinel's
                    // distance; index k selects the converted sentinel in pal[]
                    best[p] = cie94_key(pt[p], 10000.0f, 10000.0f, 10000.0f, chroma(10000.0f, 10000.0f));
                    idx[p] = k;
                } else {
                    best[p] = 1.0e10f;                             // find_centroid.wgsl:29-30
                    idx[p] = 0u;
                }
            }
        }
        argmin_scan<PPT, CHUNKED, DITHER>(pt, best, idx, s_cent, k, kpad);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pal[idx[g * 4 + q]];
            store4(out, i0[g], n, aligned != 0, o);
        }
    }
}

hipError_t launch_apply(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0,
                        const Centroid *cent, uint32_t k, const float *lut, const uint32_t *pal,
                        bool dither, float threshold, uint32_t *out, hipStream_t st)
{
    const uint64_t n = (uint64_t)w * rows;
    const uint64_t tiles = (n + (uint64_t)kBlock * kAssignPPT - 1) / ((uint64_t)kBlock * kAssignPPT);
    const uint32_t grid = (uint32_t)(tiles < 2048 ? (tiles ? tiles : 1) : 2048);
    const uint32_t kpad = (k + 3u) & ~3u;
    const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 &&
                         (reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? 1 : 0;
    const bool chunked = k >= 32;
    const size_t lds = sizeof(float4) * kpad + (256 + 16) * sizeof(float);
#define KMG_APPLY(D, C)                                                                            \
    hipLaunchKernelGGL((k_apply<kAssignPPT, D, C>), dim3(grid), dim3(kBlock), lds, st, rgba, w, n, \
                       row0, cent, k, lut, pal, threshold, out, aligned)
    if (dither) { if (chunked) KMG_APPLY(true, true); else KMG_APPLY(true, false); }
    else        { if (chunked) KMG_APPLY(false, true); else KMG_APPLY(false, false); }
#undef KMG_APPLY
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// meld output pass: mix_colors.wgsl:29-48 two_closest_colors, :85-90 meld, :117-135 main_meld,
// then lab_to_rgb.wgsl per pixel.  Literal CIE94 throughout (the values themselves are used).
// ------------------------------------------------------------------------------------------
// thr[0] unused, thr[b] for b = 1 .. 255: the smallest float c with unorm8(srgb_encode_dev(c)) >= b (found by bisection over
// the bit patterns of the non-negative floats, which order like the values)
__global__ void k_encode_thresholds(float *__restrict__ thr)
{
    const uint32_t b = threadIdx.x;
    if (b == 0u) { thr[0] = -3.0e38f; return; }
    uint32_t lo = 0u, hi = 0x3F800000u;                           // byte(0.0) = 0 < b <= 255 = byte(1.0)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (unorm8(srgb_encode_dev(bits_to_float(mid))) >= b) hi = mid; else lo = mid;
    }
    thr[b] = bits_to_float(hi);
}

hipError_t launch_encode_thresholds(float *thr, hipStream_t st)
{
    hipLaunchKernelGGL(k_encode_thresholds, dim3(1), dim3(256), 0, st, thr);
    return hipGetLastError();
}

// Test support: over EVERY non-negative float bit pattern up to +inf (2^31 - 2^23 + 1 values) and their negatives, the
// number of values whose table byte differs from the byte the encode itself gives.
__global__ __launch_bounds__(kBlock) void k_encode_check(const float *__restrict__ thr, unsigned long long *__restrict__ bad)
{
    __shared__ float s_thr[257];
    s_thr[threadIdx.x] = thr[threadIdx.x];
    if (threadIdx.x == 0) s_thr[256] = 3.0e38f;
    __syncthreads();
    uint32_t mine = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v <= 0x7F800000ull; v += stride) {
        const float c = bits_to_float((uint32_t)v);
        mine += encode_byte(s_thr, c) != unorm8(srgb_encode_dev(c));
        mine += encode_byte(s_thr, -c) != unorm8(srgb_encode_dev(-c));
    }
    if (mine) atomicAdd(bad, (unsigned long long)mine);
}

hipError_t launch_encode_check(const float *thr, unsigned long long *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_encode_check, dim3(256 * 16), dim3(kBlock), 0, st, thr, bad);
    return hipGetLastError();
}

// Test support: for one constant c, over EVERY binary32 x (NaN aside): out[0] += #x with div_const(x, c) != x / c (bit patterns;
// -0 and +0 differ), out[1] = the smallest and out[2] = the largest |x| (bit pattern) among them
__global__ __launch_bounds__(kBlock) void k_division_check(float c, float rc, unsigned long long *__restrict__ out)
{
    unsigned long long bad = 0;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < (1ull << 32); v += stride) {
        const float x = bits_to_float((uint32_t)v);
        if (x != x) continue;
        const float q = div_const(x, c, rc), want = x / c;
        if (float_to_bits(q) != float_to_bits(want)) {
            ++bad;
            const uint32_t m = (uint32_t)v & 0x7FFFFFFFu;
            lo = m < lo ? m : lo; hi = m > hi ? m : hi;
        }
    }
    if (bad) {
        atomicAdd(out, bad);
        atomicMin(out + 1, (unsigned long long)lo);
        atomicMax(out + 2, (unsigned long long)hi);
    }
}

hipError_t launch_division_check(float c, unsigned long long *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_division_check, dim3(256 * 16), dim3(kBlock), 0, st, c, 1.0f / c, out);
    return hipGetLastError();
}

// masks != NULL: per colour cell, the centroids that can be among the pixel's two closest (kmg_table.hip,
// k_meld_candidates); the ordered scan then visits only those -- same two slots, same output
__global__ __launch_bounds__(kBlock) void k_meld(const uint32_t *__restrict__ rgba, uint64_t n,
                                                 const Centroid *__restrict__ cent, uint32_t k,
                                                 const float *__restrict__ lut, const uint64_t *__restrict__ masks,
                                                 uint32_t *__restrict__ out)
{
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 3u) & ~3u;
    float4 *s_cent = smem4;
    float *s_lut = reinterpret_cast<float *>(smem4 + kpad);
    float *s_thr = s_lut + 256;                                    // the encode thresholds live behind the decode table
    s_lut[threadIdx.x] = lut[threadIdx.x];
    s_thr[threadIdx.x] = lut[256 + threadIdx.x];
    if (threadIdx.x == 0) s_thr[256] = 3.0e38f;
    stage_centroids(s_cent, cent, k, kpad);
    __syncthreads();
    const uint32_t words = (k + 63u) / 64u;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t px = rgba[i];
        float L, a, b;
        px_to_lab(s_lut, px, L, a, b);
        if (k == 1) {                                            // mix_colors.wgsl:127-131
            const float4 c = s_cent[0];
            out[i] = lab_to_rgba8_dev<true>(c.x, c.y, c.z, s_thr);
            continue;
        }
        // :30-31 closest = second_closest = vec4(10000.0)
        float cL = 10000.0f, ca = 10000.0f, cb = 10000.0f, sL = 10000.0f, sa = 10000.0f, sb = 10000.0f;
        float d_closest = cie94(L, a, b, cL, ca, cb), d_second = d_closest;
        // (the two chromas of delta_e.wgsl:8-9 are the pixel's, computed once, and the centroid's, kept in the table:
        // cie94_c performs the same operations on them, hence returns the same float -- two square roots fewer per visit)
        const float C1 = chroma(a, b);
        auto visit = [&](uint32_t j) {
            const float4 c = s_cent[j];
            const float d = cie94_c(L, a, b, C1, c.x, c.y, c.z, c.w);
            if (d < d_closest) {                                 // :36-38
                sL = cL; sa = ca; sb = cb; d_second = d_closest;
                cL = c.x; ca = c.y; cb = c.z; d_closest = d;
            } else if (d < d_second) {                           // :39-41
                sL = c.x; sa = c.y; sb = c.z; d_second = d;
            }
        };
        if (masks) {
            const uint32_t cell = (((px >> 3) & 31u) << 10) | (((px >> 11) & 31u) << 5) | ((px >> 19) & 31u);
            const uint64_t *mw = masks + (uint64_t)cell * words;
            unsigned long long first4[4];                        // k <= 256: every word requested at once
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) first4[w] = w < words ? mw[w] : 0ull;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                unsigned long long m = first4[w];
                while (m) {
                    visit(w * 64 + (uint32_t)__builtin_ctzll(m));
                    m &= m - 1;
                }
            }
            for (uint32_t w = 4; w < words; ++w) {
                unsigned long long m = mw[w];
                while (m) {
                    visit(w * 64 + (uint32_t)__builtin_ctzll(m));
                    m &= m - 1;
                }
            }
        } else {
            for (uint32_t j = 0; j < k; ++j) visit(j);
        }
        // :86-89
        const float factor = cie94(L, a, b, sL, sa, sb) / cie94(cL, ca, cb, sL, sa, sb);
        const float oL = factor * cL + (1.0f - factor) * sL;
        const float oa = factor * ca + (1.0f - factor) * sa;
        const float ob = factor * cb + (1.0f - factor) * sb;
        out[i] = lab_to_rgba8_dev<true>(oL, oa, ob, s_thr);
    }
}

hipError_t launch_meld(const uint32_t *rgba, uint64_t n, const Centroid *cent, uint32_t k, const float *lut,
                       const uint64_t *masks, uint32_t *out, hipStream_t st)
{
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    const uint32_t grid = (uint32_t)(blocks < 8192 ? (blocks ? blocks : 1) : 8192);
    const uint32_t kpad = (k + 3u) & ~3u;
    const size_t lds = sizeof(float4) * kpad + (256 + 257) * sizeof(float);
    hipLaunchKernelGGL(k_meld, dim3(grid), dim3(kBlock), lds, st, rgba, n, cent, k, lut, masks, out);
    return hipGetLastError();
}

}  // namespace kmg
