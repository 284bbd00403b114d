// kmg_table.hip -- gfx950 kernels of the colour-table strategy (see kmg_table.h).
//
// Exactness argument (tests/test_gpu_table.py checks every step against the brute-force scan):
//  * key_range() bounds the FLOAT value cie94_key() returns for any colour of a cell: every
//    operation of cie94_key is monotone in each operand (IEEE rounding is monotone), so evaluating
//    the same operations on interval end points gives rigorous bounds without any epsilon.
//  * a centroid can be the arg-min of some colour of the cell only if its lower bound does not
//    exceed U = min_j upper_j; the candidate mask keeps exactly those, in index order, so the
//    strict-'<' first-minimum-wins scan over the candidates returns the brute-force label.
//  * sums are exact integers: sum over colours of count * q equals the per-pixel sum of q.
// Compile with -ffp-contract=off.

#include "kmg_internal.h"
#include "kmg_table_dev.h"

#include <hip/hip_fp16.h>

namespace kmg {

// ------------------------------------------------------------------------------------------
// static bounds (once per processor): one workgroup per cell, two colours per thread.  Wave w meets the
// colours of sub-cell w in its first pass and of sub-cell 4 + w in its second, so its two wave reductions
// are those sub-cells' bounds; the cell's bounds are their union.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cell_bounds(const float *__restrict__ lut,
                                                        CellBounds *__restrict__ bounds,
                                                        CellBounds *__restrict__ sub_bounds,
                                                        float4 *__restrict__ lab_table)
{
    __shared__ float s_lut[256];
    __shared__ float s_min[6][8], s_max[6][8];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (uint32_t pass = 0; pass < 2u; ++pass) {
        const uint32_t c = pass * kBlock + threadIdx.x;
        float L, a, b;
        colour_to_lab(s_lut, blockIdx.x * kCellColours + c, L, a, b);
        const PixelTerms p = pixel_terms(L, a, b);
        lab_table[blockIdx.x * kCellColours + c] = make_float4(p.L, p.a, p.b, p.C);
        float mn[6] = {p.L, p.a, p.b, p.C, p.wC, p.wH}, mx[6] = {p.L, p.a, p.b, p.C, p.wC, p.wH};
#pragma unroll
        for (int q = 0; q < 6; ++q)
            for (int off = 32; off > 0; off >>= 1) {
                mn[q] = fminf(mn[q], __shfl_xor(mn[q], off, 64));
                mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], off, 64));
            }
        const uint32_t sc = pass * 4u + wv;                          // colour >> 6
        if (lane == 0) {
            CellBounds sb;
            sb.L0 = mn[0]; sb.L1 = mx[0]; sb.a0 = mn[1]; sb.a1 = mx[1]; sb.b0 = mn[2]; sb.b1 = mx[2];
            sb.C0 = mn[3]; sb.C1 = mx[3]; sb.wC0 = mn[4]; sb.wC1 = mx[4]; sb.wH0 = mn[5]; sb.wH1 = mx[5];
            sb.pad[0] = sb.pad[1] = sb.pad[2] = sb.pad[3] = 0.0f;
            sub_bounds[(uint64_t)blockIdx.x * 8u + sc] = sb;
            for (int q = 0; q < 6; ++q) { s_min[q][sc] = mn[q]; s_max[q][sc] = mx[q]; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[6], hi[6];
        for (int q = 0; q < 6; ++q) {
            lo[q] = s_min[q][0]; hi[q] = s_max[q][0];
            for (int w = 1; w < 8; ++w) { lo[q] = fminf(lo[q], s_min[q][w]); hi[q] = fmaxf(hi[q], s_max[q][w]); }
        }
        CellBounds cb;
        cb.L0 = lo[0]; cb.L1 = hi[0]; cb.a0 = lo[1]; cb.a1 = hi[1]; cb.b0 = lo[2]; cb.b1 = hi[2];
        cb.C0 = lo[3]; cb.C1 = hi[3]; cb.wC0 = lo[4]; cb.wC1 = hi[4]; cb.wH0 = lo[5]; cb.wH1 = hi[5];
        cb.pad[0] = cb.pad[1] = cb.pad[2] = cb.pad[3] = 0.0f;
        bounds[blockIdx.x] = cb;
    }
}

hipError_t launch_cell_bounds(const float *lut, CellBounds *bounds, CellBounds *sub_bounds, float4 *lab_table, hipStream_t st)
{
    hipLaunchKernelGGL(k_cell_bounds, dim3(kCells), dim3(kBlock), 0, st, lut, bounds, sub_bounds, lab_table);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// histogram of the image's colours (once per image), cell-major order
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_histogram(const uint32_t *__restrict__ rgba, uint64_t n,
                                                      uint32_t *__restrict__ hist, int aligned)
{
    constexpr uint64_t TILE = (uint64_t)kBlock * 8;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint64_t i0 = tile * TILE + (uint64_t)g * (kBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4(rgba, i0, n, aligned != 0, px);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i0 + q < n) atomicAdd(hist + colour_index(px[q]), 1u);
        }
    }
}

hipError_t launch_histogram(const uint32_t *rgba, uint64_t n, uint32_t *hist, hipStream_t st)
{
    const uint64_t tiles = (n + kBlock * 8 - 1) / (kBlock * 8);
    const uint32_t grid = (uint32_t)(tiles < 4096 ? (tiles ? tiles : 1) : 4096);
    const int aligned = (reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 ? 1 : 0;
    hipLaunchKernelGGL(k_histogram, dim3(grid), dim3(kBlock), 0, st, rgba, n, hist, aligned);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Histogram (and init tie keys) of a large image without a global atomic per pixel.
//
// 67 M random increments into a 64 MiB table cost 2.5 ms as global atomics (every one is an L2 miss).
// Instead the pixels are first partitioned by the top 10 bits of their colour index (1024 partitions
// of 16384 colours): k_part_count totals the partitions, k_part_prefix turns the totals into segment
// starts, k_part_scatter writes each pixel's low 14 colour bits (u16) -- and, for the init, the low
// half of its arg-max key -- into its partition's segment (ranks from LDS atomics, one global atomic
// per partition per 16384-pixel tile), and k_part_histogram counts one segment chunk per workgroup
// entirely in LDS (64 KiB of counters + 64 KiB of tie keys) and stores the result coalesced.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kPartBits = 14;
constexpr uint32_t kParts = 1u << (24 - kPartBits);               // 1024
constexpr uint32_t kPartColours = 1u << kPartBits;                // 16384
constexpr uint32_t kPartChunk = 1u << 20;                         // elements per k_part_histogram workgroup
constexpr int kPartBlock = 1024;

__global__ __launch_bounds__(kPartBlock) void k_part_count(const uint32_t *__restrict__ rgba, uint64_t n,
                                                           uint32_t *__restrict__ totals, int aligned)
{
    __shared__ uint32_t s_cnt[kParts];
    s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    constexpr uint64_t TILE = (uint64_t)kPartBlock * 8;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint64_t i0 = tile * TILE + (uint64_t)g * (kPartBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4(rgba, i0, n, aligned != 0, px);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i0 + q < n) atomicAdd(&s_cnt[colour_index(px[q]) >> kPartBits], 1u);
        }
    }
    __syncthreads();
    if (s_cnt[threadIdx.x]) atomicAdd(totals + threadIdx.x, s_cnt[threadIdx.x]);
}

// start[p] = first element of partition p; cursor[p] = start[p]; chunk_first[p] = first k_part_histogram
// workgroup of partition p (every partition has at least one), chunk_first[kParts] = their number
__global__ __launch_bounds__(kPartBlock) void k_part_prefix(const uint32_t *__restrict__ totals,
                                                            uint32_t *__restrict__ start, uint32_t *__restrict__ cursor,
                                                            uint32_t *__restrict__ chunk_first)
{
    __shared__ uint32_t s_a[kParts], s_b[kParts];
    const uint32_t t = threadIdx.x;
    const uint32_t mine = totals[t];
    const uint32_t chunks = mine ? (mine + kPartChunk - 1) / kPartChunk : 1u;
    s_a[t] = mine; s_b[t] = chunks;
    __syncthreads();
    for (uint32_t off = 1; off < kParts; off <<= 1) {           // Hillis-Steele inclusive scan of both arrays
        const uint32_t a = t >= off ? s_a[t - off] : 0u, b = t >= off ? s_b[t - off] : 0u;
        __syncthreads();
        s_a[t] += a; s_b[t] += b;
        __syncthreads();
    }
    start[t] = s_a[t] - mine;
    cursor[t] = s_a[t] - mine;
    chunk_first[t] = s_b[t] - chunks;
    if (t == kParts - 1) chunk_first[kParts] = s_b[t];
}

// A tile of 16384 pixels is counting-sorted by partition inside LDS (ranks from LDS atomics, a scan of
// the 1024 tile counts), one global atomic per non-empty partition reserves its run in the segment, and
// the sorted tile is written out: consecutive lanes write consecutive elements of a few runs instead of
// 64 unrelated addresses.
__global__ __launch_bounds__(kPartBlock) void k_part_scatter(const uint32_t *__restrict__ rgba, uint64_t n,
                                                             uint64_t first_index, uint32_t *__restrict__ cursor,
                                                             uint16_t *__restrict__ elems, uint32_t *__restrict__ keys,
                                                             int aligned)
{
    constexpr int G = 4;                                            // 16 pixels per thread and tile
    constexpr uint32_t TILE = (uint32_t)kPartBlock * 4 * G;
    __shared__ uint32_t s_cnt[kParts], s_pre[kParts], s_base[kParts];
    extern __shared__ uint32_t s_dyn[];                             // [keys TILE u32 (if keys)][low colour bits TILE u16][partition TILE u16]
    uint32_t *s_key = s_dyn;
    uint16_t *s_el = reinterpret_cast<uint16_t *>(s_dyn + (keys ? TILE : 0u));
    uint16_t *s_pt = s_el + TILE;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t tile0 = tile * TILE;
        const uint32_t tile_n = (uint32_t)(n - tile0 < TILE ? n - tile0 : TILE);
        s_cnt[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t where[4 * G];                                      // [partition:10][rank within the tile:15]
        uint16_t low[4 * G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint64_t i0 = tile0 + (uint64_t)g * (kPartBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4(rgba, i0, n, aligned != 0, px);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t ci = colour_index(px[q]);
                const uint32_t part = ci >> kPartBits;
                uint32_t rank = 0;
                if (i0 + q < n) rank = atomicAdd(&s_cnt[part], 1u);
                where[g * 4 + q] = (part << 15) | rank;
                low[g * 4 + q] = (uint16_t)(ci & (kPartColours - 1u));
            }
        }
        __syncthreads();
        const uint32_t mine = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = mine ? atomicAdd(cursor + threadIdx.x, mine) : 0u;
        s_pre[threadIdx.x] = mine;
        __syncthreads();
        for (uint32_t off = 1; off < kParts; off <<= 1) {          // inclusive scan of the tile counts
            const uint32_t a = threadIdx.x >= off ? s_pre[threadIdx.x - off] : 0u;
            __syncthreads();
            s_pre[threadIdx.x] += a;
            __syncthreads();
        }
        s_pre[threadIdx.x] -= mine;                                 // exclusive: first sorted position of the partition
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint64_t i0 = tile0 + (uint64_t)g * (kPartBlock * 4) + (uint64_t)threadIdx.x * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q < n) {
                    const uint32_t wq = where[g * 4 + q], part = wq >> 15;
                    const uint32_t j = s_pre[part] + (wq & 32767u);
                    s_el[j] = low[g * 4 + q];
                    s_pt[j] = (uint16_t)part;
                    if (keys) {
                        const uint64_t gi = first_index + i0 + q;
                        s_key[j] = ((uint32_t)(gi >> 4) << 4) | (15u - (uint32_t)(gi & 15u));
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < tile_n; j += kPartBlock) {
            const uint32_t part = s_pt[j];
            const uint64_t dst = (uint64_t)s_base[part] + (j - s_pre[part]);
            elems[dst] = s_el[j];
            if (keys) keys[dst] = s_key[j];
        }
        __syncthreads();
    }
}

// one workgroup per (partition, chunk of <= 2^20 elements); hist / tie are zero on entry
__global__ __launch_bounds__(kPartBlock) void k_part_histogram(const uint16_t *__restrict__ elems,
                                                               const uint32_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ start,
                                                               const uint32_t *__restrict__ totals,
                                                               const uint32_t *__restrict__ chunk_first,
                                                               uint32_t *__restrict__ hist, uint32_t *__restrict__ tie)
{
    extern __shared__ uint32_t s_tab[];                             // [counts 16384][tie keys 16384 (if keys)]
    uint32_t *s_cnt = s_tab, *s_tie = s_tab + kPartColours;
    if (blockIdx.x >= chunk_first[kParts]) return;
    uint32_t lo = 0, hi = kParts - 1;                               // last partition whose first workgroup is <= blockIdx.x
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (chunk_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const uint32_t part = lo;
    const uint32_t chunk = blockIdx.x - chunk_first[part];
    const uint32_t n_chunks = chunk_first[part + 1] - chunk_first[part];
    const uint64_t total = totals[part];
    const uint64_t b = (uint64_t)chunk * kPartChunk, e = b + kPartChunk < total ? b + kPartChunk : total;
    if (b >= e) return;                                             // empty partition: hist / tie stay zero
    for (uint32_t i = threadIdx.x; i < kPartColours; i += kPartBlock) { s_cnt[i] = 0u; if (keys) s_tie[i] = 0u; }
    __syncthreads();
    const uint64_t seg = start[part];
    for (uint64_t i0 = b + threadIdx.x; i0 < e; i0 += 8ull * kPartBlock) {      // 8 loads in flight per thread
        uint32_t c[8], kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t i = i0 + (uint64_t)u * kPartBlock;
            c[u] = i < e ? (uint32_t)elems[seg + i] : 0xFFFFFFFFu;
            kk[u] = (keys && i < e) ? keys[seg + i] + 1u : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c[u] != 0xFFFFFFFFu) {
                atomicAdd(&s_cnt[c[u]], 1u);
                if (keys) atomicMax(&s_tie[c[u]], kk[u]);
            }
        }
    }
    __syncthreads();
    const uint32_t out = part * kPartColours;
    for (uint32_t i = threadIdx.x; i < kPartColours; i += kPartBlock) {
        const uint32_t c = s_cnt[i];
        if (n_chunks == 1u) {
            hist[out + i] = c;
            if (keys) tie[out + i] = s_tie[i];
        } else if (c) {
            atomicAdd(hist + out + i, c);
            if (keys) atomicMax(tie + out + i, s_tie[i]);
        }
    }
}

// scratch: 3 * kParts + 1 u32 (totals, start, cursor, chunk_first) zeroed by the caller is NOT needed:
// this function clears what it uses.  elems: n u16; keys: n u32 or NULL (then tie is not touched).
hipError_t launch_partitioned_histogram(const uint32_t *rgba, uint64_t n, uint64_t first_index, uint32_t *small,
                                        uint16_t *elems, uint32_t *keys, uint32_t *hist, uint32_t *tie, hipStream_t st)
{
    uint32_t *totals = small, *start = small + kParts, *cursor = small + 2 * kParts, *chunk_first = small + 3 * kParts;
    const int aligned = (reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 ? 1 : 0;
    hipError_t e = hipMemsetAsync(totals, 0, sizeof(uint32_t) * kParts, st);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(hist, 0, sizeof(uint32_t) << 24, st)) != hipSuccess) return e;
    if (keys && (e = hipMemsetAsync(tie, 0, sizeof(uint32_t) << 24, st)) != hipSuccess) return e;
    const uint64_t tiles8 = (n + kPartBlock * 8 - 1) / (kPartBlock * 8);
    const uint32_t grid_count = (uint32_t)(tiles8 < 512 ? (tiles8 ? tiles8 : 1) : 512);
    hipLaunchKernelGGL(k_part_count, dim3(grid_count), dim3(kPartBlock), 0, st, rgba, n, totals, aligned);
    hipLaunchKernelGGL(k_part_prefix, dim3(1), dim3(kPartBlock), 0, st, totals, start, cursor, chunk_first);
    const uint64_t tiles16 = (n + kPartBlock * 16 - 1) / (kPartBlock * 16);
    const uint32_t grid_scatter = (uint32_t)(tiles16 < 256 ? (tiles16 ? tiles16 : 1) : 256);   // one workgroup per CU (LDS)
    const size_t lds_scatter = (size_t)kPartBlock * 16 * (keys ? 8 : 4);
    hipLaunchKernelGGL(k_part_scatter, dim3(grid_scatter), dim3(kPartBlock), lds_scatter, st, rgba, n, first_index, cursor, elems, keys, aligned);
    const uint32_t grid_hist = kParts + (uint32_t)(n / kPartChunk) + 1u;      // >= the number of chunks
    const size_t lds = sizeof(uint32_t) * kPartColours * (keys ? 2 : 1);
    hipLaunchKernelGGL(k_part_histogram, dim3(grid_hist), dim3(kPartBlock), lds, st, elems, keys, start, totals, chunk_first, hist, tie);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Farthest-point initialisation over the image's colours (plus_plus_init.wgsl, kmeans++_calc_diff.wgsl)
//
// All pixels of one colour share their distance to the chosen centroids, so the arg-max key of
// kmg_kernels.hip -- [distance bits:32][index / 16:28][15 - index % 16:4] -- is maximised per
// colour by the pixel with the largest low half.  tie[colour] = 1 + that largest low half (0 = no
// pixel has this colour), built once per image; a pass then walks the 2^24 colours (28 B each:
// tie, running distance, Lab from the static table) instead of the pixels (sRGB->Lab + 12 B each).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tie_keys(const uint32_t *__restrict__ rgba, uint64_t n,
                                                     uint64_t first_index, uint32_t *__restrict__ tie, int aligned)
{
    constexpr uint64_t TILE = (uint64_t)kBlock * 8;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint64_t i0 = tile * TILE + (uint64_t)g * (kBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4(rgba, i0, n, aligned != 0, px);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q < n) {
                    const uint64_t gi = first_index + i0 + q;
                    const uint32_t low = ((uint32_t)(gi >> 4) << 4) | (15u - (uint32_t)(gi & 15u));
                    atomicMax(tie + colour_index(px[q]), low + 1u);
                }
            }
        }
    }
}

hipError_t launch_tie_keys(const uint32_t *rgba, uint64_t n, uint64_t first_index, uint32_t *tie, hipStream_t st)
{
    const uint64_t tiles = (n + kBlock * 8 - 1) / (kBlock * 8);
    const uint32_t grid = (uint32_t)(tiles < 4096 ? (tiles ? tiles : 1) : 4096);
    const int aligned = (reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 ? 1 : 0;
    hipLaunchKernelGGL(k_tie_keys, dim3(grid), dim3(kBlock), 0, st, rgba, n, first_index, tie, aligned);
    return hipGetLastError();
}

// A float lower bound of cie94(pixel, centroid) (kmg_math.h, literal form, weights from the pixel) over
// every colour of a cell: the same operations on the interval end points that minimise each term
// (every operation is monotone, so no colour of the cell can come out lower).
__device__ __forceinline__ float cie94_lower_bound(const CellBounds &cb, float L2, float a2, float b2, float C2)
{
    float mL, ML, ma, Ma, mb, Mb, mC, MC;
    abs_range(cb.L0, cb.L1, L2, mL, ML);
    abs_range(cb.a0, cb.a1, a2, ma, Ma);
    abs_range(cb.b0, cb.b1, b2, mb, Mb);
    abs_range(cb.C0, cb.C1, C2, mC, MC);
    const float SC = 1.0f + 0.045f * cb.C1, SH = 1.0f + 0.015f * cb.C1;      // the largest divisors
    const float dH = sqrtf(fmaxf((ma * ma) + (mb * mb) - (MC * MC), 0.0f));
    const float tL = mL / 1.0f, tC = mC / SC, tH = dH / SH;
    return sqrtf(tL * tL + tC * tC + tH * tH);
}

// One init pass = ONE launch (kInitGrid workgroups of kInitBlock threads).
//
// records[i] (i = position in the work list of the bound image's occupied cells) holds everything the test of one
// cell needs: the cell, its Lab bounds, and the largest key of the cell's colours together with the Lab of the
// colour that holds it.  A new centroid can only lower the running distances, and it cannot lower any distance
// of a cell whose lower bound to it is not below the cell's largest running distance -- such a cell (most of them
// once a few dozen centroids exist) is skipped and its cached key stays valid.
//
// Who does what.  The cells a centroid reaches are neighbours in colour space, i.e. runs in the work list: the
// list is dealt out so that an 8 x 8 x 4 block of cells goes to 256 DIFFERENT workgroups (slot_work_index), each
// workgroup tests its 128 cells with one lane per cell, pools the reached ones in LDS and its 16 waves take them
// in turn -- a pass costs what the busiest workgroup costs, ~1/16 of what the busiest wave cost when every wave
// visited its own cells (measured: 28 -> 11 us per pass once a few dozen centroids exist).
// A visit reads the running distances, the occupancy byte and the Lab of the cell's colours; the tie keys (which
// pixel of a colour) are only needed for the colours that hold the cell's largest distance, normally one.
//
// A late pass is a chain of memory round trips, not work, so the chain is kept short: the records are the only
// thing the test reads (no work list -> cell -> bounds indirection) and are requested before the pick; a visit
// requests everything at once, and when no wave has a second visit to hide the latency behind (<= 16 reached
// cells in the workgroup) that includes the tie keys.
//
// The arg-max over the cells is split between two launches: every workgroup leaves the largest record it met in
// slots[j & 1][workgroup]; the NEXT launch starts by reducing the kInitGrid slots (every workgroup does, 8 KiB
// from L2) and so knows centroid j -- the Lab of the winning colour travels with the key, no pixel is fetched
// (plus_plus_init.wgsl:172-181 `pick`; all distances zero: pixel 0).  Launch j therefore is
//     [PICK: centroid j - 1 <- slots of launch j - 1]  ->  [pass against centroid j - 1 -> slots]
// and launch k (do_pass = 0) only picks the last centroid: k launches for k - 1 passes instead of 2 (k - 1).
// PICK = false (band of a sharded image: the pick is an all-reduce between two launches): centroid j - 1 is read
// from cent[], and k_init_reduce_slots turns the slots into the band's key.
constexpr uint32_t kInitGrid = 256, kInitBlock = 1024;
static_assert(kInitGrid * (kInitBlock / 64) * 8 == kCells, "one test slot per cell");
constexpr uint32_t kNoCell = 0xFFFFFFFFu;

struct alignas(16) InitSlot { unsigned long long key; uint32_t pad[2]; float4 lab; };
struct alignas(16) InitRecord {
    unsigned long long key;        // largest key of the cell's colours (0 before the first pass)
    uint32_t cell;                 // kNoCell: past the end of the work list
    uint32_t pad;
    float4 lab;                    // Lab of the colour that holds it
    CellBounds cb;
};
static_assert(sizeof(InitSlot) == 32 && sizeof(InitRecord) == 32 + sizeof(CellBounds), "init record layout");

// (init_scratch_bytes: behind k_init_cells_multi, whose rows it also holds)

// wave-wide maxima through DPP row operations (kmg_table_dev.h): a 64-bit maximum is the maximum of the high words, then of the
// low words of the lanes that hold it.  (Rounds 2-4 used six shuffle steps -- twelve ds_bpermute round trips for 64 bits -- in
// kernels that are chains of dependent steps: cfg3's initialisation 3.85 -> 3.63 ms.)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mhi = wave_max_u32_dpp(hi);
    const uint32_t mlo = wave_max_u32_dpp(hi == mhi ? lo : 0u);
    return ((unsigned long long)mhi << 32) | mlo;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return wave_max_u32_dpp(v); }

// test slot q (0..127) of workgroup g (0..255) -> index into the work list: with every cell occupied the index
// is the cell [r:5][g:5][b:5], and g = [r1 r0][g2 g1 g0][b2 b1 b0], q = [r4 r3 r2][g4 g3][b4 b3]
__device__ __forceinline__ uint32_t slot_work_index(uint32_t g, uint32_t q)
{
    const uint32_t b = (g & 7u) | ((q & 3u) << 3), gg = ((g >> 3) & 7u) | (((q >> 2) & 3u) << 3), r = (g >> 6) | ((q >> 4) << 2);
    return (r << 10) | (gg << 5) | b;
}

// once per initialisation: the records of the bound image's occupied cells, keys zero
__global__ __launch_bounds__(kBlock) void k_init_records(const uint32_t *__restrict__ work,
                                                         const CellBounds *__restrict__ bounds,
                                                         InitRecord *__restrict__ records, InitSlot *__restrict__ slots)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    InitRecord r;
    r.key = 0ull; r.cell = kNoCell; r.pad = 0u; r.lab = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    r.cb = bounds[0];
    if (i < work[0]) { r.cell = work[1u + i]; r.cb = bounds[r.cell]; }
    records[i] = r;
    if (i < 2u * kInitGrid) { InitSlot o; o.key = 0ull; o.pad[0] = 0u; o.pad[1] = 0u; o.lab = r.lab; slots[i] = o; }
}

hipError_t launch_init_records(const uint32_t *work, const CellBounds *bounds, void *init_scratch, hipStream_t st)
{
    InitRecord *records = (InitRecord *)init_scratch;
    hipLaunchKernelGGL(k_init_records, dim3(kCells / kBlock), dim3(kBlock), 0, st, work, bounds, records, (InitSlot *)(records + kCells));
    return hipGetLastError();
}

// One PART of a cell's visit: 64 lanes x N colours (N = 4: a half, N = 2: a quarter of the cell).  When a workgroup has
// few cells to visit, several of its 16 waves share a cell: the literal CIE94 of a lane's colours is a serial chain
// (~250 operations per colour), so 2 colours per lane instead of 8 take a quarter of the time.  The part's key is the
// largest key of its colours (distance, then low half), exactly as for a whole cell; the cell's key is the largest
// part key.  Returns the key in the lane that holds it (0 elsewhere) and that colour's Lab.
template <int N>
__device__ __forceinline__ unsigned long long init_visit_part(uint32_t cell, uint32_t part, uint32_t lane, uint32_t j, const Centroid &c,
                                                              const uint32_t *__restrict__ tie, const uint8_t *__restrict__ occ_bits,
                                                              const float4 *__restrict__ lab_table, float *__restrict__ dist,
                                                              float4 &lab_out)
{
    constexpr uint32_t kParts = 8u / N;
    const uint32_t within = part * (kCellColours / kParts) + lane * N;          // first colour of this lane inside the cell
    const uint32_t base = cell * kCellColours + within;
    const uint32_t occ = ((uint32_t)occ_bits[(uint64_t)cell * 64u + (within >> 3)] >> (within & 7u)) & ((1u << N) - 1u);
    float4 v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = lab_table[base + q];
    uint32_t t[N];
    float m[N];
    if (N == 4) {
        const uint4 tt = *reinterpret_cast<const uint4 *>(tie + base);
        t[0] = tt.x; t[1] = tt.y; t[2 % N] = tt.z; t[3 % N] = tt.w;
        float4 d = make_float4(1000000.0f, 1000000.0f, 1000000.0f, 1000000.0f);   // kmeans++_calc_diff.wgsl:26-30
        if (j != 1) d = *reinterpret_cast<const float4 *>(dist + base);
        m[0] = d.x; m[1] = d.y; m[2 % N] = d.z; m[3 % N] = d.w;
    } else {
        const uint2 tt = *reinterpret_cast<const uint2 *>(tie + base);
        t[0] = tt.x; t[1] = tt.y;
        float2 d = make_float2(1000000.0f, 1000000.0f);
        if (j != 1) d = *reinterpret_cast<const float2 *>(dist + base);
        m[0] = d.x; m[1] = d.y;
    }
    uint32_t md = 0u;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        if ((occ >> q) & 1u) {
            m[q] = fminf(m[q], cie94(v[q].x, v[q].y, v[q].z, c.L, c.a, c.b));
            md = max(md, float_to_bits(m[q]));
        }
    }
    if (occ || j == 1) {
        if (N == 4) *reinterpret_cast<float4 *>(dist + base) = make_float4(m[0], m[1], m[2 % N], m[3 % N]);
        else *reinterpret_cast<float2 *>(dist + base) = make_float2(m[0], m[1]);
    }
    const uint32_t wmd = wave_max_u32(md);
    uint32_t low1 = 0u;
    lab_out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (occ && md == wmd) {
#pragma unroll
        for (int q = 0; q < N; ++q)
            if (((occ >> q) & 1u) && float_to_bits(m[q]) == wmd && t[q] > low1) { low1 = t[q]; lab_out = v[q]; }
    }
    const uint32_t wlow1 = wave_max_u32(low1);
    return (low1 == wlow1 && low1 != 0u) ? (((unsigned long long)wmd << 32) | (unsigned long long)(wlow1 - 1u)) : 0ull;
}

template <bool PICK>
__global__ __launch_bounds__(kInitBlock) void k_init_fused(const uint32_t *__restrict__ tie,
                                                           const uint8_t *__restrict__ occ_bits,
                                                           const float4 *__restrict__ lab_table,
                                                           Centroid *__restrict__ cent, uint32_t j, int do_pass,
                                                           float *__restrict__ dist, InitRecord *__restrict__ records,
                                                           InitSlot *__restrict__ slots,
                                                           const uint32_t *__restrict__ rgba,
                                                           const float *__restrict__ lut)
{
    __shared__ unsigned long long s_key[kInitBlock / 64];
    __shared__ float4 s_lab[kInitBlock / 64];
    __shared__ float4 s_cent;
    __shared__ uint2 s_list[(kInitBlock / 64) * 8];                // (cell, its position in the work list)
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_pkey[kInitBlock / 64];         // split visits: the waves' part keys
    __shared__ float4 s_plab[kInitBlock / 64];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    // what the test needs does not depend on the new centroid: requested before the pick
    const uint32_t wi = slot_work_index(blockIdx.x, wv * 8u + (lane & 7u));
    const bool tester = do_pass && lane < 8u;
    InitRecord rec;
    rec.key = 0ull; rec.cell = kNoCell;
    if (tester) rec = records[wi];
    if (threadIdx.x == 0) s_count = 0u;

    Centroid c;
    if (PICK && j >= 2u) {
        // centroid j - 1 = the largest record of the previous launch
        const InitSlot *prev = slots + ((j - 1u) & 1u) * kInitGrid;
        unsigned long long key = 0ull;
        float4 lab = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (threadIdx.x < kInitGrid) { key = prev[threadIdx.x].key; lab = prev[threadIdx.x].lab; }
        if (wv < kInitGrid / 64u) {
            const unsigned long long best = wave_max_u64(key);
            const uint32_t src = (uint32_t)__builtin_ctzll(__ballot(key == best));
            if (lane == src) { s_key[wv] = best; s_lab[wv] = lab; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // (the four entries requested together, not one dependent LDS round trip per comparison)
            const unsigned long long k0 = s_key[0], k1 = s_key[1], k2 = s_key[2], k3 = s_key[3];
            static_assert(kInitGrid / 64u == 4u, "four slot waves");
            const uint32_t wa = k1 > k0 ? 1u : 0u, wb = k3 > k2 ? 3u : 2u;
            const uint32_t w = (wb == 3u ? k3 : k2) > (wa == 1u ? k1 : k0) ? wb : wa;
            float4 v = s_lab[w];
            if ((s_key[w] >> 32) == 0ull) {
                // Candidate(0, 0.0): every distance is 0 -> pixel 0
                const uint32_t px = rgba[0];
                linear100_to_lab(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u], v.x, v.y, v.z);
            }
            v.w = chroma(v.y, v.z);
            s_cent = v;
            if (blockIdx.x == 0) { Centroid o; o.L = v.x; o.a = v.y; o.b = v.z; o.C = v.w; cent[j - 1u] = o; }
        }
        __syncthreads();
        const float4 v = s_cent;
        c.L = v.x; c.a = v.y; c.b = v.z; c.C = v.w;
    } else {
        c = cent[j - 1u];
    }
    if (!do_pass) return;
    __syncthreads();                                               // s_count is zero, s_key / s_lab are free again

    unsigned long long run_key = 0ull;                             // the largest record this lane has met
    float4 run_lab = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tester && rec.cell != kNoCell) {
        bool reached = true;
        if (j != 1) {
            const float cell_max = __uint_as_float((uint32_t)(rec.key >> 32));
            reached = cie94_lower_bound(rec.cb, c.L, c.a, c.b, c.C) < cell_max;
        }
        if (reached) s_list[atomicAdd(&s_count, 1u)] = make_uint2(rec.cell, wi);
        else { run_key = rec.key; run_lab = rec.lab; }
    }
    __syncthreads();
    const uint32_t count = s_count;
    const bool eager = count <= kInitBlock / 64u;                  // one visit per wave at most: nothing to hide a round trip behind

    if (count <= 8u) {
        // few cells: 4 (count <= 4) or 2 waves per cell, each a part of its colours (init_visit_part)
        const uint32_t shift = count <= 4u ? 2u : 1u;              // log2(waves per cell)
        const uint32_t my = wv >> shift, part = wv & ((1u << shift) - 1u);
        unsigned long long pk = 0ull;
        float4 pl = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (my < count) {
            const uint32_t cell = s_list[my].x;
            pk = shift == 2u ? init_visit_part<2>(cell, part, lane, j, c, tie, occ_bits, lab_table, dist, pl)
                             : init_visit_part<4>(cell, part, lane, j, c, tie, occ_bits, lab_table, dist, pl);
        }
        if (lane == 0u) s_pkey[wv] = 0ull;
        if (pk != 0ull) { s_pkey[wv] = pk; s_plab[wv] = pl; }     // (same wave, after the 0: LDS writes of a wave stay in order)
        __syncthreads();
        if (part == 0u && my < count && lane == 0u) {
            uint32_t w = wv;
            for (uint32_t q = 1; q < (1u << shift); ++q) if (s_pkey[wv + q] > s_pkey[w]) w = wv + q;
            const unsigned long long key = s_pkey[w];
            const float4 lab = s_plab[w];
            InitRecord *r = records + s_list[my].y;
            r->key = key;
            r->lab = lab;
            if (key >= run_key) { run_key = key; run_lab = lab; }
        }
    }
    // visits: wave wv takes entries wv, wv + 16, ...; the next cell's occupancy and distances are requested
    // (unconditionally: past the end the current cell again, unused) before the current cell's Lab values are waited for
    uint32_t idx = count <= 8u ? count : wv;
    uint2 ent = idx < count ? s_list[idx] : make_uint2(0u, 0u);
    uint32_t base = ent.x * kCellColours + lane * 8u;
    uint32_t occ = occ_bits[(uint64_t)ent.x * 64u + lane];
    float4 d0 = make_float4(1000000.0f, 1000000.0f, 1000000.0f, 1000000.0f), d1 = d0;     // kmeans++_calc_diff.wgsl:26-30
    if (j != 1) { d0 = *reinterpret_cast<const float4 *>(dist + base); d1 = *reinterpret_cast<const float4 *>(dist + base + 4); }
    while (idx < count) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = lab_table[base + q];    // (a static table: every address is readable)
        uint4 t0 = make_uint4(0u, 0u, 0u, 0u), t1 = t0;
        if (eager) { t0 = *reinterpret_cast<const uint4 *>(tie + base); t1 = *reinterpret_cast<const uint4 *>(tie + base + 4); }
        const uint32_t idx_n = idx + kInitBlock / 64u;
        const uint2 ent_n = idx_n < count ? s_list[idx_n] : ent;
        const uint32_t base_n = ent_n.x * kCellColours + lane * 8u;
        const uint32_t occ_n = occ_bits[(uint64_t)ent_n.x * 64u + lane];
        float4 d0_n = d0, d1_n = d1;                               // (j = 1: the map starts at 1e6, nothing to read)
        if (j != 1) { d0_n = *reinterpret_cast<const float4 *>(dist + base_n); d1_n = *reinterpret_cast<const float4 *>(dist + base_n + 4); }

        float m[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        uint32_t md = 0u;                                          // largest distance (bits) among this lane's colours
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if ((occ >> q) & 1u) {
                m[q] = fminf(m[q], cie94(v[q].x, v[q].y, v[q].z, c.L, c.a, c.b));
                md = max(md, float_to_bits(m[q]));
            }
        }
        if (occ || j == 1) {
            *reinterpret_cast<float4 *>(dist + base) = make_float4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<float4 *>(dist + base + 4) = make_float4(m[4], m[5], m[6], m[7]);
        }
        // the cell's key = (largest distance, largest low half among the colours that hold it)
        const uint32_t wmd = wave_max_u32(md);
        uint32_t low1 = 0u;                                        // 1 + low half; 0 = this lane does not hold the maximum
        float4 best_lab = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (occ && md == wmd) {
            if (!eager) { t0 = *reinterpret_cast<const uint4 *>(tie + base); t1 = *reinterpret_cast<const uint4 *>(tie + base + 4); }
            const uint32_t t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (((occ >> q) & 1u) && float_to_bits(m[q]) == wmd && t[q] > low1) { low1 = t[q]; best_lab = v[q]; }
            }
        }
        const uint32_t wlow1 = wave_max_u32(low1);
        // (tie keys of occupied colours are distinct -- they name distinct pixels)
        if (low1 == wlow1 && low1 != 0u) {
            const unsigned long long key = ((unsigned long long)wmd << 32) | (unsigned long long)(wlow1 - 1u);
            InitRecord *r = records + ent.y;
            r->key = key;
            r->lab = best_lab;
            if (key >= run_key) { run_key = key; run_lab = best_lab; }
        }
        idx = idx_n; ent = ent_n; base = base_n; occ = occ_n; d0 = d0_n; d1 = d1_n;
    }
    // the largest record of the workgroup
    {
        const unsigned long long wbest = wave_max_u64(run_key);
        if (lane == (uint32_t)__builtin_ctzll(__ballot(run_key == wbest))) { s_key[wv] = wbest; s_lab[wv] = run_lab; }
        __syncthreads();
        if (wv == 0u) {
            // the largest of the 16 waves' entries: lane l holds entry l, one more wave maximum (a serial scan by thread 0 was
            // 16 dependent LDS round trips at the end of every launch)
            const unsigned long long e = lane < kInitBlock / 64u ? s_key[lane] : 0ull;
            const float4 el = s_lab[lane < kInitBlock / 64u ? lane : 0u];
            const unsigned long long best = wave_max_u64(e);
            if (lane == (uint32_t)__builtin_ctzll(__ballot(e == best))) {
                InitSlot o; o.key = best; o.pad[0] = 0u; o.pad[1] = 0u; o.lab = el;
                slots[(j & 1u) * kInitGrid + blockIdx.x] = o;
            }
        }
    }
}

// key = the largest slot key of pass j (band of a sharded image: the caller all-reduces it)
__global__ __launch_bounds__(kInitGrid) void k_init_reduce_slots(const InitSlot *__restrict__ slots, uint32_t j,
                                                                 unsigned long long *__restrict__ key)
{
    __shared__ unsigned long long s_key[kInitGrid / 64];
    const unsigned long long best = wave_max_u64(slots[(j & 1u) * kInitGrid + threadIdx.x].key);
    if ((threadIdx.x & 63u) == 0u) s_key[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s_key[0];
        for (uint32_t q = 1; q < kInitGrid / 64u; ++q) b = s_key[q] > b ? s_key[q] : b;
        *key = b;
    }
}

hipError_t launch_init_pass_cells(const uint32_t *tie, const uint8_t *occ_bits, const float4 *lab_table, Centroid *cent,
                                  uint32_t j, int do_pass, float *dist, void *init_scratch, unsigned long long *band_key,
                                  const uint32_t *pick_rgba, const float *lut, hipStream_t st)
{
    InitRecord *records = (InitRecord *)init_scratch;
    InitSlot *slots = (InitSlot *)(records + kCells);
    if (band_key) {
        hipLaunchKernelGGL(k_init_fused<false>, dim3(kInitGrid), dim3(kInitBlock), 0, st, tie, occ_bits, lab_table, cent, j, 1,
                           dist, records, slots, pick_rgba, lut);
        hipLaunchKernelGGL(k_init_reduce_slots, dim3(1), dim3(kInitGrid), 0, st, slots, j, band_key);
    } else {
        hipLaunchKernelGGL(k_init_fused<true>, dim3(kInitGrid), dim3(kInitBlock), 0, st, tie, occ_bits, lab_table, cent, j, do_pass,
                           dist, records, slots, pick_rgba, lut);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Several centroids per launch over the cells, exactly (whole image on one device; round 6).
//
// A pass only LOWERS running distances.  Let R1 > R2 > ... be the largest CELL records (each the largest key of its cell's
// colours, so they come from distinct cells X1, X2, ...) after the sweeps so far; centroid c = colour(R1) is the next pick
// (plus_plus_init.wgsl:172-181).  Candidate R_i is the pick AFTER the picks P made in this launch so far, without sweeping first,
// when (a) no pick lowers it -- cie94(colour(R_i), p) >= dist(R_i) for every p in P, the very comparison the sweep would make --
// and (b) every colour that may lie above it today ends below it: such colours live in the cells X_m, m < i, only (every other
// cell's record is below R_i), and after the sweeps a colour u of X_m has dist(u) <= min_p cie94(u, p) <= min_p FAR(X_m, p),
// FAR^2 = the largest dL^2 + da^2 + db^2 between p and the cell's Lab box (CIE94's divisors are >= 1, so it never exceeds that;
// the comparison leaves 1 % for the roundings of the literal form).  A candidate that fails (a) is SKIPPED, not the end of the
// list: it sits next to a pick -- the usual case, neighbouring cells share a far corner of colour space -- and (b) then disposes
// of its whole cell for the candidates after it.  The first failure of (b) ends the launch's picks.  Keys are distinct (they name
// pixels); a candidate with distance 0 is never taken this way (Candidate(0, 0.0) = pixel 0).
// Every workgroup leaves the kCellMulti largest records of its 128 cells (a ROW, sorted); the next launch merges the 256 rows in
// every workgroup.  What a row does not show is at most its last entry, so the merged list is exact above the largest last entry
// of all rows (`floor`): candidates at or below it are not used.  One launch =
//   [merge -> up to kCellMulti picks from the kCellCands largest records] -> [test every cell against each pick -> visit the
//   reached cells, each against the picks that reach it] -> [row]
// cfg3 (8192^2 noise, k = 256): 95 launches (with the empty ones at the end of a chunk) instead of 255, same centroids bit for bit;
// 3.5 -> 3.1 ms -- a launch is no longer a chain of round trips but the visits of ~12 reached cells per workgroup (profiles/NOTES.md round 6).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kCellMulti = 4;          // picks per launch = entries of a workgroup's row
constexpr uint32_t kCellCands = 8;          // candidates a launch examines

struct alignas(16) CellCand {
    unsigned long long key;                 // 0: none
    float L, a, b;                          // Lab of the colour that holds it
    uint32_t box[3];                        // the cell's (L0, L1), (a0, a1), (b0, b1): binary16 pairs rounded outward
};
static_assert(sizeof(CellCand) == 32, "CellCand layout");

size_t init_scratch_bytes()
{
    return sizeof(InitRecord) * (size_t)kCells + sizeof(InitSlot) * 2u * kInitGrid + sizeof(CellCand) * 2u * kInitGrid * kCellMulti + 256u;
}

__device__ __forceinline__ uint32_t pack_box(float lo, float hi)
{
    return (uint32_t)__half_as_ushort(__float2half_rd(lo)) | ((uint32_t)__half_as_ushort(__float2half_ru(hi)) << 16);
}

// the largest dL^2 + da^2 + db^2 between (L, a, b) and a point of the box
__device__ __forceinline__ float box_far2(const uint32_t *box, float L, float a, float b)
{
    float far[3];
    const float v[3] = {L, a, b};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float lo = __half2float(__ushort_as_half((unsigned short)(box[q] & 0xFFFFu)));
        const float hi = __half2float(__ushort_as_half((unsigned short)(box[q] >> 16)));
        far[q] = fmaxf(fabsf(hi - v[q]), fabsf(v[q] - lo));
    }
    return far[0] * far[0] + far[1] * far[1] + far[2] * far[2];
}

__device__ __forceinline__ CellCand no_cand()
{
    CellCand o; o.key = 0ull; o.L = 0.0f; o.a = 0.0f; o.b = 0.0f; o.box[0] = 0u; o.box[1] = 0u; o.box[2] = 0u;
    return o;
}

// min(m, cie94(colour v, pick c)) -- v = (L, a, b, chroma) of the Lab table, c likewise.  Most colours of a reached cell are NOT
// lowered (the reach test is a bound over the whole cell), so the literal distance with its two IEEE divides and two square roots
// is evaluated only where the ordering key does not settle it: key and literal^2 are two float evaluations of one real quantity T
// (kmg_math.h: |key - T| <= 63u T, + 12u with hardware reciprocals; |literal^2 - T| <= 66u T), hence
//     key (1 - 2^-10) > m^2   =>   T > m^2 (1 + 2^-11)   =>   literal^2 > m^2   =>   sqrtf(literal^2) >= m   =>   the minimum is m.
__device__ __forceinline__ float lowered_by(float m, const float4 &v, const float4 &c)
{
    const PixelTerms t = pixel_terms_fast(v.x, v.y, v.z, v.w);
    if (!(cie94_key(t, c.x, c.y, c.z, c.w) * (1.0f - 0.0009765625f) > m * m))
        m = fminf(m, cie94_c(v.x, v.y, v.z, v.w, c.x, c.y, c.z, c.w));
    return m;
}

// init_visit_part against the picks of `mask` (bit r: s_pick[r]); first: the map starts at 1e6, nothing to read
template <int N>
__device__ __forceinline__ unsigned long long init_visit_part_multi(uint32_t cell, uint32_t part, uint32_t lane, bool first,
                                                                    const float4 *s_pick, uint32_t mask,
                                                                    const uint32_t *__restrict__ tie, const uint8_t *__restrict__ occ_bits,
                                                                    const float4 *__restrict__ lab_table, float *__restrict__ dist,
                                                                    float4 &lab_out)
{
    constexpr uint32_t kParts = 8u / N;
    const uint32_t within = part * (kCellColours / kParts) + lane * N;
    const uint32_t base = cell * kCellColours + within;
    const uint32_t occ = ((uint32_t)occ_bits[(uint64_t)cell * 64u + (within >> 3)] >> (within & 7u)) & ((1u << N) - 1u);
    float4 v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = lab_table[base + q];
    uint32_t t[N];
    float m[N];
    if (N == 4) {
        const uint4 tt = *reinterpret_cast<const uint4 *>(tie + base);
        t[0] = tt.x; t[1] = tt.y; t[2 % N] = tt.z; t[3 % N] = tt.w;
        float4 d = make_float4(1000000.0f, 1000000.0f, 1000000.0f, 1000000.0f);   // kmeans++_calc_diff.wgsl:26-30
        if (!first) d = *reinterpret_cast<const float4 *>(dist + base);
        m[0] = d.x; m[1] = d.y; m[2 % N] = d.z; m[3 % N] = d.w;
    } else {
        const uint2 tt = *reinterpret_cast<const uint2 *>(tie + base);
        t[0] = tt.x; t[1] = tt.y;
        float2 d = make_float2(1000000.0f, 1000000.0f);
        if (!first) d = *reinterpret_cast<const float2 *>(dist + base);
        m[0] = d.x; m[1] = d.y;
    }
    for (uint32_t mm = mask; mm; mm &= mm - 1u) {
        const float4 c = s_pick[__builtin_ctz(mm)];
#pragma unroll
        for (int q = 0; q < N; ++q)
            if ((occ >> q) & 1u) m[q] = lowered_by(m[q], v[q], c);
    }
    uint32_t md = 0u;
#pragma unroll
    for (int q = 0; q < N; ++q)
        if ((occ >> q) & 1u) md = max(md, float_to_bits(m[q]));
    if (occ || first) {
        if (N == 4) *reinterpret_cast<float4 *>(dist + base) = make_float4(m[0], m[1], m[2 % N], m[3 % N]);
        else *reinterpret_cast<float2 *>(dist + base) = make_float2(m[0], m[1]);
    }
    const uint32_t wmd = wave_max_u32(md);
    uint32_t low1 = 0u;
    lab_out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (occ && md == wmd) {
#pragma unroll
        for (int q = 0; q < N; ++q)
            if (((occ >> q) & 1u) && float_to_bits(m[q]) == wmd && t[q] > low1) { low1 = t[q]; lab_out = v[q]; }
    }
    const uint32_t wlow1 = wave_max_u32(low1);
    return (low1 == wlow1 && low1 != 0u) ? (((unsigned long long)wmd << 32) | (unsigned long long)(wlow1 - 1u)) : 0ull;
}

__global__ __launch_bounds__(kInitBlock) void k_init_cells_multi(const uint32_t *__restrict__ tie, const uint8_t *__restrict__ occ_bits,
                                                                 const float4 *__restrict__ lab_table, Centroid *__restrict__ cent,
                                                                 uint32_t k, uint32_t launch, float *__restrict__ dist,
                                                                 InitRecord *__restrict__ records, CellCand *__restrict__ rows,
                                                                 uint32_t *__restrict__ count, const uint32_t *__restrict__ rgba,
                                                                 const float *__restrict__ lut)
{
    constexpr uint32_t kWaves = kInitBlock / 64, kRowWaves = kInitGrid / 64, kOwn = kWaves * 8;
    __shared__ CellCand s_top[kRowWaves * kCellCands];
    __shared__ CellCand s_best[kCellCands];
    __shared__ unsigned long long s_floor[kRowWaves];
    __shared__ float4 s_pick[kCellMulti];
    __shared__ uint32_t s_npick;
    __shared__ CellCand s_met[kOwn];                               // the records of the workgroup's cells as they stand
    __shared__ uint2 s_list[kOwn];                                 // (cell, slot | picks that reach it << 8)
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_pkey[kWaves];
    __shared__ float4 s_plab[kWaves];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool first = launch == 1u;

    // what the test needs does not depend on the picks: requested before them
    const uint32_t slot = wv * 8u + (lane & 7u);
    const bool tester = lane < 8u;
    InitRecord rec;                                                // (lanes 8 .. 31: the same eight cells again, one pick each)
    rec.key = 0ull; rec.cell = kNoCell;
    if (lane < 8u * kCellMulti) rec = records[slot_work_index(blockIdx.x, slot)];
    if (threadIdx.x == 0) s_count = 0u;

    uint32_t have = 0u;
    if (first) {
        // centroid 0 is there (k_init_first): nothing to pick
        if (threadIdx.x == 0) {
            const Centroid c0 = cent[0];
            s_pick[0] = make_float4(c0.L, c0.a, c0.b, c0.C);
            s_npick = 1u;
            if (blockIdx.x == 0) count[1] = 1u;
        }
    } else {
        const CellCand *prev = rows + (uint64_t)((launch - 1u) & 1u) * kInitGrid * kCellMulti;
        CellCand row[kCellMulti];
#pragma unroll
        for (uint32_t e = 0; e < kCellMulti; ++e) {
            row[e] = no_cand();
            if (threadIdx.x < kInitGrid) row[e] = prev[(uint64_t)threadIdx.x * kCellMulti + e];
        }
        have = count[(launch - 1u) & 1u];
        if (have >= k) {                                           // the table is complete: an empty launch of a chunk
            if (blockIdx.x == 0 && threadIdx.x == 0) count[launch & 1u] = have;
            return;
        }
        if (wv < kRowWaves) {
            // (what the rows hide is at most their last entries)
            const unsigned long long fl = wave_max_u64(row[kCellMulti - 1].key);
            if (lane == 0u) s_floor[wv] = fl;
            // the wave's kCellCands largest heads, one at a time: the lane that holds it hands it over and moves its list up
#pragma unroll
            for (uint32_t r = 0; r < kCellCands; ++r) {
                const unsigned long long g = wave_max_u64(row[0].key);
                if (g == 0ull) {
                    if (lane == 0u) s_top[wv * kCellCands + r] = no_cand();
                } else if (row[0].key == g) {
                    s_top[wv * kCellCands + r] = row[0];
#pragma unroll
                    for (uint32_t e = 0; e + 1 < kCellMulti; ++e) row[e] = row[e + 1];
                    row[kCellMulti - 1] = no_cand();
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < kRowWaves * kCellCands) {
            const CellCand mine = s_top[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < kRowWaves * kCellCands; ++q) {
                const unsigned long long o = s_top[q].key;
                rank += (o > mine.key || (o == mine.key && q < threadIdx.x)) ? 1u : 0u;
            }
            if (rank < kCellCands) s_best[rank] = mine;
        }
        __syncthreads();
        if (wv == 0u) {
            // lane 8 x + q: how far candidate x's cell reaches from candidate q's colour, and what a sweep against candidate x's
            // colour would make of candidate q's distance; then lane x < 8 collects its row of each
            float far2[kCellCands], de[kCellCands];
            {
                const CellCand cx = s_best[lane >> 3], cq = s_best[lane & 7u];
                const float f = box_far2(cx.box, cq.L, cq.a, cq.b);
                const float d = cie94(cq.L, cq.a, cq.b, cx.L, cx.a, cx.b);
#pragma unroll
                for (uint32_t q = 0; q < kCellCands; ++q) {
                    far2[q] = __shfl(f, (int)(((lane & 7u) << 3) | q));
                    de[q] = __shfl(d, (int)(((lane & 7u) << 3) | q));
                }
            }
            unsigned long long floor_key = s_floor[0];
#pragma unroll
            for (uint32_t q = 1; q < kRowWaves; ++q) floor_key = s_floor[q] > floor_key ? s_floor[q] : floor_key;
            const uint32_t room = min(kCellMulti, k - have);
            uint32_t picked = 1u, npick = 1u, order = 0u;
            bool stop = (uint32_t)(s_best[0].key >> 32) == 0u;     // (all distances zero: one pick, pixel 0)
#pragma unroll
            for (uint32_t i = 1; i < kCellCands; ++i) {
                const unsigned long long ki = s_best[i].key;
                const float di = __uint_as_float((uint32_t)(ki >> 32));
                if (npick >= room || ki <= floor_key || (uint32_t)(ki >> 32) == 0u) stop = true;
                float m = 3.0e38f;
#pragma unroll
                for (uint32_t q = 0; q < kCellCands; ++q)
                    if ((picked >> q) & 1u) m = fminf(m, far2[q]);
                // (b) the cells of the candidates before this one end below it
                if (__ballot(lane < i && !(m * 1.01f < di * di)) != 0ull) stop = true;
                // (a) no pick lowers it
                const bool lowered = __ballot(lane < kCellCands && ((picked >> lane) & 1u) && !(de[i] >= di)) != 0ull;
                if (!stop && !lowered) { picked |= 1u << i; order |= i << (4u * npick); ++npick; }
            }
            if (lane < npick) {
                const CellCand o = s_best[(order >> (4u * lane)) & 15u];
                float4 v = make_float4(o.L, o.a, o.b, 0.0f);
                if ((uint32_t)(o.key >> 32) == 0u) {
                    // Candidate(0, 0.0): every distance is 0 -> pixel 0
                    const uint32_t px = rgba[0];
                    linear100_to_lab(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u], v.x, v.y, v.z);
                }
                v.w = chroma(v.y, v.z);
                s_pick[lane] = v;
                if (blockIdx.x == 0) { Centroid c; c.L = v.x; c.a = v.y; c.b = v.z; c.C = v.w; cent[have + lane] = c; }
            }
            if (lane == 0u) {
                s_npick = npick;
                if (blockIdx.x == 0) count[launch & 1u] = have + npick;
            }
        }
    }
    __syncthreads();
    const uint32_t npick = s_npick;
    if (have + npick >= k) return;                                 // the last centroids: no distances are needed any more

    // ---- the test: which picks reach the cell?  lane 8 r + c: cell c of the wave against pick r ----
    {
        bool reached = false;
        const uint32_t r = lane >> 3;
        if (r < npick && rec.cell != kNoCell) {
            reached = true;
            if (!first) {
                const float4 c = s_pick[r];
                reached = cie94_lower_bound(rec.cb, c.x, c.y, c.z, c.w) < __uint_as_float((uint32_t)(rec.key >> 32));
            }
        }
        const unsigned long long hits = __ballot(reached);
        if (tester) {
            CellCand met = no_cand();
            if (rec.cell != kNoCell) {
                const uint32_t mine = (uint32_t)(hits >> lane);
                const uint32_t mask = (mine & 1u) | ((mine >> 7) & 2u) | ((mine >> 14) & 4u) | ((mine >> 21) & 8u);
                met.key = rec.key; met.L = rec.lab.x; met.a = rec.lab.y; met.b = rec.lab.z;
                met.box[0] = pack_box(rec.cb.L0, rec.cb.L1); met.box[1] = pack_box(rec.cb.a0, rec.cb.a1); met.box[2] = pack_box(rec.cb.b0, rec.cb.b1);
                if (mask) s_list[atomicAdd(&s_count, 1u)] = make_uint2(rec.cell, slot | (mask << 8));
            }
            s_met[slot] = met;
        }
    }
    __syncthreads();
    const uint32_t count_ = s_count;
    const bool eager = count_ <= kWaves;

    if (count_ <= 8u) {
        // few cells: 4 (count <= 4) or 2 waves per cell, each a part of its colours
        const uint32_t shift = count_ <= 4u ? 2u : 1u;
        const uint32_t my = wv >> shift, part = wv & ((1u << shift) - 1u);
        unsigned long long pk = 0ull;
        float4 pl = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (my < count_) {
            const uint2 e = s_list[my];
            pk = shift == 2u ? init_visit_part_multi<2>(e.x, part, lane, first, s_pick, e.y >> 8, tie, occ_bits, lab_table, dist, pl)
                             : init_visit_part_multi<4>(e.x, part, lane, first, s_pick, e.y >> 8, tie, occ_bits, lab_table, dist, pl);
        }
        if (lane == 0u) s_pkey[wv] = 0ull;
        if (pk != 0ull) { s_pkey[wv] = pk; s_plab[wv] = pl; }
        __syncthreads();
        if (part == 0u && my < count_ && lane == 0u) {
            uint32_t w = wv;
            for (uint32_t q = 1; q < (1u << shift); ++q) if (s_pkey[wv + q] > s_pkey[w]) w = wv + q;
            const unsigned long long key = s_pkey[w];
            const float4 lab = s_plab[w];
            const uint32_t sl = s_list[my].y & 255u;
            InitRecord *r = records + slot_work_index(blockIdx.x, sl);
            r->key = key;
            r->lab = lab;
            s_met[sl].key = key; s_met[sl].L = lab.x; s_met[sl].a = lab.y; s_met[sl].b = lab.z;
        }
    }
    uint32_t idx = count_ <= 8u ? count_ : wv;
    uint2 ent = idx < count_ ? s_list[idx] : make_uint2(0u, 0u);
    uint32_t base = ent.x * kCellColours + lane * 8u;
    uint32_t occ = occ_bits[(uint64_t)ent.x * 64u + lane];
    float4 d0 = make_float4(1000000.0f, 1000000.0f, 1000000.0f, 1000000.0f), d1 = d0;     // kmeans++_calc_diff.wgsl:26-30
    if (!first) { d0 = *reinterpret_cast<const float4 *>(dist + base); d1 = *reinterpret_cast<const float4 *>(dist + base + 4); }
    while (idx < count_) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = lab_table[base + q];
        uint4 t0 = make_uint4(0u, 0u, 0u, 0u), t1 = t0;
        if (eager) { t0 = *reinterpret_cast<const uint4 *>(tie + base); t1 = *reinterpret_cast<const uint4 *>(tie + base + 4); }
        const uint32_t idx_n = idx + kWaves;
        const uint2 ent_n = idx_n < count_ ? s_list[idx_n] : ent;
        const uint32_t base_n = ent_n.x * kCellColours + lane * 8u;
        const uint32_t occ_n = occ_bits[(uint64_t)ent_n.x * 64u + lane];
        float4 d0_n = d0, d1_n = d1;
        if (!first) { d0_n = *reinterpret_cast<const float4 *>(dist + base_n); d1_n = *reinterpret_cast<const float4 *>(dist + base_n + 4); }

        float m[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        for (uint32_t mm = ent.y >> 8; mm; mm &= mm - 1u) {
            const float4 c = s_pick[__builtin_ctz(mm)];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((occ >> q) & 1u) m[q] = lowered_by(m[q], v[q], c);
        }
        uint32_t md = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if ((occ >> q) & 1u) md = max(md, float_to_bits(m[q]));
        if (occ || first) {
            *reinterpret_cast<float4 *>(dist + base) = make_float4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<float4 *>(dist + base + 4) = make_float4(m[4], m[5], m[6], m[7]);
        }
        const uint32_t wmd = wave_max_u32(md);
        uint32_t low1 = 0u;
        float4 best_lab = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (occ && md == wmd) {
            if (!eager) { t0 = *reinterpret_cast<const uint4 *>(tie + base); t1 = *reinterpret_cast<const uint4 *>(tie + base + 4); }
            const uint32_t t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (((occ >> q) & 1u) && float_to_bits(m[q]) == wmd && t[q] > low1) { low1 = t[q]; best_lab = v[q]; }
            }
        }
        const uint32_t wlow1 = wave_max_u32(low1);
        if (low1 == wlow1 && low1 != 0u) {
            const unsigned long long key = ((unsigned long long)wmd << 32) | (unsigned long long)(wlow1 - 1u);
            const uint32_t sl = ent.y & 255u;
            InitRecord *r = records + slot_work_index(blockIdx.x, sl);
            r->key = key;
            r->lab = best_lab;
            s_met[sl].key = key; s_met[sl].L = best_lab.x; s_met[sl].a = best_lab.y; s_met[sl].b = best_lab.z;
        }
        idx = idx_n; ent = ent_n; base = base_n; occ = occ_n; d0 = d0_n; d1 = d1_n;
    }
    // ---- the workgroup's row: the kCellMulti largest of its cells' records ----
    __syncthreads();
    if (wv < kOwn / 64u) {
        CellCand mine = s_met[threadIdx.x];
#pragma unroll
        for (uint32_t r = 0; r < kCellMulti; ++r) {
            const unsigned long long g = wave_max_u64(mine.key);
            if (g == 0ull) {
                if (lane == 0u) s_top[wv * kCellMulti + r] = no_cand();
            } else if (mine.key == g) {
                s_top[wv * kCellMulti + r] = mine;
                mine.key = 0ull;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < (kOwn / 64u) * kCellMulti) {
        const CellCand mine = s_top[threadIdx.x];
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t q = 0; q < (kOwn / 64u) * kCellMulti; ++q) {
            const unsigned long long o = s_top[q].key;
            rank += (o > mine.key || (o == mine.key && q < threadIdx.x)) ? 1u : 0u;
        }
        if (rank < kCellMulti) rows[((uint64_t)(launch & 1u) * kInitGrid + blockIdx.x) * kCellMulti + rank] = mine;
    }
}

hipError_t launch_init_cells_multi(const uint32_t *tie, const uint8_t *occ_bits, const float4 *lab_table, Centroid *cent, uint32_t k,
                                   uint32_t launch, float *dist, void *init_scratch, const uint32_t *pick_rgba, const float *lut,
                                   hipStream_t st)
{
    InitRecord *records = (InitRecord *)init_scratch;
    CellCand *rows = (CellCand *)((InitSlot *)(records + kCells) + 2u * kInitGrid);
    uint32_t *count = (uint32_t *)(rows + 2u * kInitGrid * kCellMulti);
    hipLaunchKernelGGL(k_init_cells_multi, dim3(kInitGrid), dim3(kInitBlock), 0, st, tie, occ_bits, lab_table, cent, k, launch, dist,
                       records, rows, count, pick_rgba, lut);
    return hipGetLastError();
}

const uint32_t *init_cells_multi_count(const void *init_scratch, uint32_t launch)
{
    const InitRecord *records = (const InitRecord *)init_scratch;
    const CellCand *rows = (const CellCand *)((const InitSlot *)(records + kCells) + 2u * kInitGrid);
    return (const uint32_t *)(rows + 2u * kInitGrid * kCellMulti) + (launch & 1u);
}

// ------------------------------------------------------------------------------------------
// per-cell and per-sub-cell sums of the image (once per image): one wave per cell, 8 colours per lane
// (lane l holds colours 8 l .. 8 l + 7, i.e. an eighth of sub-cell l >> 3)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cell_aggregates(const uint32_t *__restrict__ hist,
                                                            const float4 *__restrict__ lab_table,
                                                            int64_t *__restrict__ agg, int64_t *__restrict__ sub_agg,
                                                            uint8_t *__restrict__ occ_bits)
{
    const uint32_t cell = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t base = cell * kCellColours + lane * 8;
    const uint4 c0 = *reinterpret_cast<const uint4 *>(hist + base);
    const uint4 c1 = *reinterpret_cast<const uint4 *>(hist + base + 4);
    float4 lab8[8];                                                // all loads in flight together (one latency)
#pragma unroll
    for (int q = 0; q < 8; ++q) lab8[q] = lab_table[base + q];
    const uint32_t cnt[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    long long s[4] = {0, 0, 0, 0};
    uint32_t occ = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (cnt[q]) {
            const float4 v = lab8[q];
            const long long m = (long long)cnt[q];
            s[0] += m * (long long)lab_fix(v.x);
            s[1] += m * (long long)lab_fix(v.y);
            s[2] += m * (long long)lab_fix(v.z);
            s[3] += m;
            occ |= 1u << q;
        }
    }
    occ_bits[(uint64_t)cell * 64u + lane] = (uint8_t)occ;          // bit q of byte (colour >> 3) = colour occupied
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int off = 1; off < 8; off <<= 1) s[j] += __shfl_xor(s[j], off, 64);     // the sub-cell's 8 lanes
    if ((lane & 7u) == 0)
        for (int j = 0; j < 4; ++j) sub_agg[4ull * (cell * 8u + (lane >> 3)) + j] = s[j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int off = 8; off < 64; off <<= 1) s[j] += __shfl_xor(s[j], off, 64);
    if (lane == 0)
        for (int j = 0; j < 4; ++j) agg[4ull * cell + j] = s[j];
}

hipError_t launch_cell_aggregates(const uint32_t *hist, const float4 *lab_table, int64_t *agg, int64_t *sub_agg, uint8_t *occ_bits,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(k_cell_aggregates, dim3(kCells / (kBlock / 64)), dim3(kBlock), 0, st, hist, lab_table, agg, sub_agg, occ_bits);
    return hipGetLastError();
}

// dense list of the occupied cells in ascending order (once per image): work[0] = count, work[1..] = cells; behind it the
// hot cells (kmg_table.h): the cells holding at least n_pixels >> j pixels for the smallest share (largest j <= 10) that
// leaves at most kHotMax of them, in ascending order -- none if together they hold less than a tenth of the pixels
__global__ __launch_bounds__(1024) void k_work_list(const int64_t *__restrict__ agg, uint32_t *__restrict__ work, uint64_t n_pixels)
{
    __shared__ uint32_t s_n[1024];
    __shared__ uint32_t s_above[5];                                // cells with at least n >> (6 + 2 j) pixels... see thresholds below
    __shared__ unsigned long long s_hot_pixels;
    constexpr uint32_t PER = kCells / 1024;                        // 32 consecutive cells per thread
    const uint32_t c0 = threadIdx.x * PER;
    uint32_t occupied = 0;                                         // bit i = cell c0 + i has pixels
    long long cnt[PER];
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) cnt[i] = agg[4ull * (c0 + i) + 3];      // all 32 loads in flight
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) occupied |= (cnt[i] != 0 ? 1u : 0u) << i;
    if (threadIdx.x < 5) s_above[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_hot_pixels = 0ull;
    __syncthreads();
    // thresholds n / 64, n / 128, n / 256, n / 512, n / 1024 (a cell of a noise image holds n / 32768)
    uint32_t above[5] = {0u, 0u, 0u, 0u, 0u};
    if (n_pixels) {
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i)
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) above[j] += ((uint64_t)cnt[i] >= ((n_pixels >> (6u + j)) | 1ull)) ? 1u : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 5; ++j)
            if (above[j]) atomicAdd(&s_above[j], above[j]);
    }
    __syncthreads();
    uint32_t jsel = 0;                                              // the lowest threshold that still leaves at most kHotMax cells
    for (uint32_t j = 1; j < 5; ++j)
        if (s_above[j] <= kHotMax) jsel = j;
    const uint64_t thr = n_pixels ? ((n_pixels >> (6u + jsel)) | 1ull) : ~0ull;
    uint32_t hot = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) hot |= ((uint64_t)cnt[i] >= thr ? 1u : 0u) << i;
    const uint32_t mine = (uint32_t)__builtin_popcount(occupied) | ((uint32_t)__builtin_popcount(hot) << 16);
    s_n[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t a = threadIdx.x >= off ? s_n[threadIdx.x - off] : 0u;
        __syncthreads();
        s_n[threadIdx.x] += a;
        __syncthreads();
    }
    const uint32_t before = s_n[threadIdx.x] - mine;
    uint32_t at = 1u + (before & 0xFFFFu);
    for (uint32_t i = 0; i < PER; ++i)
        if ((occupied >> i) & 1u) work[at++] = c0 + i;
    uint32_t hat = before >> 16;                                    // hot cells before this thread's
    unsigned long long hot_px = 0ull;
    for (uint32_t i = 0; i < PER; ++i)
        if ((hot >> i) & 1u) {
            if (hat < kHotMax) { work[kCells + 2u + hat] = c0 + i; hot_px += (unsigned long long)cnt[i]; }
            ++hat;
        }
    if (hot_px) atomicAdd(&s_hot_pixels, hot_px);
    __syncthreads();
    if (threadIdx.x == 1023) {
        work[0] = s_n[1023] & 0xFFFFu;
        const uint32_t n_hot = min(s_n[1023] >> 16, kHotMax);
        work[kCells + 1u] = (s_hot_pixels * 10ull >= n_pixels) ? n_hot : 0u;
    }
}

__global__ __launch_bounds__(1024) void k_work_share(const uint32_t *__restrict__ work, uint32_t part, uint32_t parts,
                                                     uint32_t *__restrict__ out)
{
    // share `part` = the occupied cells with index in [kCells part / parts, kCells (part + 1) / parts): equal ranges of the
    // CUBE, so that the shares' pieces of the label tables are equal, contiguous chunks (an in-place all-gather)
    const uint32_t n = work[0];
    const uint32_t c_lo = (uint32_t)((uint64_t)kCells * part / parts), c_hi = (uint32_t)((uint64_t)kCells * (part + 1u) / parts);
    auto lower_bound = [&](uint32_t cell) {                        // first position whose cell is >= `cell` (the list ascends)
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (work[1u + mid] < cell) lo = mid + 1u; else hi = mid; }
        return lo;
    };
    const uint32_t lo = lower_bound(c_lo), hi = lower_bound(c_hi);
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < hi - lo; i += gridDim.x * 1024u) out[1u + i] = work[1u + lo + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = hi - lo;
}

hipError_t launch_work_share(const uint32_t *work, uint32_t part, uint32_t parts, uint32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_work_share, dim3(32), dim3(1024), 0, st, work, part, parts, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// What a cost model may know about an image before anything is built for it: kProbeSamples pixels at equal strides, counted per
// 8x8x8 cell in LDS (ONE workgroup, ~10 us).  out[0] = cells the sample occupies, out[1] = samples that fall into crowded cells
// (a cell with at least 1 / 560 of the samples: the dark corner of a photograph), out[2] = samples taken.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kProbeSamples = 16384;

__global__ __launch_bounds__(1024) void k_sparsity_probe(const uint32_t *__restrict__ rgba, uint64_t n, uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_cnt[kCells / 2];                         // two 16-bit counters per word
    __shared__ uint32_t s_tot[2];
    for (uint32_t i = threadIdx.x; i < kCells / 2; i += 1024u) s_cnt[i] = 0u;
    if (threadIdx.x < 2u) s_tot[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t samples = n < kProbeSamples ? (uint32_t)n : kProbeSamples;
    const uint64_t stride = n / samples;
    for (uint32_t i = threadIdx.x; i < samples; i += 1024u) {
        const uint32_t cell = colour_index(rgba[(uint64_t)i * stride]) >> 9;
        atomicAdd(&s_cnt[cell >> 1], (cell & 1u) ? 0x10000u : 1u);
    }
    __syncthreads();
    const uint32_t crowded = samples / 560u + 1u;
    uint32_t occ = 0u, hot = 0u;
    for (uint32_t i = threadIdx.x; i < kCells / 2; i += 1024u) {
        const uint32_t lo = s_cnt[i] & 0xFFFFu, hi = s_cnt[i] >> 16;
        occ += (lo != 0u) + (hi != 0u);
        hot += (lo >= crowded ? lo : 0u) + (hi >= crowded ? hi : 0u);
    }
    occ = wave_add_u32(occ);                                         // (per-lane <= 32 cells, <= 16384 samples: no overflow)
    hot = wave_add_u32(hot);
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s_tot[0], occ); atomicAdd(&s_tot[1], hot); }
    __syncthreads();
    if (threadIdx.x == 0u) { out[0] = s_tot[0]; out[1] = s_tot[1]; out[2] = samples; }
}

hipError_t launch_sparsity_probe(const uint32_t *rgba, uint64_t n, uint32_t *out3, hipStream_t st)
{
    hipLaunchKernelGGL(k_sparsity_probe, dim3(1), dim3(1024), 0, st, rgba, n, out3);
    return hipGetLastError();
}

hipError_t launch_work_list(const int64_t *agg, uint32_t *work, uint64_t n_pixels, hipStream_t st)
{
    hipLaunchKernelGGL(k_work_list, dim3(1), dim3(1024), 0, st, agg, work, n_pixels);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// labels[i] = colour_labels[colour_index(pixel i)]   (find_centroid.wgsl:43 output)
// ------------------------------------------------------------------------------------------
// Three levels, coarse to fine; an entry is the common label of the occupied colours below it, or
// kSubMixed:
//   8x8x8 cells   (32768 x u16 = 64 KiB)  staged in LDS by every workgroup  -> no memory request
//   4x4x4 cells   (2^18 x u16 = 512 KiB)  L2 resident
//   single colour (2^24 x u8/u16)         Infinity Cache / HBM, only for pixels of mixed sub-cells
// Pixel and label streams are non-temporal so they do not evict the tables from L2.
constexpr int kLabelBlock = 1024;

template <typename LabelT>
__global__ __launch_bounds__(kLabelBlock) void k_labels(const uint32_t *__restrict__ rgba, uint64_t n,
                                                        const LabelT *__restrict__ colour_labels,
                                                        const uint16_t *__restrict__ sub_table,
                                                        const uint32_t *__restrict__ pal, uint32_t k,
                                                        uint32_t *__restrict__ labels, int aligned)
{
    __shared__ uint16_t s_cell[kCells];
    __shared__ uint32_t s_pal[3072];                       // KMG_MAX_K output colours (pal != NULL)
    if (pal)
        for (uint32_t i = threadIdx.x; i < k; i += kLabelBlock) s_pal[i] = pal[i];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(sub_table + kSubCells);
        uint4 *dst = reinterpret_cast<uint4 *>(s_cell);
        for (uint32_t i = threadIdx.x; i < kCells / 8; i += kLabelBlock) dst[i] = src[i];
    }
    __syncthreads();
    constexpr uint64_t TILE = (uint64_t)kLabelBlock * 8;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t ci[8];
        uint64_t i0[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            i0[g] = tile * TILE + (uint64_t)g * (kLabelBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4_stream(rgba, i0[g], n, aligned != 0, px);
#pragma unroll
            for (int q = 0; q < 4; ++q) ci[g * 4 + q] = colour_index(px[q]);
        }
        uint32_t lab[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) lab[p] = (uint32_t)s_cell[ci[p] >> 9];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (lab[p] == kSubMixed) lab[p] = (uint32_t)sub_table[ci[p] >> 6];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (lab[p] == kSubMixed) lab[p] = (uint32_t)colour_labels[ci[p]];
        if (pal) {                                           // swap.wgsl + lab_to_rgb: colour of the label
#pragma unroll
            for (int p = 0; p < 8; ++p) lab[p] = s_pal[lab[p]];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) store4_stream(labels, i0[g], n, aligned != 0, lab + g * 4);
    }
}

// k <= 256: the whole first two levels live in LDS.  One u32 per 8x8x8 cell (128 KiB, one workgroup
// of 1024 threads per CU): two labels and, for each, the mask of 4x4x4 sub-cells all of whose occupied
// colours carry it.  Pixels of other sub-cells (mixed, or a third label) go to the per-colour table.
// Divergent global gathers retire at ~1 per 2 clocks per CU whether they hit L2 or not
// (tools/gather_rate.hip), so resolving ~2/3 of the pixels from LDS is what pays.
// HOT: the image has hot cells (kmg_table.h): their 512 per-colour labels are copied into LDS behind the pair table and their
// pair entries (in this workgroup's LDS copy only) become [block : 8][block : 8][direction code 127] = "look the colour up in
// block `block`" -- the per-colour table itself, so right by construction.  A hot cell whose entry already resolves every
// colour (one label) keeps it: the cube pass does not write per-colour labels for single-candidate cells.
constexpr uint32_t kHotCode = 127u;                                // direction codes 0..124 are planes
constexpr size_t kLabelLdsPlain = sizeof(uint32_t) * (kCells + 256 + 128);
constexpr size_t kLabelLdsHot = kLabelLdsPlain + (size_t)kHotMax * kCellColours + sizeof(uint32_t) * 64;

// (The knock-outs this kernel carried in rounds 3-5 -- gathers from a window / from LDS / none, a cheaper colour index, three
// quarters of the pixel bytes, half the pair table, a second-plane extension entry, one more LDS read per pixel -- have served
// their measurements (profiles/r03_label_knock.txt, r05_label_knock.txt: ~0.5 us per vector instruction and pixel, +4.5 us per
// LDS read) and live on as tools/experiments/r06_label_pass_knockouts.patch.)
template <bool HOT>
__global__ __launch_bounds__(kLabelBlock) void k_labels_pairs(const uint32_t *__restrict__ rgba, uint64_t n,
                                                              const uint8_t *__restrict__ colour_labels,
                                                              const uint32_t *__restrict__ pair_table,
                                                              const uint32_t *__restrict__ pal, uint32_t k,
                                                              uint32_t *__restrict__ labels, int aligned,
                                                              const uint32_t *__restrict__ hot, int64_t *__restrict__ tail_sums,
                                                              CubeTail tail)
{
    // Static LDS with the small tables in front: every table's address is then a compile-time constant that fits the ds_read
    // offset field (as dynamic LDS the pair table's base was a v_add per pixel, the direction table's another).
    __shared__ uint32_t s_label_lds[(HOT ? kLabelLdsHot : kLabelLdsPlain) / sizeof(uint32_t)];
    // The cube pass's tail (kmg_table.h CubeTail) when that pass has no launch left to carry it (k_cube_small): the sums are
    // final -- the cube pass is a launch of its own before this one -- and nothing in this pass reads them or the centroids.
    if (tail.acc_out && blockIdx.x == gridDim.x - 1u) {
        for (uint32_t i = threadIdx.x; i < 4u * k; i += kLabelBlock) tail.acc_out[i] = tail_sums[i];
        if (tail.do_update) update_centroids(tail_sums, k, tail.convergence, tail.cent, tail.n_converged, s_label_lds, kLabelBlock);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 4u * k; i += kLabelBlock) tail_sums[i] = 0;      // ready for the next pass
        __syncthreads();
    }
    constexpr uint32_t kPairWords = kCells;
    uint32_t *s_pal = s_label_lds, *s_dir = s_label_lds + 256, *s_pair = s_label_lds + 256 + 128;
    uint8_t *s_hot = reinterpret_cast<uint8_t *>(s_label_lds + 256 + 128 + kPairWords);
    uint32_t *s_hcell = reinterpret_cast<uint32_t *>(s_hot + (size_t)kHotMax * kCellColours);
    if (pal && threadIdx.x < k) s_pal[threadIdx.x] = pal[threadIdx.x];
    if (threadIdx.x < 128) s_dir[threadIdx.x] = threadIdx.x < kPairDirs ? pair_dir_word(threadIdx.x) : 0u;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(pair_table);
        uint4 *dst = reinterpret_cast<uint4 *>(s_pair);
        for (uint32_t i = threadIdx.x; i < kPairWords / 4; i += kLabelBlock) dst[i] = src[i];
    }
    __syncthreads();
    if (HOT) {
        const uint32_t n_hot = min(hot[0], kHotMax);
        if (threadIdx.x < n_hot) {
            const uint32_t cell = hot[1u + threadIdx.x];
            const uint32_t e = s_pair[cell];
            const bool one_label = (e & 0xFFu) == ((e >> 8) & 0xFFu) && (e >> 23) == 0u;     // A == B, tlo = 0, w = 0
            s_hcell[threadIdx.x] = one_label ? 0xFFFFFFFFu : cell;
        }
        __syncthreads();
        const uint32_t *lab32 = reinterpret_cast<const uint32_t *>(colour_labels);
        uint32_t *hot32 = reinterpret_cast<uint32_t *>(s_hot);
        for (uint32_t i = threadIdx.x; i < n_hot * (kCellColours / 4u); i += kLabelBlock) {
            const uint32_t cell = s_hcell[i / (kCellColours / 4u)];
            if (cell != 0xFFFFFFFFu) hot32[i] = lab32[(uint64_t)cell * (kCellColours / 4u) + (i % (kCellColours / 4u))];
        }
        if (threadIdx.x < n_hot && s_hcell[threadIdx.x] != 0xFFFFFFFFu)
            s_pair[s_hcell[threadIdx.x]] = threadIdx.x | (threadIdx.x << 8) | (kHotCode << 16);
        __syncthreads();
    }
    constexpr uint64_t TILE = (uint64_t)kLabelBlock * 8;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t ci[8], xyz[8];
        uint64_t i0[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            i0[g] = tile * TILE + (uint64_t)g * (kLabelBlock * 4) + (uint64_t)threadIdx.x * 4;
            uint32_t px[4];
            load4_stream(rgba, i0[g], n, aligned != 0, px);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ci[g * 4 + q] = colour_index(px[q]);
                xyz[g * 4 + q] = (px[q] & 0x00070707u) | 0x01000000u;   // (r & 7, g & 7, b & 7, 1)
            }
        }
        uint32_t lab[8];
        bool fine[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t e = s_pair[ci[p] >> 9];
            const uint32_t code = (e >> 16) & 127u;
            const uint32_t dirw = s_dir[code];
            const int proj = __builtin_amdgcn_sdot4((int)xyz[p], (int)dirw, 0, false);
            const int tlo = (int)((e >> 23) & 63u), w = (int)(e >> 29);
            const bool inA = proj < tlo, inB = proj >= tlo + w + (w == 7 ? 64 : 0);
            lab[p] = inA ? (e & 0xFFu) : ((e >> 8) & 0xFFu);
            fine[p] = !(inA || inB);
            if (HOT && code == kHotCode) {                           // (tlo = w = 0: the line above said "B", not "fine")
                lab[p] = (uint32_t)s_hot[(e & 0xFFu) * kCellColours + (ci[p] & (kCellColours - 1u))];
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (fine[p]) lab[p] = (uint32_t)colour_labels[ci[p]];
        if (pal) {
#pragma unroll
            for (int p = 0; p < 8; ++p) lab[p] = s_pal[lab[p]];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) store4_stream(labels, i0[g], n, aligned != 0, lab + g * 4);
    }
}

hipError_t launch_labels(const uint32_t *rgba, uint64_t n, const void *colour_labels, const uint16_t *sub_table,
                         uint32_t k, const uint32_t *pal, uint32_t *labels, hipStream_t st, uint32_t reserve_cus, const uint32_t *hot,
                         const CubeTail *tail, int64_t *tail_sums)
{
    const CubeTail tl = (tail && tail_sums && k <= 256) ? *tail : CubeTail();
    if (k <= 256) {
        const uint64_t tiles = (n + kLabelBlock * 8 - 1) / (kLabelBlock * 8);
        const uint32_t all = device_info().cus;
        // 1 workgroup per CU (its 130 KiB of LDS see to that); reserve_cus fewer workgroups than CUs -- which leaves that
        // many CUs without one only as long as nothing else occupies them first
        const uint32_t cus = all - (reserve_cus < all / 2u ? reserve_cus : all / 2u);
        const uint32_t grid = (uint32_t)(tiles < cus ? (tiles ? tiles : 1) : cus);
        const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 &&
                             (reinterpret_cast<uintptr_t>(labels) & 15u) == 0) ? 1 : 0;
        const uint32_t *pairs = reinterpret_cast<const uint32_t *>(sub_table + kSubCells + kCells);
        if (hot)
            hipLaunchKernelGGL(k_labels_pairs<true>, dim3(grid), dim3(kLabelBlock), 0, st, rgba, n,
                               (const uint8_t *)colour_labels, pairs, pal, k, labels, aligned, hot, tail_sums, tl);
        else
            hipLaunchKernelGGL(k_labels_pairs<false>, dim3(grid), dim3(kLabelBlock), 0, st, rgba, n,
                               (const uint8_t *)colour_labels, pairs, pal, k, labels, aligned, hot, tail_sums, tl);
        return hipGetLastError();
    }
    const uint64_t tiles = (n + kLabelBlock * 8 - 1) / (kLabelBlock * 8);
    const uint32_t grid = (uint32_t)(tiles < 512 ? (tiles ? tiles : 1) : 512);   // 2 workgroups per CU
    const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 &&
                         (reinterpret_cast<uintptr_t>(labels) & 15u) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_labels<uint16_t>, dim3(grid), dim3(kLabelBlock), 0, st, rgba, n,
                       (const uint16_t *)colour_labels, sub_table, pal, k, labels, aligned);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// ordered-dither output pass with candidate pruning (mix_colors.wgsl:50-83)
//
// The dithered colour of a pixel is Lab(colour) + off, off = one of the 16 Bayer offsets of the
// pass, so its nearest centroid depends on (colour, Bayer index) only.  For every (cell, Bayer
// index) k_offset_candidates builds the same kind of conservative candidate mask as
// k_cube_stage (kmg_cube.hip), from the static cell bounds shifted by the offset; the output pass then
// scans, per pixel, only the candidates of its (cell, Bayer index) instead of all k centroids.
// ------------------------------------------------------------------------------------------
// bounds of the per-pixel terms after adding `off` to L, a and b of every colour of the cell
__device__ __forceinline__ CellBounds shifted_bounds(const CellBounds &cb, float off)
{
    CellBounds s;
    s.L0 = cb.L0 + off; s.L1 = cb.L1 + off;                     // rounding is monotone: still bounds
    s.a0 = cb.a0 + off; s.a1 = cb.a1 + off;
    s.b0 = cb.b0 + off; s.b1 = cb.b1 + off;
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

// masks[(cell * 16 + bayer) * words + w]; one wave per cell.
// Two levels (k <= 256).  (1) lanes strided over the centroids, against the cell's bounds WIDENED by the whole range of the
// 16 offsets: lo*_j <= lo_j(off) and hi*_j >= hi_j(off) for every offset (the same monotone operations on wider intervals),
// so S = {j : lo*_j <= (1 + slack) min_m hi*_m} contains the candidate set of every offset -- and the centroid that attains
// min_m hi_m(off), which is a candidate of its offset.  (2) only the members of S (a dozen) are bounded per offset, several
// offsets per wave step (lane = (offset, member)); the test per offset is the one-level test restricted to S, and since the
// minimiser of the upper bounds is in S the threshold is the same: identical masks, ~10x fewer interval evaluations.
__global__ __launch_bounds__(kBlock) void k_offset_candidates(const CellBounds *__restrict__ bounds,
                                                              const Centroid *__restrict__ cent, uint32_t k,
                                                              float threshold, uint64_t *__restrict__ masks)
{
    __shared__ uint32_t s_list_all[kBlock / 64][64];
    __shared__ unsigned long long s_out_all[kBlock / 64][16 * 4];
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t *s_list = s_list_all[wv];
    unsigned long long *s_out = s_out_all[wv];
    const uint32_t cell = blockIdx.x * (kBlock / 64) + wv;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t words = (k + 63u) / 64u;
    const CellBounds cb = bounds[cell];
    const float my_off = threshold * (bayer16(lane & 15u) / 16.0f - 0.5f);   // as k_apply computes it
    const CellBounds mine = shifted_bounds(cb, my_off);
    Centroid c[4];
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) {
        const uint32_t j = w * 64 + lane;
        c[w] = cent[j < k ? j : 0u];
    }
    bool two_level = words <= 4u;
    if (two_level) {
        // ---- (1) the superset S ----
        float off_lo = my_off, off_hi = my_off;
        for (int o = 8; o > 0; o >>= 1) { off_lo = fminf(off_lo, __shfl_xor(off_lo, o, 64)); off_hi = fmaxf(off_hi, __shfl_xor(off_hi, o, 64)); }
        CellBounds wide;
        wide.L0 = cb.L0 + off_lo; wide.L1 = cb.L1 + off_hi;         // rounding is monotone: bounds of every shifted interval
        wide.a0 = cb.a0 + off_lo; wide.a1 = cb.a1 + off_hi;
        wide.b0 = cb.b0 + off_lo; wide.b1 = cb.b1 + off_hi;
        {
            float ma, Ma, mb, Mb;
            abs_range(wide.a0, wide.a1, 0.0f, ma, Ma);
            abs_range(wide.b0, wide.b1, 0.0f, mb, Mb);
            wide.C0 = chroma(ma, mb);
            wide.C1 = chroma(Ma, Mb);
            const PixelTerms lo = pixel_terms_c(0.0f, 0.0f, 0.0f, wide.C0), hi = pixel_terms_c(0.0f, 0.0f, 0.0f, wide.C1);
            wide.wC0 = hi.wC; wide.wC1 = lo.wC;
            wide.wH0 = hi.wH; wide.wH1 = lo.wH;
        }
        float U = 3.0e38f, lo[4];
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) {
            lo[w] = 3.0e38f;
            if (w < words) {
                const KeyRange r = key_range(wide, c[w].L, c[w].a, c[w].b, c[w].C);
                lo[w] = r.lo;
                if (w * 64 + lane < k) U = fminf(U, r.hi);
            }
        }
        for (int o = 32; o > 0; o >>= 1) U = fminf(U, __shfl_xor(U, o, 64));
        U = mask_threshold(U);
        uint32_t n_s = 0;
        unsigned long long sw[4];
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) {
            sw[w] = __ballot(w < words && w * 64 + lane < k && lo[w] <= U);
            n_s += (uint32_t)__builtin_popcountll(sw[w]);
        }
        if (n_s > 64u) {
            two_level = false;                                      // (never seen: the one-level scan below handles it)
        } else {
            // ---- (2) lane (o, p): member p of S against offset o ----
            uint32_t base = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                if ((sw[w] >> lane) & 1ull) s_list[base + bits_below_lane(sw[w])] = w * 64u + lane;
                base += (uint32_t)__builtin_popcountll(sw[w]);
            }
            s_out[lane] = 0ull;
            __builtin_amdgcn_wave_barrier();
            const uint32_t G = n_s <= 8u ? 8u : (n_s <= 16u ? 16u : (n_s <= 32u ? 32u : 64u));    // lanes per offset
            const uint32_t per_step = 64u / G;                                                   // offsets per wave step
            const uint32_t p = lane & (G - 1u);
            const uint32_t j = s_list[p < n_s ? p : 0u];
            const Centroid cj = cent[j];
            for (uint32_t o0 = 0; o0 < 16u; o0 += per_step) {
                const uint32_t bi = o0 + lane / G;
                // the shifted bounds of Bayer index bi were derived once, by lane bi (12 permutes instead of two square roots
                // and four divisions per lane and step)
                CellBounds sb;
                sb.L0 = __shfl(mine.L0, (int)bi, 64); sb.L1 = __shfl(mine.L1, (int)bi, 64);
                sb.a0 = __shfl(mine.a0, (int)bi, 64); sb.a1 = __shfl(mine.a1, (int)bi, 64);
                sb.b0 = __shfl(mine.b0, (int)bi, 64); sb.b1 = __shfl(mine.b1, (int)bi, 64);
                sb.C0 = __shfl(mine.C0, (int)bi, 64); sb.C1 = __shfl(mine.C1, (int)bi, 64);
                sb.wC0 = __shfl(mine.wC0, (int)bi, 64); sb.wC1 = __shfl(mine.wC1, (int)bi, 64);
                sb.wH0 = __shfl(mine.wH0, (int)bi, 64); sb.wH1 = __shfl(mine.wH1, (int)bi, 64);
                const KeyRange r = key_range(sb, cj.L, cj.a, cj.b, cj.C);
                float Ug = p < n_s ? r.hi : 3.0e38f;
                for (uint32_t o = 1; o < G; o <<= 1) Ug = fminf(Ug, __shfl_xor(Ug, (int)o, 64));
                Ug = mask_threshold(Ug);
                const bool keep = p < n_s && r.lo <= Ug;
                if (keep) atomicOr(&s_out[bi * 4u + (j >> 6)], 1ull << (j & 63u));
            }
            __builtin_amdgcn_wave_barrier();
            // the cell's 16 x words mask words are contiguous
            const uint32_t bi = lane >> 2, w = lane & 3u;
            const unsigned long long mw = s_out[lane];
            if (w < words) masks[((uint64_t)cell * 16u + bi) * words + w] = mw;
        }
    }
    if (two_level) return;
    for (uint32_t bi = 0; bi < 16u; ++bi) {
        CellBounds sb;
        sb.L0 = lane_value(mine.L0, bi); sb.L1 = lane_value(mine.L1, bi);
        sb.a0 = lane_value(mine.a0, bi); sb.a1 = lane_value(mine.a1, bi);
        sb.b0 = lane_value(mine.b0, bi); sb.b1 = lane_value(mine.b1, bi);
        sb.C0 = lane_value(mine.C0, bi); sb.C1 = lane_value(mine.C1, bi);
        sb.wC0 = lane_value(mine.wC0, bi); sb.wC1 = lane_value(mine.wC1, bi);
        sb.wH0 = lane_value(mine.wH0, bi); sb.wH1 = lane_value(mine.wH1, bi);
        uint64_t *out = masks + ((uint64_t)cell * 16u + bi) * words;
        float U = 3.0e38f;
        if (words <= 4u) {
            // k <= 256: one evaluation per centroid, the lower bounds wait in registers for U
            float lo[4];
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                lo[w] = 0.0f;
                if (w < words) {
                    const KeyRange r = key_range(sb, c[w].L, c[w].a, c[w].b, c[w].C);
                    lo[w] = r.lo;
                    if (w * 64 + lane < k) U = fminf(U, r.hi);
                }
            }
            for (int o = 32; o > 0; o >>= 1) U = fminf(U, __shfl_xor(U, o, 64));
            U = mask_threshold(U);                               // keep what can still be a near-tie (kmg_math.h)
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                const unsigned long long m = __ballot(w * 64 + lane < k && lo[w] <= U);
                if (w < words && lane == 0) out[w] = m;
            }
            continue;
        }
        for (uint32_t j = lane; j < k; j += 64) {
            const Centroid ce = cent[j];
            U = fminf(U, key_range(sb, ce.L, ce.a, ce.b, ce.C).hi);
        }
        for (int o = 32; o > 0; o >>= 1) U = fminf(U, __shfl_xor(U, o, 64));
        U = mask_threshold(U);
        for (uint32_t w = 0; w < words; ++w) {
            const uint32_t j = w * 64 + lane;
            bool keep = false;
            if (j < k) {
                const Centroid ce = cent[j];
                keep = key_range(sb, ce.L, ce.a, ce.b, ce.C).lo <= U;
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) out[w] = m;
        }
    }
}

hipError_t launch_offset_candidates(const CellBounds *bounds, const Centroid *cent, uint32_t k, float threshold,
                                    uint64_t *masks, hipStream_t st)
{
    hipLaunchKernelGGL(k_offset_candidates, dim3(kCells / (kBlock / 64)), dim3(kBlock), 0, st, bounds, cent, k,
                       threshold, masks);
    return hipGetLastError();
}

// out[i] = pal[arg-min over the candidates of (cell, Bayer index) of the key of Lab(pixel) + off];
// the running minimum starts at the sentinel's distance with index k (mix_colors.wgsl:73-80)
// WORDS = number of 64-bit mask words when it is 1, 2 or 4 (k <= 64, 128, 256: all words of the four pixels are
// requested up front), 0 = any k (first word up front, the others on demand)
template <int WORDS>
__global__ __launch_bounds__(kBlock) void k_dither_pruned(const uint32_t *__restrict__ rgba, uint32_t w, uint64_t n,
                                                          uint32_t row0, const Centroid *__restrict__ cent, uint32_t k,
                                                          const float *__restrict__ lut, const uint32_t *__restrict__ pal,
                                                          float threshold, const uint64_t *__restrict__ masks,
                                                          uint32_t *__restrict__ out, int aligned)
{
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 3u) & ~3u;
    float4 *s_cent = smem4;
    float *s_lut = reinterpret_cast<float *>(smem4 + kpad);
    float *s_off = s_lut + 256;
    s_lut[threadIdx.x] = lut[threadIdx.x];
    stage_centroids(s_cent, cent, k, kpad);
    if (threadIdx.x < 16) s_off[threadIdx.x] = threshold * (bayer16(threadIdx.x) / 16.0f - 0.5f);
    __syncthreads();
    const uint32_t words = (k + 63u) / 64u;
    const float sentinel_C = chroma(10000.0f, 10000.0f);

    constexpr uint64_t TILE = (uint64_t)kBlock * 4;
    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t i0 = tile * TILE + (uint64_t)threadIdx.x * 4;
        uint32_t px[4];
        load4_stream(rgba, i0, n, aligned != 0, px);
        const uint32_t i32 = (uint32_t)i0;                          // n < 2^32
        uint32_t gy = i32 / w, gx = i32 - gy * w;
        gy += row0;
        constexpr int UP = WORDS > 0 ? WORDS : 1;
        uint32_t slot[4];
        unsigned long long m0[4][UP];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t bi = (gx & 3u) + ((gy & 3u) << 2);
            const uint32_t cell = (((px[q] >> 3) & 31u) << 10) | (((px[q] >> 11) & 31u) << 5) | ((px[q] >> 19) & 31u);
            slot[q] = cell * 16u + bi;
#pragma unroll
            for (int u = 0; u < UP; ++u) m0[q][u] = masks[(uint64_t)slot[q] * words + u];   // gathers in flight during the Lab conversion
            gx += 1;
            if (gx == w) { gx = 0; gy += 1; }
        }
        uint32_t res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float L, a, b;
            px_to_lab(s_lut, px[q], L, a, b);
            const float off = s_off[slot[q] & 15u];
            L = L + off; a = a + off; b = b + off;                   // mix_colors.wgsl:72
            // (the weights only order the candidates: hardware reciprocals, near-ties are settled by the literal distance)
            const PixelTerms pt = kLiteralArgmin ? pixel_terms_fast(L, a, b, chroma(a, b)) : pixel_terms(L, a, b);
            float best = cie94_key(pt, 10000.0f, 10000.0f, 10000.0f, sentinel_C), second = 3.0e38f;
            uint32_t idx = k;
            auto scan_word = [&](unsigned long long m, uint32_t wd) {
                while (m) {
                    const uint32_t j = wd * 64 + (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const float4 c = s_cent[j];
                    const float d = cie94_key(pt, c.x, c.y, c.z, c.w);
                    if (kLiteralArgmin) second = __builtin_amdgcn_fmed3f(d, best, second);
                    if (d < best) { best = d; idx = j; }
                }
            };
#pragma unroll
            for (int u = 0; u < UP; ++u) scan_word(m0[q][u], (uint32_t)u);
            if (WORDS == 0)
                for (uint32_t wd = 1; wd < words; ++wd) scan_word(masks[(uint64_t)slot[q] * words + wd], wd);
            if (kLiteralArgmin && second <= tie_threshold(best)) {
                // near-tie (kmg_math.h): mix_colors.wgsl:73-80 with the literal distance, the sentinel first
                const float thr = tie_threshold(best);
                float lb = cie94_c(pt.L, pt.a, pt.b, pt.C, 10000.0f, 10000.0f, 10000.0f, sentinel_C);
                uint32_t li = k;
                auto rescan_word = [&](unsigned long long m, uint32_t wd) {
                    while (m) {
                        const uint32_t j = wd * 64 + (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        const float4 c = s_cent[j];
                        if (cie94_key(pt, c.x, c.y, c.z, c.w) <= thr) {
                            const float d = cie94_c(pt.L, pt.a, pt.b, pt.C, c.x, c.y, c.z, c.w);
                            if (d < lb) { lb = d; li = j; }
                        }
                    }
                };
#pragma unroll
                for (int u = 0; u < UP; ++u) rescan_word(m0[q][u], (uint32_t)u);
                if (WORDS == 0)
                    for (uint32_t wd = 1; wd < words; ++wd) rescan_word(masks[(uint64_t)slot[q] * words + wd], wd);
                idx = li;
            }
            res[q] = pal[idx];
        }
        store4_stream(out, i0, n, aligned != 0, res);
    }
}

// The same pass with the pixels of a tile SORTED by the length of their candidate list (k <= 256).
// On noise the lanes of a wave see unrelated slots, and a wave's candidate loop runs to its longest list: ~12
// iterations for 5.3 candidates on average (64-entry palette), ~2x the work that is needed.  Here a workgroup
// converts a tile of 256 x PPT pixels, counting-sorts them in LDS by popcount(mask) (32 buckets: LDS atomics for the
// rank, one wave for the prefix), and then scans them in sorted order -- the 64 pixels of a wave have (nearly) equal
// list lengths; the results go back through LDS to their pixels' places so that the store stays coalesced.  What
// is computed per pixel is exactly what k_dither_pruned computes: only the lane a pixel is scanned by changes.
// LDS per workgroup: PPT x 256 x (12 B Lab + 8 B x WORDS mask + 2 B origin + 4 B result) = 26.5 KiB (WORDS = 1, PPT = 4).
constexpr uint32_t kSortCounters = 512;      // 32 list lengths x 16 counters (= 2 per thread of a 256-thread workgroup)

template <int WORDS, int PPT>
__global__ __launch_bounds__(kBlock) void k_dither_sorted(const uint32_t *__restrict__ rgba, uint32_t w, uint64_t n,
                                                          uint32_t row0, const Centroid *__restrict__ cent, uint32_t k,
                                                          const float *__restrict__ lut, const uint32_t *__restrict__ pal,
                                                          float threshold, const uint64_t *__restrict__ masks,
                                                          uint32_t *__restrict__ out, int aligned)
{
    constexpr uint32_t TILE = kBlock * PPT;
    extern __shared__ float4 smem4[];
    const uint32_t kpad = (k + 3u) & ~3u;
    float4 *s_cent = smem4;
    unsigned long long *s_m = reinterpret_cast<unsigned long long *>(smem4 + kpad);    // [WORDS][TILE]
    float *s_L = reinterpret_cast<float *>(s_m + (size_t)WORDS * TILE), *s_a = s_L + TILE, *s_b = s_a + TILE;
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_b + TILE);
    float *s_lut = reinterpret_cast<float *>(s_out + TILE);
    float *s_off = s_lut + 256;
    uint32_t *s_pal = reinterpret_cast<uint32_t *>(s_off + 16);                          // k + 1 output colours
    uint32_t *s_hist = s_pal + ((k + 4u) & ~3u), *s_wsum = s_hist + kSortCounters;   // counts, then (in place) bases
    uint16_t *s_org = reinterpret_cast<uint16_t *>(s_wsum + 4);
    s_lut[threadIdx.x] = lut[threadIdx.x];
    stage_centroids(s_cent, cent, k, kpad);
    for (uint32_t i = threadIdx.x; i <= k; i += kBlock) s_pal[i] = pal[i];
    if (threadIdx.x < 16) s_off[threadIdx.x] = threshold * (bayer16(threadIdx.x) / 16.0f - 0.5f);
    for (uint32_t i = threadIdx.x; i < kSortCounters; i += kBlock) s_hist[i] = 0u;
    __syncthreads();
    const float sentinel_C = chroma(10000.0f, 10000.0f);
    const uint32_t lane = threadIdx.x & 63u;

    const uint64_t tiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        // ---- 1. convert this thread's pixels, fetch their masks, rank them inside their bucket ----
        const uint64_t i0 = tile * TILE + (uint64_t)threadIdx.x * PPT;
        uint32_t px[4] = {0u, 0u, 0u, 0u};
        if (PPT == 4) load4_stream(rgba, i0, n, aligned != 0, px);
        else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) px[q] = i0 + q < n ? __builtin_nontemporal_load(rgba + i0 + q) : 0u;
        }
        const uint32_t i32 = (uint32_t)i0;                          // n < 2^32
        uint32_t gy = i32 / w, gx = i32 - gy * w;
        gy += row0;
        unsigned long long m[PPT][WORDS];
        uint32_t bi[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            bi[q] = (gx & 3u) + ((gy & 3u) << 2);
            const uint32_t cell = (((px[q] >> 3) & 31u) << 10) | (((px[q] >> 11) & 31u) << 5) | ((px[q] >> 19) & 31u);
            const uint64_t slot = (uint64_t)cell * 16u + bi[q];
#pragma unroll
            for (int u = 0; u < WORDS; ++u) m[q][u] = i0 + q < n ? masks[slot * WORDS + u] : 0ull;
            gx += 1;
            if (gx == w) { gx = 0; gy += 1; }
        }
        float L[PPT], a[PPT], b[PPT];
        uint32_t pc[PPT], rank[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            px_to_lab(s_lut, px[q], L[q], a[q], b[q]);
            const float off = s_off[bi[q]];
            L[q] = L[q] + off; a[q] = a[q] + off; b[q] = b[q] + off;   // mix_colors.wgsl:72
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < WORDS; ++u) c += (uint32_t)__builtin_popcountll(m[q][u]);
            // 16 counters per list length, chosen by the lane: 64 lanes with ~10 distinct lengths would otherwise queue
            // up on ~10 addresses (same-address LDS atomics are served one after the other)
            pc[q] = (c < 31u ? c : 31u) * 16u + (lane & 15u);
            rank[q] = atomicAdd(&s_hist[pc[q]], 1u);
        }
        __syncthreads();
        // ---- 2. counter bases (exclusive prefix over the 512 counters: two per thread), records to their sorted places ----
        {
            const uint32_t c0 = s_hist[2u * threadIdx.x], c1 = s_hist[2u * threadIdx.x + 1u];
            uint32_t incl = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
                if ((int)lane >= o) incl += t;
            }
            if (lane == 63u) s_wsum[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (uint32_t v = 0; v < kBlock / 64u; ++v) before += v < (threadIdx.x >> 6) ? s_wsum[v] : 0u;
            const uint32_t excl = before + incl - (c0 + c1);
            s_hist[2u * threadIdx.x] = excl;
            s_hist[2u * threadIdx.x + 1u] = excl + c0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const uint32_t pos = s_hist[pc[q]] + rank[q];
            s_L[pos] = L[q]; s_a[pos] = a[q]; s_b[pos] = b[q];
#pragma unroll
            for (int u = 0; u < WORDS; ++u) s_m[(size_t)u * TILE + pos] = m[q][u];
            s_org[pos] = (uint16_t)(threadIdx.x * PPT + q);
        }
        __syncthreads();
        s_hist[2u * threadIdx.x] = 0u;                              // the bases are consumed: counters of the next tile
        s_hist[2u * threadIdx.x + 1u] = 0u;
        // ---- 3. scan in sorted order: wave v of batch q takes places [q 256 + 64 v, + 64) ----
#pragma unroll 1
        for (int q = 0; q < PPT; ++q) {
            const uint32_t pos = (uint32_t)q * kBlock + threadIdx.x;
            const float pL = s_L[pos], pa = s_a[pos], pb = s_b[pos];
            unsigned long long mm[WORDS];
#pragma unroll
            for (int u = 0; u < WORDS; ++u) mm[u] = s_m[(size_t)u * TILE + pos];
            // (the weights only order the candidates: hardware reciprocals, near-ties are settled by the literal distance)
            const PixelTerms pt = kLiteralArgmin ? pixel_terms_fast(pL, pa, pb, chroma(pa, pb)) : pixel_terms(pL, pa, pb);
            float best = cie94_key(pt, 10000.0f, 10000.0f, 10000.0f, sentinel_C), second = 3.0e38f;
            uint32_t idx = k;
#pragma unroll
            for (int u = 0; u < WORDS; ++u) {
                unsigned long long mw = mm[u];
                while (mw) {
                    const uint32_t j = (uint32_t)u * 64u + (uint32_t)__builtin_ctzll(mw);
                    mw &= mw - 1;
                    const float4 c = s_cent[j];
                    const float d = cie94_key(pt, c.x, c.y, c.z, c.w);
                    if (kLiteralArgmin) second = __builtin_amdgcn_fmed3f(d, best, second);
                    if (d < best) { best = d; idx = j; }
                }
            }
            if (kLiteralArgmin && second <= tie_threshold(best)) {
                // near-tie (kmg_math.h): mix_colors.wgsl:73-80 with the literal distance, the sentinel first
                const float thr = tie_threshold(best);
                float lb = cie94_c(pt.L, pt.a, pt.b, pt.C, 10000.0f, 10000.0f, 10000.0f, sentinel_C);
                uint32_t li = k;
#pragma unroll
                for (int u = 0; u < WORDS; ++u) {
                    unsigned long long mw = mm[u];
                    while (mw) {
                        const uint32_t j = (uint32_t)u * 64u + (uint32_t)__builtin_ctzll(mw);
                        mw &= mw - 1;
                        const float4 c = s_cent[j];
                        if (cie94_key(pt, c.x, c.y, c.z, c.w) <= thr) {
                            const float d = cie94_c(pt.L, pt.a, pt.b, pt.C, c.x, c.y, c.z, c.w);
                            if (d < lb) { lb = d; li = j; }
                        }
                    }
                }
                idx = li;
            }
            s_out[s_org[pos]] = s_pal[idx];
        }
        __syncthreads();
        // ---- 4. results back in pixel order ----
        if (PPT == 4) {
            const uint4 r = *reinterpret_cast<const uint4 *>(s_out + threadIdx.x * 4u);
            const uint32_t res[4] = {r.x, r.y, r.z, r.w};
            store4_stream(out, i0, n, aligned != 0, res);
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                if (i0 + q < n) __builtin_nontemporal_store(s_out[threadIdx.x * PPT + q], out + i0 + q);
        }
        // (the next tile's first LDS writes are its ranks: the counters were cleared before the last barrier; its records
        // and results are rewritten only after its own barriers)
    }
}

hipError_t launch_dither_pruned(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0, const Centroid *cent,
                                uint32_t k, const float *lut, const uint32_t *pal, float threshold,
                                const uint64_t *masks, uint32_t *out, hipStream_t st)
{
    const uint64_t n = (uint64_t)w * rows;
    const uint64_t tiles = (n + kBlock * 4 - 1) / (kBlock * 4);
    const uint32_t grid = (uint32_t)(tiles < 8192 ? (tiles ? tiles : 1) : 8192);
    const uint32_t kpad = (k + 3u) & ~3u;
    const size_t lds = sizeof(float4) * kpad + (256 + 16) * sizeof(float);
    const int aligned = ((reinterpret_cast<uintptr_t>(rgba) & 15u) == 0 &&
                         (reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? 1 : 0;
#define KMG_DP(W) hipLaunchKernelGGL(k_dither_pruned<W>, dim3(grid), dim3(kBlock), lds, st, rgba, w, n, row0, cent, k, lut, \
                                     pal, threshold, masks, out, aligned)
    const uint32_t n_words = (k + 63u) / 64u;
    static const bool sorted = tools_env_int(KMG_TOOLS_ENV("KMG_DITHER_SORT"), 1) != 0;
    if (sorted && n_words == 1u) {
        // the pixels of a tile sorted by candidate-list length (k_dither_sorted): 8192^2, 64-entry palette 0.95 -> 0.88 ms.
        // With more mask words the records outgrow the LDS a well-occupied CU can give them (k = 256, 2 pixels per
        // thread: 1.56 -> 1.72 ms); k <= 256 takes the byte lists of kmg_lists.hip.
        const uint32_t ppt = 4u;
        const uint32_t tile = kBlock * ppt;
        const uint64_t tiles_s = (n + tile - 1) / tile;
        const uint32_t grid_s = (uint32_t)(tiles_s < 4096 ? (tiles_s ? tiles_s : 1) : 4096);
        const size_t lds_s = sizeof(float4) * kpad + (size_t)tile * (8u * n_words + 12u + 4u + 2u) + (256 + 16) * sizeof(float) +
                             sizeof(uint32_t) * (((k + 4u) & ~3u) + 512u + 4u);
        hipLaunchKernelGGL((k_dither_sorted<1, 4>), dim3(grid_s), dim3(kBlock), lds_s, st, rgba, w, n, row0, cent, k, lut, pal, threshold,
                           masks, out, aligned);
        return hipGetLastError();
    }
    if (n_words == 1) KMG_DP(1); else if (n_words == 2) KMG_DP(2); else if (n_words == 4) KMG_DP(4); else KMG_DP(0);
#undef KMG_DP
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// meld output pass with candidate pruning (mix_colors.wgsl:29-48 two_closest_colors, :85-90 meld)
//
// The pass needs the two closest centroids of a pixel under the literal CIE94, found by one ordered
// scan with `<` comparisons.  Removing from that scan any centroid whose distance exceeds the second
// smallest one changes neither slot of the result (such a centroid can only sit in a slot until
// something smaller arrives, and never displaces an equal value), so it is enough to scan, in index
// order, a superset of the centroids within the second smallest distance: per cell, those whose
// lower bound is not above the second smallest upper bound.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float cie94_upper_bound(const CellBounds &cb, float L2, float a2, float b2, float C2)
{
    float mL, ML, ma, Ma, mb, Mb, mC, MC;
    abs_range(cb.L0, cb.L1, L2, mL, ML);
    abs_range(cb.a0, cb.a1, a2, ma, Ma);
    abs_range(cb.b0, cb.b1, b2, mb, Mb);
    abs_range(cb.C0, cb.C1, C2, mC, MC);
    const float SC = 1.0f + 0.045f * cb.C0, SH = 1.0f + 0.015f * cb.C0;      // the smallest divisors
    const float dH = sqrtf(fmaxf((Ma * Ma) + (Mb * Mb) - (mC * mC), 0.0f));
    const float tL = ML / 1.0f, tC = MC / SC, tH = dH / SH;
    return sqrtf(tL * tL + tC * tC + tH * tH);
}

// (m1 <= m2) <- the two smallest of {m1, m2, o1, o2}, o1 <= o2
__device__ __forceinline__ void merge_two_smallest(float &m1, float &m2, float o1, float o2)
{
    const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
    m2 = fminf(hi, fminf(m2, o2));
    m1 = lo;
}

// masks[cell * words + w]: one wave per cell, lanes strided over the centroids (k >= 2)
__global__ __launch_bounds__(kBlock) void k_meld_candidates(const CellBounds *__restrict__ bounds,
                                                            const Centroid *__restrict__ cent, uint32_t k,
                                                            uint64_t *__restrict__ masks)
{
    const uint32_t cell = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t words = (k + 63u) / 64u;
    const CellBounds cb = bounds[cell];
    float u1 = 3.0e38f, u2 = 3.0e38f;                              // the two smallest upper bounds
    for (uint32_t j = lane; j < k; j += 64) {
        const Centroid c = cent[j];
        merge_two_smallest(u1, u2, cie94_upper_bound(cb, c.L, c.a, c.b, c.C), 3.0e38f);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o1 = __shfl_xor(u1, off, 64), o2 = __shfl_xor(u2, off, 64);
        merge_two_smallest(u1, u2, o1, o2);
    }
    uint64_t *out = masks + (uint64_t)cell * words;
    for (uint32_t w = 0; w < words; ++w) {
        const uint32_t j = w * 64 + lane;
        bool keep = false;
        if (j < k) {
            const Centroid c = cent[j];
            keep = cie94_lower_bound(cb, c.L, c.a, c.b, c.C) <= u2;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) out[w] = m;
    }
}

hipError_t launch_meld_candidates(const CellBounds *bounds, const Centroid *cent, uint32_t k, uint64_t *masks, hipStream_t st)
{
    hipLaunchKernelGGL(k_meld_candidates, dim3(kCells / (kBlock / 64)), dim3(kBlock), 0, st, bounds, cent, k, masks);
    return hipGetLastError();
}

// test support: violations += #colours whose two closest centroids (mix_colors.wgsl:29-48) differ between
// the scan of all centroids and the scan of the cell's candidates
__global__ __launch_bounds__(kBlock) void k_check_meld_masks(const Centroid *__restrict__ cent, uint32_t k,
                                                             const uint64_t *__restrict__ masks,
                                                             const float *__restrict__ lut,
                                                             unsigned long long *__restrict__ violations)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t cell = blockIdx.x;
    const uint32_t words = (k + 63u) / 64u;
    unsigned long long bad = 0;
    for (uint32_t c = threadIdx.x; c < kCellColours; c += kBlock) {
        float L, a, b;
        colour_to_lab(s_lut, cell * kCellColours + c, L, a, b);
        const float d0 = cie94(L, a, b, 10000.0f, 10000.0f, 10000.0f);
        float dc = d0, ds = d0, pc = d0, ps = d0;
        uint32_t ic = k, is = k, jc = k, js = k;
        for (uint32_t j = 0; j < k; ++j) {
            const Centroid ce = cent[j];
            const float d = cie94(L, a, b, ce.L, ce.a, ce.b);
            if (d < dc) { ds = dc; is = ic; dc = d; ic = j; } else if (d < ds) { ds = d; is = j; }
            const unsigned long long m = masks[(uint64_t)cell * words + j / 64u];
            if ((m >> (j & 63u)) & 1ull) {
                if (d < pc) { ps = pc; js = jc; pc = d; jc = j; } else if (d < ps) { ps = d; js = j; }
            }
        }
        bad += (ic != jc) || (is != js);
    }
    if (bad) atomicAdd(violations, bad);
}

hipError_t launch_check_meld_masks(const Centroid *cent, uint32_t k, const uint64_t *masks, const float *lut,
                                   unsigned long long *violations, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_meld_masks, dim3(kCells), dim3(kBlock), 0, st, cent, k, masks, lut, violations);
    return hipGetLastError();
}

// test support: violations += #(colour, Bayer index) whose brute-force dither arg-min (sentinel
// included) differs from the arg-min over the candidates of its (cell, Bayer index)
__global__ __launch_bounds__(kBlock) void k_check_offset_masks(const Centroid *__restrict__ cent, uint32_t k,
                                                               const uint64_t *__restrict__ masks,
                                                               const float *__restrict__ lut, float threshold,
                                                               unsigned long long *__restrict__ violations)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t cell = blockIdx.x;
    const uint32_t words = (k + 63u) / 64u;
    const float sentinel_C = chroma(10000.0f, 10000.0f);
    unsigned long long bad = 0;
    for (uint32_t c = threadIdx.x; c < kCellColours; c += kBlock) {
        float L0, a0, b0;
        colour_to_lab(s_lut, cell * kCellColours + c, L0, a0, b0);
        for (uint32_t bi = 0; bi < 16u; ++bi) {
            const float off = threshold * (bayer16(bi) / 16.0f - 0.5f);
            const PixelTerms pt = pixel_terms(L0 + off, a0 + off, b0 + off);
            // the arg-min of mix_colors.wgsl:73-80: literal distance (kLiteralArgmin), the sentinel first
            const float start = kLiteralArgmin ? cie94_c(pt.L, pt.a, pt.b, pt.C, 10000.0f, 10000.0f, 10000.0f, sentinel_C)
                                               : cie94_key(pt, 10000.0f, 10000.0f, 10000.0f, sentinel_C);
            float best = start, bestp = start;
            uint32_t idx = k, idxp = k;
            for (uint32_t j = 0; j < k; ++j) {
                const Centroid ce = cent[j];
                const float d = kLiteralArgmin ? cie94_c(pt.L, pt.a, pt.b, pt.C, ce.L, ce.a, ce.b, ce.C)
                                               : cie94_key(pt, ce.L, ce.a, ce.b, ce.C);
                if (d < best) { best = d; idx = j; }
                const unsigned long long m = masks[((uint64_t)cell * 16u + bi) * words + j / 64u];
                if (((m >> (j & 63u)) & 1ull) && d < bestp) { bestp = d; idxp = j; }
            }
            bad += idx != idxp;
        }
    }
    if (bad) atomicAdd(violations, bad);
}

hipError_t launch_check_offset_masks(const Centroid *cent, uint32_t k, const uint64_t *masks, const float *lut,
                                     float threshold, unsigned long long *violations, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_offset_masks, dim3(kCells), dim3(kBlock), 0, st, cent, k, masks, lut, threshold, violations);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// test support: exhaustive check of the cube pass over all 2^24 colours (run without an image, per-colour
// labels of every cell stored)
// violations[0] += #(colour, centroid) pairs with key outside [lo, hi] of the colour's cell or sub-cell
// violations[1] += #colours whose arg-min is not in the cell's mask
// violations[2] += #colours whose per-colour label is not the arg-min
// The arg-min is the reference's: first index of the smallest LITERAL distance (kLiteralArgmin), or of the
// smallest key.
// ------------------------------------------------------------------------------------------
template <typename LabelT>
__global__ __launch_bounds__(kBlock) void k_check_bounds(const CellBounds *__restrict__ bounds,
                                                         const CellBounds *__restrict__ sub_bounds,
                                                         const Centroid *__restrict__ cent, uint32_t k,
                                                         const uint64_t *__restrict__ masks,
                                                         const LabelT *__restrict__ colour_labels,
                                                         const float *__restrict__ lut,
                                                         unsigned long long *__restrict__ violations)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t cell = blockIdx.x;
    const uint32_t words = (k + 63u) / 64u;
    const CellBounds cb = bounds[cell];
    unsigned long long bad_range = 0, bad_mask = 0, bad_label = 0;
    for (uint32_t c = threadIdx.x; c < kCellColours; c += kBlock) {
        float L, a, b;
        colour_to_lab(s_lut, cell * kCellColours + c, L, a, b);
        const PixelTerms pt = pixel_terms(L, a, b);
        const CellBounds sb = sub_bounds[cell * 8u + (c >> 6)];
        float best = kLiteralArgmin ? 100000.0f : 1.0e10f;       // find_centroid.wgsl:29-30
        uint32_t idx = 0;
        for (uint32_t j = 0; j < k; ++j) {
            const Centroid ce = cent[j];
            const float d = cie94_key(pt, ce.L, ce.a, ce.b, ce.C);
            const KeyRange r = key_range(cb, ce.L, ce.a, ce.b, ce.C);
            const KeyRange q = key_range(sb, ce.L, ce.a, ce.b, ce.C);
            if (!(r.lo <= d && d <= r.hi)) ++bad_range;
            if (!(q.lo <= d && d <= q.hi)) ++bad_range;
            const float v = kLiteralArgmin ? cie94_c(pt.L, pt.a, pt.b, pt.C, ce.L, ce.a, ce.b, ce.C) : d;
            if (v < best) { best = v; idx = j; }
        }
        const unsigned long long m = masks[(uint64_t)cell * words + idx / 64u];
        if (!((m >> (idx & 63u)) & 1ull)) ++bad_mask;
        if ((uint32_t)colour_labels[cell * kCellColours + c] != idx) ++bad_label;
    }
    if (bad_range) atomicAdd(violations, bad_range);
    if (bad_mask) atomicAdd(violations + 1, bad_mask);
    if (bad_label) atomicAdd(violations + 2, bad_label);
}

hipError_t launch_check_bounds(const CellBounds *bounds, const CellBounds *sub_bounds, const Centroid *cent, uint32_t k,
                               const uint64_t *masks, const void *colour_labels, const float *lut,
                               unsigned long long *violations, hipStream_t st)
{
    if (k <= 256)
        hipLaunchKernelGGL(k_check_bounds<uint8_t>, dim3(kCells), dim3(kBlock), 0, st, bounds, sub_bounds, cent, k, masks,
                           (const uint8_t *)colour_labels, lut, violations);
    else
        hipLaunchKernelGGL(k_check_bounds<uint16_t>, dim3(kCells), dim3(kBlock), 0, st, bounds, sub_bounds, cent, k, masks,
                           (const uint16_t *)colour_labels, lut, violations);
    return hipGetLastError();
}

}  // namespace kmg
