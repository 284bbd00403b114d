// kmg_table.h -- the colour-table strategy of the Lloyd iteration (internal to libkmeans_hip).
//
// Labels are a pure function of the 24-bit colour (alpha is ignored, rgb_to_lab.wgsl:78) and the
// per-cluster sums are exact integers, so for a large image the iteration can run over the
// image's colour histogram instead of its pixels and still give bit-identical labels, sums and
// centroids:
//   bind (once per image)  : hist[2^24] (u32 counts, cell-major colour order) + per-cell sums
//   per iteration          : 1. k_cube (kmg_cube.hip) -- one wave per 8x8x8 colour cell: a conservative
//                                superset of the centroids that can be the arg-min of any colour in
//                                the cell (float-monotone interval bounds, no epsilons); one candidate
//                                -> the whole cell goes to it (precomputed cell sums).  Otherwise the
//                                same test per 4x4x4 sub-cell with its own bounds (precomputed sub-cell
//                                sums), and only the colours of still undecided sub-cells are scanned,
//                                against their sub-cell's candidates.  Writes label-per-colour (LUT),
//                                the LDS tables of the label pass and the k x 4 sums.
//                             2. k_labels -- labels[i] = LUT[colour(pixel i)]  (4 B in, 4 B out); for
//                                k <= 256 most pixels are resolved from a per-cell plane + slab entry
//                                held in LDS (pair entries below), the rest gathers from the LUT
// The same machinery (static cell bounds, monotone interval evaluation) also serves, on large images,
// the farthest-point initialisation over the image's colours and the candidate-pruned dither and meld
// output passes; all of them are bit-identical to the per-pixel kernels of kmg_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmg_kernels.h"

namespace kmg {

constexpr uint32_t kCells = 32768;        // 32^3 cells of 8x8x8 colours
constexpr uint32_t kCellColours = 512;
constexpr uint32_t kSubCells = kCells * 8;  // 4x4x4 sub-cells (64 colours)
constexpr uint32_t kCubeGrid = 2048;      // workgroups of k_cube (4 waves each, a wave walks cells wave, wave + n_waves, ...)
constexpr uint32_t kMergeRows = 64;       // rows of the partial slab the cube workgroups add their sums into
// "hot" cells of a bound image: the (at most kHotMax) cells that hold the largest share of its pixels, chosen once per image
// when together they hold at least a tenth of them.  The k <= 256 label pass keeps their 512 per-colour labels in the LDS its
// pair table leaves free: on a photograph a few dozen dark cells hold more than half of the pixels AND most of the centroids,
// so their pair entries resolve almost nothing (28 % of the pixels of the test photograph against 84 % of a noise image).
constexpr uint32_t kHotMax = 56;
constexpr uint32_t kWorkWords = kCells + 1 + 1 + kHotMax;   // work list: [n][cells ...] then [n_hot][hot cells ...]

// sub-cell table entry (u16): a label, or one of
constexpr uint16_t kSubEmpty = 0xFFFF;    // no pixel of the image has a colour in this sub-cell
constexpr uint16_t kSubMixed = 0xFFFE;    // occupied colours of the sub-cell carry different labels

// Range of the per-pixel quantities of kmg_math.h over the 512 colours of a cell (exact float
// minima / maxima), image independent.
struct alignas(16) CellBounds {
    float L0, L1, a0, a1, b0, b1, C0, C1, wC0, wC1, wH0, wH1;
    float pad[4];
};

// colour index = [r7..r3 g7..g3 b7..b3][r2 g2 b2][r1 r0 g1 g0 b1 b0]: cell-major, sub-cell-major
__host__ __device__ inline uint32_t colour_index(uint32_t px)
{
    // = ((r >> 3) << 19) | ((g >> 3) << 14) | ((b >> 3) << 9) | (r2 << 8) | (g2 << 7) | (b2 << 6) | ((r & 3) << 4) | ((g & 3) << 2) | (b & 3).
    // The bits of the three bytes that go to ADJACENT places are gathered by one 24-bit multiply each (the partial products land on
    // distinct bits, so nothing carries): 14 vector instructions instead of 19 in a pass that pays ~0.5 us per instruction and
    // pixel (profiles/NOTES.md round 5).  colour_index_reference below is the plain form; tests compare the two over all 2^24.
    const uint32_t cell = ((px << 16) & 0xF80000u) | ((px << 3) & 0x7C000u) | ((px >> 10) & 0x3E00u);
    const uint32_t low6 = (((px & 0x030303u) * 0x100401u) >> 16) & 63u;          // [r1 r0 g1 g0 b1 b0]
    const uint32_t hi3 = (((px & 0x040404u) * 0x040201u) >> 12) & 0x1C0u;        // [r2 g2 b2] << 6
    return cell | hi3 | low6;
}

__host__ __device__ inline uint32_t colour_index_reference(uint32_t px)
{
    const uint32_t r = px & 255u, g = (px >> 8) & 255u, b = (px >> 16) & 255u;
    return ((r >> 3) << 19) | ((g >> 3) << 14) | ((b >> 3) << 9) |
           (((r >> 2) & 1u) << 8) | (((g >> 2) & 1u) << 7) | (((b >> 2) & 1u) << 6) |
           ((r & 3u) << 4) | ((g & 3u) << 2) | (b & 3u);
}

__host__ __device__ inline void index_to_rgb(uint32_t idx, uint32_t &r, uint32_t &g, uint32_t &b)
{
    r = (((idx >> 19) & 31u) << 3) | (((idx >> 8) & 1u) << 2) | ((idx >> 4) & 3u);
    g = (((idx >> 14) & 31u) << 3) | (((idx >> 7) & 1u) << 2) | ((idx >> 2) & 3u);
    b = (((idx >> 9) & 31u) << 3) | (((idx >> 6) & 1u) << 2) | (idx & 3u);
}

inline uint32_t mask_words(uint32_t k) { return (k + 63u) / 64u; }
// bytes of the cube pass's `masks` buffer: the cell candidate masks (kCells x mask_words(k) u64) and, for k <= 256, behind
// them the per-SUB-CELL candidate masks (kCells x 8 x 4 u64) of the cells with more candidates than the sub-cell stage lists
inline size_t cube_masks_bytes(uint32_t k)
{
    return sizeof(uint64_t) * ((size_t)kCells * mask_words(k) + (k <= 256 ? (size_t)kCells * 8u * 4u : 0u));
}

// ---- pair entries (k <= 256): one u32 per 8x8x8 cell, all 32768 of them live in LDS during the
// label pass.  [label A:8][label B:8][direction:7][tlo:6][w:3]
//   p = n . (r & 7, g & 7, b & 7) - min over the cell of the same (so 0 <= p <= 42), n = one of the
//   125 integer directions with components in -2..2;
//   p < tlo      -> label A
//   p >= tlo + w -> label B           (w = 7: no B side)
//   otherwise    -> the per-colour table decides (kPairFine)
// The cube pass picks n so that the plane separates the cell's most frequent label from the rest
// and computes tlo / w exactly from the labels of the occupied colours, so the entry is always
// right and only the colours in a thin slab around the boundary need the per-colour gather.
constexpr uint32_t kPairDirs = 125;
constexpr uint32_t kPairFine = 0xFFFFFFFFu;

__host__ __device__ inline uint32_t pair_dir_code(int nx, int ny, int nz)
{
    return (uint32_t)((nx + 2) * 25 + (ny + 2) * 5 + (nz + 2));
}

// bytes (nx, ny, nz, -min p) as i8: one v_dot4_i32_i8 against (r & 7, g & 7, b & 7, 1) gives p
__host__ __device__ inline uint32_t pair_dir_word(uint32_t code)
{
    const int nx = (int)(code / 25u) - 2, ny = (int)((code / 5u) % 5u) - 2, nz = (int)(code % 5u) - 2;
    const int bias = 7 * ((nx < 0 ? -nx : 0) + (ny < 0 ? -ny : 0) + (nz < 0 ? -nz : 0));
    return ((uint32_t)nx & 255u) | (((uint32_t)ny & 255u) << 8) | (((uint32_t)nz & 255u) << 16) | ((uint32_t)bias << 24);
}

__host__ __device__ inline uint32_t pair_entry(uint32_t A, uint32_t B, uint32_t code, uint32_t tlo, uint32_t w)
{
    return (A & 255u) | ((B & 255u) << 8) | (code << 16) | (tlo << 23) | (w << 29);
}

// p of a pixel (portable form of the dot4 used by the label kernel)
__host__ __device__ inline int pair_project(uint32_t dir_word, uint32_t px)
{
    return (int)(int8_t)(dir_word & 255u) * (int)(px & 7u) + (int)(int8_t)((dir_word >> 8) & 255u) * (int)((px >> 8) & 7u) +
           (int)(int8_t)((dir_word >> 16) & 255u) * (int)((px >> 16) & 7u) + (int)(dir_word >> 24);
}

// label of a pixel from its cell's entry and p, or kPairFine
__host__ __device__ inline uint32_t pair_decode(uint32_t e, int p)
{
    const int tlo = (int)((e >> 23) & 63u);
    const int w = (int)(e >> 29);
    const int thr = tlo + w + (w == 7 ? 64 : 0);
    if (p < tlo) return e & 255u;
    if (p >= thr) return (e >> 8) & 255u;
    return kPairFine;
}

// once per processor: bounds[kCells], sub_bounds[kSubCells] (same quantities over a 4x4x4 sub-cell, index
// cell * 8 + [r2 g2 b2] = colour index >> 6) and lab_table[2^24] = (L, a, b, C) of every colour (256 MiB,
// image independent) so that the per-iteration cube pass loads Lab instead of recomputing it
hipError_t launch_cell_bounds(const float *lut, CellBounds *bounds, CellBounds *sub_bounds, float4 *lab_table, hipStream_t st);
// once per image: hist[2^24] must be zero on entry
hipError_t launch_histogram(const uint32_t *rgba, uint64_t n, uint32_t *hist, hipStream_t st);
// the same histogram for large images without a global atomic per pixel (partition by the top 10 colour
// bits, then one LDS histogram per partition chunk); with keys != NULL also tie[2^24] of the init below.
// small: 4 * 1024 + 1 u32 of scratch, elems: n u16, keys: n u32 or NULL.  hist / tie are cleared here.
hipError_t launch_partitioned_histogram(const uint32_t *rgba, uint64_t n, uint64_t first_index, uint32_t *small,
                                        uint16_t *elems, uint32_t *keys, uint32_t *hist, uint32_t *tie, hipStream_t st);
// once per image: agg[kCells][4] = (sum qL, sum qa, sum qb, count) of the image's pixels per cell,
// sub_agg[kSubCells][4] the same per 4x4x4 sub-cell, occ_bits[2^24 / 8]: bit (colour & 7) of byte (colour >> 3) = the
// image has pixels of this colour
hipError_t launch_cell_aggregates(const uint32_t *hist, const float4 *lab_table, int64_t *agg, int64_t *sub_agg, uint8_t *occ_bits,
                                  hipStream_t st);
// once per image: work[0] = number of occupied cells, work[1..] = their indices in ascending order;
// work[kCells + 1] = number of hot cells (n_pixels = 0: none wanted), work[kCells + 2 ..] = their indices (kWorkWords in all)
hipError_t launch_work_list(const int64_t *agg, uint32_t *work, uint64_t n_pixels, hipStream_t st);
// before anything is built for an image (the cost model's look at it): 16384 pixels at equal strides, one workgroup -- out3[0] = cells of
// the 32^3 grid the sample occupies, out3[1] = samples in crowded cells (>= 1 / 560 of the samples each), out3[2] = samples taken
hipError_t launch_sparsity_probe(const uint32_t *rgba, uint64_t n, uint32_t *out3, hipStream_t st);
// a share of the work list: out[0] = its number of cells, out[1..] = the occupied cells with index in
// [kCells part / parts, kCells (part + 1) / parts) (cell-sharded cube pass: every rank of a sharded image labels one share of the cube)
hipError_t launch_work_share(const uint32_t *work, uint32_t part, uint32_t parts, uint32_t *out, hipStream_t st);
// farthest-point initialisation over the colours of a large image: tie[2^24] (zero on entry) = 1 + the
// largest low half of the init key among the pixels of each colour (first_index = image-wide index of
// rgba[0], first_index + n <= 0xFFFFFFF0).
hipError_t launch_tie_keys(const uint32_t *rgba, uint64_t n, uint64_t first_index, uint32_t *tie, hipStream_t st);
size_t init_scratch_bytes();    // scratch of the passes: one record per occupied cell + the workgroups' slots
// once per initialisation, after the image is bound: the records of its occupied cells (work), keys zero
hipError_t launch_init_records(const uint32_t *work, const CellBounds *bounds, void *init_scratch, hipStream_t st);
// Launch j = 1 .. k of an initialisation (kmg_table.hip, k_init_fused): centroid j - 1 is picked from what
// launch j - 1 left behind (j >= 2; j = 1: cent[0] is there already), then -- do_pass != 0, j < k -- the
// running min-distance per colour is lowered against it for the cells it can reach (occ_bits: one bit per colour
// of the bound image).  Launch k (do_pass = 0) only picks centroid k - 1.  pick_rgba / lut: pixel 0 and the
// sRGB table (Candidate(0, 0.0) when every distance is zero).
// band_key != NULL (band of a sharded image): pass j against cent[j - 1] as it stands, then *band_key = the
// band's largest key -- the pick happens between the launches, by the caller's all-reduce.
hipError_t launch_init_pass_cells(const uint32_t *tie, const uint8_t *occ_bits, const float4 *lab_table, Centroid *cent,
                                  uint32_t j, int do_pass, float *dist, void *init_scratch, unsigned long long *band_key,
                                  const uint32_t *pick_rgba, const float *lut, hipStream_t st);
// The same for a whole image on one device with several picks per launch (k_init_cells_multi): launch 1 sweeps against cent[0];
// every later launch picks one to four centroids from the largest cell records the previous one left -- exactly the centroids the
// single picks would choose -- and sweeps against them; a launch that finds the table complete does nothing.  The number of
// centroids chosen after `launch` is *init_cells_multi_count(init_scratch, launch) (device memory; the host reads it between chunks).
hipError_t launch_init_cells_multi(const uint32_t *tie, const uint8_t *occ_bits, const float4 *lab_table, Centroid *cent, uint32_t k,
                                   uint32_t launch, float *dist, void *init_scratch, const uint32_t *pick_rgba, const float *lut,
                                   hipStream_t st);
const uint32_t *init_cells_multi_count(const void *init_scratch, uint32_t launch);
// per iteration (kmg_cube.hip): candidates + sub-cell stage, dominance tests, colour scan, pair entries -- ONE launch for k <= 256
// on images without hot cells (k_cube_small, k_cube_one), three otherwise (k_cube_stage, k_cube_scan, k_cube_pairs).
// work: [0] = number of occupied cells of the bound image, [1..] = their indices (built at bind time).
// masks: the cell candidate masks, mask_words(k) u64 per cell; cell_work: cube_work_bytes() of scratch (the
// stage kernel's records for the scan kernel and, behind them, its list of cells with long candidate lists).
// Every workgroup adds the sums of the clusters it met into row (workgroup % n_rows) of `sums` (k x 4 int64
// per row, zero on entry; n_rows = 1: the final sums).  hist == NULL: output pass of replace mode -- every
// colour of every cell is labelled, nothing is accumulated (agg, sub_agg, occ_bits, work, sums unused).
// flags bit 0: also write the per-colour labels of single-candidate cells (the label passes never read them).
// stats (optional, 8 x u64, zero on entry): single-candidate cells, other cells, sub-cells decided by their
// bounds, sub-cells scanned, candidates summed over the scanned sub-cells, cells beyond the listing limit, candidates the
// dominance phase removed, scanned sub-cells it left with one candidate.
size_t cube_work_bytes();
// flags bit 16: the pass leaves the cells' pair entries / summaries (what the LABEL pass reads first) to a later
// launch_cube_entries -- a loop that only needs the sums (kmg_lloyd_run) pays for them once, after its last iteration
constexpr uint32_t kCubeNoEntries = 0x10000u;
// flags bit 18 (callers): the bound image has hot cells (a photograph).  Its pass is the few hundred cells with long candidate lists --
// up to 215 candidates at k = 256, crowded into the dark corner of the cube -- which the three-launch pass spreads over the whole
// device (kCubeSplitLong: four waves per such cell); the one-launch pass would leave them to the few workgroups that own them.
constexpr uint32_t kCubeHot = 0x40000u;
constexpr uint32_t kCubeSplitLong = 0x80000u;  // (set by launch_cube itself, k <= 256, three launches: long-list cells are scanned by four waves each)
constexpr uint32_t kCubeSmallMaxK = 32;      // k up to which the cube pass is the one-launch k_cube_small
// the cube pass of this k with these caller flags (kCubeHot) is ONE launch -- k_cube_small (k <= 32), or k_cube_one for 32 < k <= 256
// on images without hot cells: when a label pass follows, the pass's tail (CubeTail) rides on that one
bool cube_single_launch(uint32_t k, uint32_t flags);
hipError_t launch_cube_entries(const uint32_t *work, const uint8_t *occ_bits, const void *colour_labels, uint16_t *sub_table,
                               uint32_t k, hipStream_t st);
// What the LAST launch of the cube pass does on top (one of its workgroups, once the sums are complete): hand the k x 4
// sums over from the pass's own accumulation buffer `sums` (which it leaves zeroed for the next pass -- no memset launch),
// and -- do_update -- the centroid update of choose_centroid.wgsl:180-206 from them (no k_update launch).
struct CubeTail {
    int64_t *acc_out = nullptr;      // receives the sums (NULL: `sums` is the caller's buffer, nothing is handed over)
    int do_update = 0;
    float convergence = 0.0f;
    Centroid *cent = nullptr;        // updated in place
    uint32_t *n_converged = nullptr;
};
// Load balance of the one-launch pass (k_cube_one) across the passes of a loop over ONE work list: `pass` counts the launches since
// the list was first walked (0: the tasks are dealt out afresh); state: cube_balance_bytes() of device memory that nothing else
// touches between the passes (two sets of 4 096 task ids + what each task's items cost).  Every pass each workgroup re-deals its
// tasks with one partner by the previous pass's weights (kmg_cube.hip).  NULL: the fixed deal, the state is left alone.
struct CubeBalance {
    uint16_t *state = nullptr;
    uint32_t pass = 0;
};
size_t cube_balance_bytes();
hipError_t launch_cube(const uint32_t *hist, const int64_t *agg, const int64_t *sub_agg, const uint8_t *occ_bits,
                       const uint32_t *work, const CellBounds *bounds, const CellBounds *sub_bounds, const Centroid *cent,
                       uint32_t k, const float4 *lab_table, uint64_t *masks, void *cell_work, void *colour_labels,
                       uint16_t *sub_table, int64_t *sums, uint32_t n_rows, uint32_t flags, unsigned long long *stats,
                       hipStream_t st, const CubeTail *tail = nullptr, const float *sub_affine = nullptr,
                       const CubeBalance *balance = nullptr);
// sub_affine (optional, once per processor, sub_affine_bytes() = 24 MiB, image independent; k_cube_small and k_cube_one use it):
// per sub-cell affine models (binary16) of the seven per-colour features the difference of two keys is linear in, with exact
// residual ranges -- the dominance test that removes, from a sub-cell's candidates, those another candidate beats on every
// colour of the sub-cell (3x fewer scanned sub-cells at k = 16, 2.3x fewer at k = 256).
size_t sub_affine_bytes();
hipError_t launch_sub_affine(const float4 *lab_table, const CellBounds *sub_bounds, float *affine, hipStream_t st);
// pal == NULL: labels[i] = label; pal != NULL: labels[i] = pal[label] (RGBA8 output of replace mode).
// reserve_cus: compute units left without a workgroup of the k <= 256 label pass (kmg_lloyd_reserve_cus)
// hot: NULL, or the image's hot cells ([n_hot][cells ...], n_hot > 0 known to the host): k <= 256 keeps their labels in LDS
// tail + tail_sums (k <= 256): the cube pass's CubeTail performed by this launch's last workgroup instead of the cube pass's own
// (the pass of small centroid tables, k <= kCubeSmallMaxK, is one launch and would need another for it)
hipError_t launch_labels(const uint32_t *rgba, uint64_t n, const void *colour_labels,
                         const uint16_t *sub_table, uint32_t k, const uint32_t *pal, uint32_t *labels,
                         hipStream_t st, uint32_t reserve_cus = 0, const uint32_t *hot = nullptr,
                         const CubeTail *tail = nullptr, int64_t *tail_sums = nullptr);

// ordered-dither output pass with candidate pruning: masks[(cell * 16 + Bayer index) * words + w] are the
// centroids that can be the arg-min of Lab(colour) + threshold * (M[Bayer index] / 16 - 0.5) for any
// colour of the cell; the pass scans only those (pal: k + 1 RGBA8 words, entry k = the sentinel)
hipError_t launch_offset_candidates(const CellBounds *bounds, const Centroid *cent, uint32_t k, float threshold,
                                    uint64_t *masks, hipStream_t st);
hipError_t launch_dither_pruned(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0, const Centroid *cent,
                                uint32_t k, const float *lut, const uint32_t *pal, float threshold,
                                const uint64_t *masks, uint32_t *out, hipStream_t st);

// The same pass for k <= 512 (meld: 256) over byte lists per cell of a 4 x 4 x 4 grid over Lab (kmg_lists.hip): lists = 2 x kLabCells
// records of kListBytes, record c = [count][the first 31 candidate indices, ascending], record kLabCells + c = indices 31 .. 62;
// count 255: scan all centroids.  No image-independent table is needed (no CellBounds): the cells are boxes of Lab.
constexpr uint32_t kListBytes = 32, kListMax = 2 * kListBytes - 1;
constexpr uint32_t kLabCells = 40u * 72u * 72u;
constexpr size_t kLabListBytes = 2ull * kLabCells * kListBytes;
// the dither pass also takes 256 < k <= kLabListMaxK: two byte lists per cell (centroids 0 .. 255 / 256 .. k - 1), twice the table
constexpr uint32_t kLabListMaxK = 512;
size_t lab_list_bytes(uint32_t k);
// two_closest: the lists of the meld pass (whatever can be one of a pixel's TWO closest centroids; threshold 0)
hipError_t launch_lab_candidates(const Centroid *cent, uint32_t k, float threshold, bool two_closest, uint8_t *lists, hipStream_t st);
hipError_t launch_meld_lists(const uint32_t *rgba, uint64_t n, const Centroid *cent, uint32_t k, const float *lut,
                             const uint8_t *lists, uint32_t *out, hipStream_t st);
hipError_t launch_check_lab_lists_two(const Centroid *cent, uint32_t k, const uint8_t *lists, const float *lut,
                                      unsigned long long *violations, hipStream_t st);
hipError_t launch_dither_lists(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0, const Centroid *cent, uint32_t k,
                               const float *lut, const uint32_t *pal, float threshold, const uint8_t *lists, uint32_t *out,
                               hipStream_t st);
// test support: number of (colour, Bayer index) pairs (all 2^24 x 16) whose literal arg-min is not the arg-min over the list
hipError_t launch_check_lab_lists(const Centroid *cent, uint32_t k, const uint8_t *lists, const float *lut, float threshold,
                                  unsigned long long *violations, hipStream_t st);
// test support: number of (colour, Bayer index) pairs whose arg-min over the candidates differs from
// the brute-force dither arg-min
hipError_t launch_check_offset_masks(const Centroid *cent, uint32_t k, const uint64_t *masks, const float *lut,
                                     float threshold, unsigned long long *violations, hipStream_t st);

// meld output pass with candidate pruning: masks[cell * words + w] = the centroids that can be one of the two
// closest (literal CIE94) of any colour of the cell, k >= 2; launch_meld (kmg_kernels.h) scans only those
hipError_t launch_meld_candidates(const CellBounds *bounds, const Centroid *cent, uint32_t k, uint64_t *masks, hipStream_t st);
// test support: number of colours whose two closest centroids differ between the full and the pruned scan
hipError_t launch_check_meld_masks(const Centroid *cent, uint32_t k, const uint64_t *masks, const float *lut,
                                   unsigned long long *violations, hipStream_t st);

// debug / test support (see k_check_bounds): bound violations, arg-mins missing from the cell masks, per-colour
// labels that are not the arg-min, over all 2^24 colours
hipError_t launch_check_bounds(const CellBounds *bounds, const CellBounds *sub_bounds, const Centroid *cent, uint32_t k,
                               const uint64_t *masks, const void *colour_labels, const float *lut,
                               unsigned long long *violations, hipStream_t st);

}  // namespace kmg
