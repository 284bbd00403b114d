/*
 * kmeans_hip.h -- C ABI of libkmeans_hip.so: the MI355X (gfx950) implementation of the
 * Lloyd-iteration hot path of redwarp/kmeans-gpu (per-pixel CIE94 assignment, per-cluster
 * sum/count reduction + centroid update, ordered-dither / replace output pass).
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  The high-level calls take HOST buffers exactly like the reference
 * crate's `ImageProcessor` (core/src/lib.rs:24-165); the `kmg_dev_*` / `kmg_lloyd_*` calls take
 * DEVICE pointers and an explicit hipStream_t so a host runtime (one process per GPU) can
 * shard an image, run the exchange step of the update itself (RCCL all-reduce of the k
 * accumulators) and keep everything resident in HBM.
 *
 * Conventions
 *  - images: tightly packed row-major RGBA8, 4 bytes per pixel, no row padding
 *    (core/src/image.rs:20-48); alpha is ignored on input and 255 on output.
 *  - centroid tables: k x 4 floats (L, a, b, 1.0) -- the `vec4<f32>` array of the reference's
 *    CentroidsBuffer (core/src/structures.rs:501-521) without its 16-byte count header.
 *  - accumulators: k x 4 int64 = (sum qL, sum qa, sum qb, count), q = rint(Lab * 2^20).
 *    Integer sums are order independent, so results are identical for any tiling or GPU count.
 *  - all functions return KMG_OK (0) or a negative kmg_status; kmg_last_error() gives the
 *    thread-local message of the last failure.
 *  - every entry point is re-entrant on one kmg_processor (per-call workspace + stream), like
 *    the reference's Send+Sync ImageProcessor (core/examples/parallel.rs:36-50).
 *  - there is NO CPU fallback: without a usable HIP device kmg_processor_create fails.
 *  - no C++ exception leaves the library (every entry point is a function-try-block): host allocation failures come back
 *    as KMG_ERR_OUT_OF_MEMORY, any other internal exception as KMG_ERR_HIP -- the reference's anyhow::Result (lib.rs:38).
 */
#ifndef KMEANS_HIP_H
#define KMEANS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMG_API __attribute__((visibility("default")))

typedef enum kmg_status {
    KMG_OK = 0,
    KMG_ERR_INVALID_ARGUMENT = -1, /* null pointer, zero-sized image, color_count == 0 ... */
    KMG_ERR_NO_DEVICE = -2,        /* no HIP device / device init failed                    */
    KMG_ERR_HIP = -3,              /* a HIP runtime call failed (message has the detail)    */
    KMG_ERR_OUT_OF_MEMORY = -4,
    KMG_ERR_UNSUPPORTED = -5       /* e.g. k > KMG_MAX_K, more than 2^32-1 pixels            */
} kmg_status;

/* core/src/lib.rs:215-219 `enum Algorithm` */
typedef enum kmg_algorithm { KMG_ALGO_KMEANS = 0, KMG_ALGO_OCTREE = 1 } kmg_algorithm;
/* core/src/lib.rs:234-239 `enum ReduceMode` */
typedef enum kmg_reduce_mode { KMG_MODE_REPLACE = 0, KMG_MODE_DITHER = 1, KMG_MODE_MELD = 2 } kmg_reduce_mode;

#define KMG_FIX_SHIFT 20
/* largest k: the kernels keep 48 bytes of LDS per cluster (centroid + int64 sums), 160 KiB per CU */
#define KMG_MAX_K 3072u

/* Compile-time constants of the reference exposed as options (defaults = reference values). */
typedef struct kmg_options {
    uint32_t struct_size;     /* sizeof(kmg_options)                                           */
    int32_t  device;          /* HIP device ordinal, -1 = current device                       */
    uint32_t shrink_max_dim;  /* MAX_IMAGE_DIMENSION = 256 (core/src/structures.rs:23); 0 = off */
    uint32_t max_iterations;  /* MAX_ITERATION = 128 (core/src/modules.rs:765)                  */
    uint32_t check_period;    /* MAX_ITERATION_BEFORE_CONVERGENCE_CHECK = 8 (modules.rs:766)    */
    float    convergence;     /* ColorSpace::Lab.convergence() = 1.0 (core/src/lib.rs:189-194)  */
    int32_t  strategy;        /* KMG_STRATEGY_*: 0 = the library's cost models pick per call (default)                     */
} kmg_options;

/* kmg_options.strategy / kmg_processor_set_strategy: which of the library's interchangeable strategies a call takes.  Results are
 * IDENTICAL either way (that is what the tests use the switch for); only the time differs.  The low two bits choose between the
 * per-pixel scans and the colour-table / candidate-list passes for every decision the cost models otherwise make (Lloyd
 * iteration, initialisation, output passes); KMG_STRATEGY_MASK_WORDS sends the pruned dither / meld passes of every k through
 * the mask words per (RGB cell, Bayer index) instead of the byte lists over Lab cells (the path of k > 512).  (An options
 * struct of the previous size -- without this field -- is accepted and means KMG_STRATEGY_AUTO.)                                */
#define KMG_STRATEGY_AUTO       0
#define KMG_STRATEGY_SCAN       1
#define KMG_STRATEGY_TABLE      2
#define KMG_STRATEGY_MASK_WORDS 4

typedef struct kmg_processor kmg_processor;
typedef struct kmg_lloyd kmg_lloyd;
typedef struct kmg_apply_plan kmg_apply_plan;

KMG_API const char *kmg_last_error(void);
KMG_API const char *kmg_version(void);
KMG_API void kmg_default_options(kmg_options *opt);

/* ---- ImageProcessor::new  (core/src/lib.rs:38-65) ------------------------------------- */
KMG_API int kmg_processor_create(kmg_processor **out);
KMG_API int kmg_processor_create_ex(const kmg_options *opt, kmg_processor **out);
KMG_API void kmg_processor_destroy(kmg_processor *p);
/* changes kmg_options.strategy of a live processor (tests and tuning: one processor, both strategies); calls that are
 * already running keep the strategy they started with                                                                        */
KMG_API int kmg_processor_set_strategy(kmg_processor *p, int strategy);
/* Page-locked host memory for images that cross the boundary often (a frame loop): a result buffer from kmg_host_alloc has
 * its pages resident and is copied to by DMA directly -- kmg_reduce of 8192 x 8192 into a fresh pageable buffer spends 30-50 ms
 * in the caller's page faults, 10 ms into one of these.  Plain memory otherwise; release with kmg_host_free.  (No counterpart
 * in the reference, whose results are fresh Vecs: structures.rs:441-470.)                                                    */
KMG_API int kmg_host_alloc(size_t bytes, void **out);
KMG_API void kmg_host_free(void *ptr);
/* Test support: out[0] = device blocks the processor has allocated with hipMalloc so far, out[1] = blocks it has handed out
 * again (colour tables, workspaces and output-pass scratch of finished objects are kept and reused).                 */
KMG_API int kmg_debug_block_counts(kmg_processor *p, uint64_t out[2]);
/* Test support: out[0] = blocks the processor holds idle right now, out[1] = their bytes.  The idle list is bounded (24 blocks,
 * 3 GiB: the oldest go first) and is emptied when an allocation fails for lack of memory.                                  */
KMG_API int kmg_debug_idle_blocks(kmg_processor *p, uint64_t out[2]);
/* Test support: the meld pass turns a linear channel value into its sRGB8 byte with a 255-entry threshold table made on
 * the device by the encode of lab_to_rgb.wgsl:21-35 itself; *mismatches = the float values (every bit pattern, NaN aside)
 * for which table and encode give different bytes (0 = the table IS the encode).                                     */
KMG_API int kmg_debug_encode_table_check(kmg_processor *p, uint64_t *mismatches);
/* Test support: the device divides by the constants of lab_to_rgb.wgsl:45-59 (116, 500, 200, 100, 7.787) and of
 * rgb_to_lab.wgsl (the white point) with a reciprocal and one residual correction; out[0] = the number of binary32 x (every
 * bit pattern, NaN aside) for which that is not the IEEE quotient x / c, out[1] / out[2] = the smallest / largest |x| bit
 * pattern among them (out[1] = 2^64 - 1 when there is none).                                                          */
KMG_API int kmg_debug_division_check(kmg_processor *p, float c, uint64_t out[3]);

/* ---- ImageProcessor::palette  (core/src/lib.rs:67-77, 255-286) -------------------------
 * out_rgba: capacity color_count*4 bytes; *out_count receives the number of colours
 * (== color_count for k-means), sorted ascending by Lab L.                                 */
KMG_API int kmg_palette(kmg_processor *p, const uint8_t *rgba, uint32_t width, uint32_t height,
                        uint32_t color_count, int algo, uint8_t *out_rgba, uint32_t *out_count);

/* ---- ImageProcessor::find  (core/src/lib.rs:79-114) ------------------------------------ */
KMG_API int kmg_find(kmg_processor *p, const uint8_t *rgba, uint32_t width, uint32_t height,
                     const uint8_t *palette_rgba, uint32_t n_colors, int mode, uint8_t *out_rgba);

/* ---- ImageProcessor::reduce  (core/src/lib.rs:116-164) --------------------------------- */
KMG_API int kmg_reduce(kmg_processor *p, const uint8_t *rgba, uint32_t width, uint32_t height,
                       uint32_t color_count, int algo, int mode, uint8_t *out_rgba);

/* ---- host-side colour helpers the reference takes from the `palette` crate -------------
 * CentroidsBuffer::fixed_centroids (core/src/structures.rs:523-553): sRGB8 -> Lab (L,a,b,1)  */
KMG_API int kmg_palette_to_centroids(const uint8_t *palette_rgba, uint32_t n_colors, float *centroids4);
/* CentroidsBuffer::pull_values (core/src/structures.rs:581-617): Lab -> sRGB8 (alpha 255)    */
KMG_API int kmg_centroids_to_palette(const float *centroids4, uint32_t k, uint8_t *out_rgba);
/* ColorTree::{add_color, reduce} (core/src/octree.rs:28-115, core/src/operations.rs:90-97): the
 * reference's CPU octree quantiser on n_pixels RGBA8 pixels.  out_rgba: capacity 4 * min(color_count,
 * n_pixels) bytes; *out_count <= color_count colours, sorted (r, g, b, a), deduplicated.  Host only.  */
KMG_API int kmg_octree_palette(const uint8_t *rgba, uint64_t n_pixels, uint32_t color_count,
                               uint8_t *out_rgba, uint32_t *out_count);

/* ======================= device-pointer API (hot path building blocks) ================== */
/* `stream` is a hipStream_t (NULL = the default stream).  Calls only enqueue work unless the
 * comment says they synchronise.                                                            */

/* ColorConverterModule, rgb_to_lab.wgsl:66-80: RGBA8 -> Lab, 3 floats per pixel.            */
KMG_API int kmg_dev_rgb_to_lab(kmg_processor *p, const uint8_t *d_rgba, uint64_t n_pixels,
                               float *d_lab3, void *stream);

/* InputTexture::resized (core/src/structures.rs:76-182, resize.wgsl:7-18).                   */
KMG_API void kmg_resized_dims(uint32_t width, uint32_t height, uint32_t max_size,
                              uint32_t *new_width, uint32_t *new_height);
KMG_API int kmg_dev_resize(kmg_processor *p, const uint8_t *d_rgba, uint32_t width, uint32_t height,
                           uint32_t new_width, uint32_t new_height, uint8_t *d_out_rgba, void *stream);

/* One Lloyd problem = one image (or one row band of it) with k centroids.
 * ChooseCentroidModule + FindCentroidModule state (core/src/modules.rs:452-761).             */
KMG_API int kmg_lloyd_create(kmg_processor *p, uint32_t k, kmg_lloyd **out);
KMG_API void kmg_lloyd_destroy(kmg_lloyd *s);
/* centroid table up/down (synchronise `stream`) */
KMG_API int kmg_lloyd_set_centroids(kmg_lloyd *s, const float *centroids4, void *stream);
KMG_API int kmg_lloyd_get_centroids(kmg_lloyd *s, float *centroids4, void *stream);

/* PlusPlusInitModule::compute (core/src/modules.rs:946-1246, plus_plus_init.wgsl,
 * kmeans++_calc_diff.wgsl): deterministic farthest-point initialisation on the device.       */
KMG_API int kmg_lloyd_init_centroids(kmg_lloyd *s, const uint8_t *d_rgba, uint32_t width,
                                     uint32_t height, void *stream);

/* The same initialisation for an image sharded in row bands (one kmg_lloyd per band / GPU).  Step j:
 *   kmg_lloyd_init_step      band-local pass for centroid j-1; *d_key (device u64) = arg-max key of
 *                            this band over IMAGE-wide pixel indices (first_index = index of the band's
 *                            first pixel).  The caller all-reduces the key with MAX (compare as signed
 *                            or unsigned 64-bit: the top bit is never set).
 *   kmg_lloyd_init_pick_band d_colour2[0..1] = {RGBA8 of the pixel the reduced key names, 1} on the band
 *                            that owns that pixel, {0, 0} elsewhere.  The caller all-reduces with SUM.
 *   kmg_lloyd_set_centroid_rgba   centroid j <- shader Lab of d_colour[0].
 * Centroid 0 uses the same two calls with kmg_init_first_key(width, height) as the key.  Everything
 * is enqueued on `stream`; nothing synchronises.                                                     */
KMG_API int kmg_lloyd_init_step(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_local, uint64_t first_index,
                                uint32_t j, uint64_t *d_key, void *stream);
KMG_API int kmg_lloyd_init_pick_band(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_local, uint64_t first_index,
                                     const uint64_t *d_key, uint32_t *d_colour2, void *stream);
KMG_API int kmg_lloyd_set_centroid_rgba(kmg_lloyd *s, uint32_t j, const uint32_t *d_colour, void *stream);
KMG_API uint64_t kmg_init_first_key(uint32_t width, uint32_t height);

/* FindCentroidModule::dispatch (find_centroid.wgsl:15-44) fused with the masked sums of
 * choose_centroid.wgsl:75-178: labels for every pixel AND the k x 4 int64 accumulators of this
 * pixel range, one pass over the RGBA8 data.  d_acc4 may be NULL (assignment only);
 * d_labels may be NULL (sums only).                                                          */
KMG_API int kmg_lloyd_assign_accumulate(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels,
                                        uint32_t *d_labels, int64_t *d_acc4, void *stream);

/* Optional, for large images: build the image's colour table (24-bit colour histogram + per-cell
 * sums) once.  Subsequent assign passes on the SAME (d_rgba, n_pixels) then iterate over distinct
 * colours with conservatively pruned candidate sets and materialise labels with one gather pass;
 * labels, sums and centroids are bit-identical to the per-pixel scan.  kmg_lloyd_run binds by
 * itself when its cost model says it pays (kmg_options.strategy overrides).  The caller
 * must not modify the pixel buffer while it is bound.  kmg_lloyd_init_centroids / _init_step (j = 1)
 * start a new problem: they drop any earlier binding of the buffer (and bind it afresh when the
 * initialisation itself runs over the colour table).                                             */
KMG_API int kmg_lloyd_bind_image(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, void *stream);
KMG_API int kmg_lloyd_unbind_image(kmg_lloyd *s);
/* Tuning support: what the binding found in the image -- out[0] = occupied cells of the 32^3 grid over the colour cube (the cube
 * pass's work list), out[1] = hot cells (the few cells that hold a tenth or more of the pixels: a photograph's dark corner; 0
 * on noise).  These are what the cost model looks at AFTER a binding (kmg_lloyd_prepare): a sparse image has a cheaper cube pass
 * than the noise the model was fitted on, an image with hot cells a dearer per-pixel scan (near-tie repairs).                  */
KMG_API int kmg_debug_bound_image(kmg_lloyd *s, uint64_t out[2]);
/* One-time preparation of (d_rgba, n_pixels) for repeated assign passes: applies the library's cost
 * model (want_labels = whether the passes will materialise labels) and binds the image if the colour
 * table pays.  *strategy (optional) receives 0 = per-pixel scan, 1 = colour table.              */
KMG_API int kmg_lloyd_prepare(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, int want_labels,
                              int *strategy, void *stream);
/* Test support: exhaustive check over all 2^24 colours of the colour-table pass for the current
 * centroids.  out[0] = (colour, centroid) pairs whose key lies outside the interval bounds of the colour's
 * cell or sub-cell, out[1] = colours whose true arg-min is missing from its cell's candidate set,
 * out[2] = colours whose label in the per-colour table is not the true arg-min (all must be 0). */
KMG_API int kmg_debug_check_table(kmg_lloyd *s, uint64_t out[3], void *stream);
/* Tuning support: statistics of the last colour-table pass over the bound image (synchronises).
 * out = {occupied cells, sum of candidate counts, cells with one candidate, max candidates,
 *        cells with one label, occupied sub-cells, sub-cells with one label, distinct colours,
 *        sub-cells decided from their bounds, sub-cells scanned, candidates over the scanned sub-cells,
 *        cells with too many candidates for the sub-cell stage, candidates the dominance phase removed from
 *        scanned sub-cells, scanned sub-cells it left with one candidate}.                      */
KMG_API int kmg_debug_table_stats(kmg_lloyd *s, uint64_t out[14], void *stream);
/* Test support (k <= 256): checks the per-cell pair entries the label pass keeps in LDS against the
 * per-colour label table of the last colour-table pass (synchronises).  out[0] = occupied colours
 * whose entry disagrees (must be 0), out[1] = pixels resolved by the entries alone, out[2] = pixels. */
KMG_API int kmg_debug_check_pairs(kmg_lloyd *s, uint64_t out[3], void *stream);
/* Test support: exhaustive check (2^24 colours x 16 Bayer offsets) that the candidate masks of the
 * pruned dither pass for this centroid table (k >= 2, (L, a, b, pad) per entry) contain every pixel's
 * true arg-min of mix_colors.wgsl:73-80.  *violations must come back 0.                            */
KMG_API int kmg_debug_check_dither_masks(kmg_processor *p, const float *centroids4, uint32_t k, uint64_t *violations,
                                         void *stream);
/* Test support: the same exhaustive check (2^24 colours) for the pruned meld pass: the two closest
 * centroids of mix_colors.wgsl:29-48 found among a cell's candidates must be those of the full scan.  */
KMG_API int kmg_debug_check_meld_masks(kmg_processor *p, const float *centroids4, uint32_t k, uint64_t *violations,
                                       void *stream);

/* Labels only, for the CURRENT centroid table: find_centroid.wgsl:15-44 without the sums.  With a
 * bound image whose label tables are current (an assign pass ran since the last centroid change) this
 * is just the label-gather pass.                                                                 */
KMG_API int kmg_lloyd_labels(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, uint32_t *d_labels, void *stream);
/* Cell-sharded cube pass, for ONE image sharded over `parts` ranks (strong scaling; no counterpart in the reference, which
 * is single-device: core/src/lib.rs:38-65).  Every rank binds the WHOLE image's colour histogram (its band's histogram,
 * all-reduced) and labels the colours of one share of the colour cube per iteration: after _set_cell_share(part, parts)
 * the assign passes of a bound image visit only the occupied cells whose index lies in [32768 part / parts,
 * 32768 (part + 1) / parts) -- equal RANGES of the colour cube (slabs of the red axis), so that the shares of the label tables
 * are equal contiguous chunks an in-place all-gather can move; the WORK per share is equal only when the occupied cells are
 * spread evenly (noise: yes; the test photograph: the fullest of 2 / 4 / 8 shares holds 1.20 / 1.25 / 1.31 x the mean number of
 * occupied cells -- counted, not timed; its crowded dark cells carry more candidates each).  The sums such a pass returns
 * are those of the share's colours (the all-reduce of the k x 4 accumulators makes them the image's), and only the share's
 * per-colour labels and cell entries are (re)written.  The ranks then exchange their shares of the label tables
 * (_table_buffers: the per-colour labels, cell-major, 512 per cell, and the cell entries) and write their band's label map
 * with _labels_from_tables, which applies the tables as they stand to ANY pixels whose colours occur in the bound image.
 * While a share is set, a pass that asks for a label map, a centroid update (kmg_lloyd_assign_update with do_update,
 * kmg_lloyd_run, kmg_lloyd_iterate) or the two-step partial sums is refused with KMG_ERR_INVALID_ARGUMENT -- it would be that
 * of a fraction of the image.  parts = 1 restores the whole list.                                                        */
KMG_API int kmg_lloyd_set_cell_share(kmg_lloyd *s, uint32_t part, uint32_t parts, void *stream);
/* The bound image's colour histogram (2^24 u32 counts in the library's cell-major colour order) for the all-reduce that
 * turns the band's histogram into the image's, and the call that re-derives everything a binding derives from it (cell
 * sums, occupied and hot cells; synchronises) -- n_pixels = the pixels the histogram now counts (< 2^32).            */
KMG_API int kmg_lloyd_histogram_buffer(kmg_lloyd *s, void **hist, uint64_t *bytes);
KMG_API int kmg_lloyd_rebuild_from_histogram(kmg_lloyd *s, uint64_t n_pixels, void *stream);
KMG_API int kmg_lloyd_labels_from_tables(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, uint32_t *d_labels,
                                         void *stream);
KMG_API int kmg_lloyd_table_buffers(kmg_lloyd *s, void **colour_labels, uint64_t *colour_label_bytes, void **entries,
                                    uint64_t *entry_bytes);
/* Two launches fewer per iteration of such a loop (k <= 256): the cube pass ADDS its sums to d_acc4 as it stands -- the caller keeps
 * the buffer zero between passes; no hand-over launch -- and the label pass of the band also performs kmg_lloyd_update from
 * d_acc4 (the all-reduced sums) and clears it for the next pass: per iteration  _accumulate_into -> all-reduce -> all-gather ->
 * _labels_from_tables_update  instead of  _assign_accumulate -> all-reduce -> all-gather -> _labels_from_tables -> _update.
 * One assignment with its label map and one update per iteration, shifted by half a step (as kmg_lloyd_assign_update).        */
KMG_API int kmg_lloyd_accumulate_into(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, int64_t *d_acc4, void *stream);
KMG_API int kmg_lloyd_labels_from_tables_update(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, uint32_t *d_labels,
                                                int64_t *d_acc4, void *stream);

/* The label pass of the colour-table strategy (k <= 256) runs one 1024-thread workgroup per compute unit for its whole
 * duration.  A kernel launched beside it on another stream -- the RCCL all-reduce of the sums that a sharded loop issues
 * asynchronously: 256 threads, 20 KiB LDS, 280 registers per lane -- finds no CU it fits on and would run
 * BEHIND the pass.  n_cus > 0 leaves that many CUs without a label workgroup (n_cus / 256 of the pass's throughput) so
 * that the collective runs beside it.  0 (default): all CUs.                                                          */
KMG_API int kmg_lloyd_reserve_cus(kmg_lloyd *s, uint32_t n_cus);

/* The two halves of kmg_lloyd_assign_accumulate, for callers that time or batch them:
 * _assign_partials runs the fused per-pixel kernel (labels + per-workgroup partial sums kept in
 * the state), _reduce_partials folds the partial sums of that same launch into d_acc4.        */
KMG_API int kmg_lloyd_assign_partials(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels,
                                      uint32_t *d_labels, void *stream);
KMG_API int kmg_lloyd_reduce_partials(kmg_lloyd *s, uint64_t n_pixels, int64_t *d_acc4, void *stream);

/* Per-launch timing of the state's kernels with HIP events recorded on the launch stream itself
 * (bench.py's roofline leg).  _profile(mask) starts collecting for the kernel ids whose bit is set
 * (-1 = all, 0 = stop); _profile_read synchronises the recorded events, returns the summed duration
 * and launch count per kernel id and resets.  Every timed launch adds two event records to the
 * stream, so time only what is needed inside a throughput measurement.
 * KMG_K_CUBE covers the launches of the cube pass (k_cube_stage, k_cube_scan, k_cube_pairs; k <= 32: k_cube_small) as one interval;
 * KMG_K_CANDIDATES is the candidate kernel of the first release, now the first phase of k_cube_stage: the id is
 * kept so that the others do not move, nothing is reported under it.                                             */
typedef enum kmg_kernel_id {
    KMG_K_ASSIGN = 0, KMG_K_REDUCE = 1, KMG_K_UPDATE = 2, KMG_K_CANDIDATES = 3, KMG_K_CUBE = 4,
    KMG_K_LABELS = 5, KMG_K_COUNT = 6
} kmg_kernel_id;
KMG_API const char *kmg_kernel_name(int id);
KMG_API int kmg_lloyd_profile(kmg_lloyd *s, int enable);
KMG_API int kmg_lloyd_profile_read(kmg_lloyd *s, double total_ms[KMG_K_COUNT], uint32_t launches[KMG_K_COUNT]);

/* choose_centroid.wgsl:180-206 `pick` for all k at once: centroid <- sum/count, convergence
 * flags.  d_acc4 holds the (all-reduced) accumulators.                                        */
KMG_API int kmg_lloyd_update(kmg_lloyd *s, const int64_t *d_acc4, void *stream);
/* The loop body of ChooseCentroidModule::compute (modules.rs:769-800) shifted by half a step: the assign pass of
 * kmg_lloyd_assign_accumulate (labels optional, sums into d_acc4) and then -- do_update != 0 -- kmg_lloyd_update from
 * those sums.  Same results as the two calls; with a bound image (colour table) the update is done by the last launch of
 * the assign pass itself (one launch and one memset fewer per iteration).  d_acc4 holds the sums on return, the label
 * tables keep describing the assignment just made.  Sharded images need the all-reduce between the two halves and use
 * the separate calls.                                                                           */
KMG_API int kmg_lloyd_assign_update(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, uint32_t *d_labels,
                                    int64_t *d_acc4, int do_update, void *stream);
/* convergence[K] of choose_centroid.wgsl:196-202 after the last update (synchronises).       */
KMG_API int kmg_lloyd_converged_count(kmg_lloyd *s, uint32_t *count, void *stream);

/* One iteration of the loop of ChooseCentroidModule::compute (modules.rs:769-800) as ONE asynchronous call:
 * update_first != 0: centroids <- kmg_lloyd_update(d_acc4); then labels (optional) + sums of the new
 * assignment into d_acc4 (cleared first).  With a bound image (colour table) and d_labels != NULL the label
 * pass runs on an internal high-priority stream beside the NEXT iteration's update + cube pass -- it feeds
 * nothing in the loop; `stream` is ordered after everything that produces d_acc4 and the centroids, but NOT
 * after the label map: call kmg_lloyd_flush (or synchronise the device) before reading d_labels.
 * Between two calls the caller may all-reduce d_acc4 on `stream` (sharded images).  Without a bound image it
 * is kmg_lloyd_update + kmg_lloyd_assign_accumulate.                                                  */
KMG_API int kmg_lloyd_iterate(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels, uint32_t *d_labels,
                              int64_t *d_acc4, int update_first, void *stream);
/* `stream` waits (on the device, no host synchronisation) for the label passes kmg_lloyd_iterate started. */
KMG_API int kmg_lloyd_flush(kmg_lloyd *s, void *stream);

/* ChooseCentroidModule::compute (core/src/modules.rs:763-840): the whole loop on one device.
 * Expects the centroid table initialised.  Synchronises.  *iterations = the reference's
 * `current_iteration` when the loop stopped.  d_labels (optional) receives the final assignment;
 * the loop applies kmg_lloyd_prepare's cost model itself and, with the colour table, iterates on the
 * sums only and writes the label map once after the last iteration (same result).               */
KMG_API int kmg_lloyd_run(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_pixels,
                          uint32_t *d_labels, uint32_t *iterations, void *stream);

/* find_colors / dither_colors (core/src/operations.rs:99-155,215-271; find_centroid.wgsl,
 * swap.wgsl, mix_colors.wgsl main_dither, lab_to_rgb.wgsl) on a row band: `row0` is the image
 * row of the band's first pixel (the Bayer index uses image coordinates).  centroids4 is a
 * HOST table.  Output RGBA8 in d_out_rgba.  mode = replace, dither or meld (mix_colors.wgsl main_meld).  */
KMG_API int kmg_dev_apply(kmg_processor *p, const uint8_t *d_rgba, uint32_t width, uint32_t rows,
                          uint32_t row0, const float *centroids4, uint32_t k, int mode,
                          uint8_t *d_out_rgba, void *stream);

/* The same output pass as a PLAN, for callers that process an image in row bands (a multi-GPU runtime, an image streamed
 * through a pinned double buffer): _create builds everything that depends on the centroid table only -- the device copies of
 * centroids and palette, the dither threshold, the candidate lists / masks or the label tables of the colour cube -- once,
 * asynchronously on `stream`; n_pixels_hint = the pixels the plan is made for (it selects the route exactly as kmg_dev_apply
 * does for a band of that size).  _run launches the per-pixel kernel of one band on any stream (it waits for the tables on the
 * device, never on the host) and returns at once; any number of bands, any order.  _destroy returns the plan's scratch block:
 * the caller has synchronised every stream that ran the plan, or passes synchronise != 0.  kmg_dev_apply = create + run +
 * stream synchronisation + destroy.  (The reference runs the pass on whole textures only: operations.rs:99-155.)              */
KMG_API int kmg_apply_plan_create(kmg_processor *p, const float *centroids4, uint32_t k, int mode, uint64_t n_pixels_hint,
                                  void *stream, kmg_apply_plan **out);
KMG_API int kmg_apply_plan_run(kmg_apply_plan *plan, const uint8_t *d_rgba, uint32_t width, uint32_t rows, uint32_t row0,
                               uint8_t *d_out_rgba, void *stream);
KMG_API void kmg_apply_plan_destroy(kmg_apply_plan *plan, int synchronise);

/* mix_colors.wgsl:53-67: the dither threshold of a centroid table (host helper).             */
KMG_API int kmg_dither_threshold(const float *centroids4, uint32_t k, float *threshold);

/* ======================= several GPUs: a group of devices ================================
 * ImageProcessor::new (core/src/lib.rs:38-65) picks ONE adapter; the reference has no multi-device path.  A kmg_group is the
 * same constructor over a device LIST: one kmg_processor, one compute stream and one RCCL communicator rank per device.
 * RCCL is loaded at run time (dlopen of librccl.so.1 -- the copy a host process already maps, e.g. PyTorch's, is reused),
 * so a single-GPU host needs no RCCL at all.  The path's one exchange step is ncclAllReduce(sum) of the
 * k x 4 int64 accumulators over xGMI between the assign pass and the centroid update; because the sums are exact integers,
 * centroids and labels are bit-identical for any number of devices.
 *
 *   one process, n devices   kmg_group_create: ncclCommInitAll, one worker thread per device inside the library; the
 *                            host-buffer calls kmg_group_{palette, find, reduce} tile one image in row bands (device g of G
 *                            owns rows [g H / G, (g + 1) H / G)), kmg_group_reduce_batch places whole images (BASELINE
 *                            config 4: no collective at all).
 *   one process per GPU      rank 0 calls kmg_group_unique_id, the host runtime hands the 128 bytes to every process (MPI,
 *                            a file, torch.distributed ...), each calls kmg_group_create_rank(first_rank = its rank): the
 *                            kmg_group_lloyd_* calls then drive THIS process's band(s) of the sharded image.
 * Ranks are numbered first_rank + i for the group's local device i; `world` = ranks over all processes.
 *
 * Failures.  A call that issues no collective (kmg_group_find, kmg_group_reduce_batch, kmg_group_palette / _reduce unless the
 * k-means itself runs sharded) fails like its single-device counterpart: the error is returned, the group stays usable.  A rank
 * that fails where a collective may be in flight (kmg_group_lloyd_*, the sharded full-resolution k-means) would leave its peers
 * waiting inside RCCL for ever, so the communicators of this process are aborted (ncclCommAbort) and the group is BROKEN: every
 * later call on it returns KMG_ERR_HIP at once; destroy it and create a new one (the other processes of a multi-process world
 * see their own collectives fail or time out and must do the same).  No C++ exception crosses this ABI: std::bad_alloc comes
 * back as KMG_ERR_OUT_OF_MEMORY, anything else as KMG_ERR_HIP with the text in kmg_last_error().                              */
#define KMG_MAX_DEVICES 16
#define KMG_UNIQUE_ID_BYTES 128
/* kmg_group_options.flags */
#define KMG_GROUP_FORCE_COLLECTIVES 1u /* issue every collective even in a world of ONE rank (a one-GPU box then drives the
                                          RCCL calls exactly as a multi-rank job does: tests)                              */
#define KMG_GROUP_LOOPBACK          2u /* exchange through this process's device memory instead of RCCL; the device list may
                                          then name a device more than once (RCCL refuses two ranks on one device) -- the
                                          multi-rank code path on a one-GPU box (tests); one process only                  */
typedef struct kmg_group_options {
    uint32_t struct_size;               /* sizeof(kmg_group_options)                                                    */
    uint32_t n_devices;                 /* 0 = every visible HIP device, in ordinal order                               */
    int32_t  devices[KMG_MAX_DEVICES];  /* HIP device ordinals of the n_devices local ranks                             */
    uint32_t flags;                     /* KMG_GROUP_*                                                                  */
    kmg_options processor;              /* options of every member processor (its `device` field is ignored)            */
} kmg_group_options;

typedef struct kmg_group kmg_group;
typedef struct kmg_group_lloyd kmg_group_lloyd;

KMG_API void kmg_default_group_options(kmg_group_options *opt);
KMG_API int kmg_group_create(const kmg_group_options *opt, kmg_group **out);
KMG_API int kmg_group_unique_id(uint8_t id[KMG_UNIQUE_ID_BYTES]);
KMG_API int kmg_group_create_rank(const kmg_group_options *opt, const uint8_t id[KMG_UNIQUE_ID_BYTES], uint32_t first_rank,
                                  uint32_t world, kmg_group **out);
KMG_API void kmg_group_destroy(kmg_group *g);
/* n_local = devices of this process, first_rank / world as above, rccl_version = ncclGetVersion() or 0 when RCCL was not
 * needed (one rank without KMG_GROUP_FORCE_COLLECTIVES, or KMG_GROUP_LOOPBACK).  Any out pointer may be NULL.            */
KMG_API int kmg_group_info(kmg_group *g, uint32_t *n_local, uint32_t *first_rank, uint32_t *world, int *rccl_version);
/* local device i's processor / compute stream (hipStream_t): everything the group enqueues for that device runs on it   */
KMG_API kmg_processor *kmg_group_processor(kmg_group *g, uint32_t i);
KMG_API void *kmg_group_stream(kmg_group *g, uint32_t i);

/* ---- ImageProcessor::{palette, find, reduce} (core/src/lib.rs:67-164) on a one-process group: same arguments and results
 * as kmg_palette / kmg_find / kmg_reduce, byte for byte.  Every device uploads its band (and one halo row for the bilinear
 * shrink), shrinks its share of the <= 256-pixel working image (structures.rs:67-182), device 0 runs the k-means of that tiny
 * image (launch-bound: nothing to shard), and every device runs the output pass of its band -- the step that touches every
 * pixel -- and downloads it.  With shrink_max_dim = 0 and an image of at least 2^20 pixels the k-means itself runs sharded
 * (kmg_group_lloyd_*: initialisation, loop and the RCCL all-reduce per iteration).                                        */
KMG_API int kmg_group_palette(kmg_group *g, const uint8_t *rgba, uint32_t width, uint32_t height, uint32_t color_count, int algo,
                              uint8_t *out_rgba, uint32_t *out_count);
KMG_API int kmg_group_find(kmg_group *g, const uint8_t *rgba, uint32_t width, uint32_t height, const uint8_t *palette_rgba,
                           uint32_t n_colors, int mode, uint8_t *out_rgba);
KMG_API int kmg_group_reduce(kmg_group *g, const uint8_t *rgba, uint32_t width, uint32_t height, uint32_t color_count, int algo,
                             int mode, uint8_t *out_rgba);
/* A batch of images, WHOLE images per device (image i on local device i % n_local), no collective: kmg_reduce of every image
 * on its device's processor, the devices side by side (BASELINE config 4 as this build places it).  Returns the first failure.  */
KMG_API int kmg_group_reduce_batch(kmg_group *g, uint32_t n_images, const uint8_t *const *rgba, const uint32_t *widths,
                                   const uint32_t *heights, uint32_t color_count, int algo, int mode, uint8_t *const *out_rgba);

/* ---- ChooseCentroidModule::compute (core/src/modules.rs:763-840) over row bands that are resident on the group's devices.
 * d_rgba[i] (device memory of local device i) holds image rows [row0[i], row0[i] + rows[i]) of a width x height image,
 * d_labels[i] (optional, may be NULL as a whole) receives that band's u32 label map; rows[i] may be 0.  The bands of all
 * ranks of the world tile the image.  _bind only records the bands (it may be called again for a new image); the calls below
 * enqueue on the devices' compute streams and return -- only _run, _sync and _get_centroids synchronise.
 *   _init       PlusPlusInitModule::compute (modules.rs:946-1246) sharded: per centroid ONE all-gather of every band's
 *               {64-bit arg-max key, colour of the pixel it names} (kmg_lloyd_init_step / _init_pick_band); the largest key wins
 *   _prime      the initial assignment (operations.rs:75-83) with its sums, all-reduced
 *   _step       one iteration (modules.rs:769-800): centroid update from the global sums, labels + sums of the new
 *               assignment, all-reduce of the sums
 *   _run        _prime, then _step until the convergence count read every check_period-th iteration reaches k or
 *               max_iterations (kmg_options of the group's processors); *iterations as kmg_lloyd_run
 * flags (kmg_group_lloyd_bind):                                                                                            */
#define KMG_GROUP_CELLS   1u /* strong scaling of ONE image (colour table, k <= 256): the cube pass is sharded by cells of the
                                colour cube as well -- band histograms all-reduced once per image, per iteration the k x 4
                                all-reduce and an in-place all-gather of the label tables (kmg_lloyd_set_cell_share)        */
#define KMG_GROUP_OVERLAP 2u /* the all-reduce of the sums runs on a second stream beside the label pass (two cross-stream
                                dependencies per iteration) instead of in line on the compute stream                       */
#define KMG_GROUP_FUSED_UPDATE 4u /* _prime and _step perform the update on the last launch of the assignment that produced its sums:
                                     per call still one assignment with its label map and one update, shifted by half a step.  A
                                     world of ONE rank without collectives: kmg_lloyd_assign_update.  With KMG_GROUP_CELLS (bands
                                     with rows and label maps): the cube pass adds into the accumulators, the band's label pass
                                     updates from the all-reduced sums and clears them (kmg_lloyd_accumulate_into /
                                     _labels_from_tables_update: two launches per iteration instead of four).  Refused by _run,
                                     which reads the convergence count between update and re-assignment                       */
KMG_API int kmg_group_lloyd_create(kmg_group *g, uint32_t k, kmg_group_lloyd **out);
KMG_API void kmg_group_lloyd_destroy(kmg_group_lloyd *gl);
KMG_API int kmg_group_lloyd_bind(kmg_group_lloyd *gl, const uint8_t *const *d_rgba, const uint32_t *row0, const uint32_t *rows,
                                 uint32_t width, uint32_t height, uint32_t *const *d_labels, uint32_t flags);
KMG_API int kmg_group_lloyd_set_centroids(kmg_group_lloyd *gl, const float *centroids4);
KMG_API int kmg_group_lloyd_get_centroids(kmg_group_lloyd *gl, float *centroids4);
KMG_API int kmg_group_lloyd_init(kmg_group_lloyd *gl);
KMG_API int kmg_group_lloyd_prime(kmg_group_lloyd *gl);
KMG_API int kmg_group_lloyd_step(kmg_group_lloyd *gl);
KMG_API int kmg_group_lloyd_sync(kmg_group_lloyd *gl);
KMG_API int kmg_group_lloyd_run(kmg_group_lloyd *gl, uint32_t *iterations);
/* local device i's kmg_lloyd (profiling, statistics) and "table" (1) / "scan" (0) strategy of its band, once primed          */
KMG_API kmg_lloyd *kmg_group_lloyd_member(kmg_group_lloyd *gl, uint32_t i, int *strategy);

/* ---- a BATCH of images, each tiled over ALL ranks in row bands (BASELINE config 4 as north_star words it: "16 x 8192^2 tiled across 8
 * GPUs with one centroid all-reduce"; SURVEY 8e: "batch the 16 images' accumulators into one collective").  Every image is an
 * independent k-means problem with the same k; the accumulators of the whole batch are ONE block of n_images x k x 4 int64 per
 * rank, so an iteration costs a single ncclAllReduce of n_images * k * 32 bytes instead of one per image; the sharded
 * initialisation batches its keys and colours the same way.  (kmg_group_reduce_batch PLACES whole images instead -- zero
 * collectives, the faster split whenever the batch has at least as many images as the node has GPUs; this is for batches of fewer,
 * larger images, and for the configuration as worded.)
 *   _create_batch   like _create, n_images >= 1 (1 = kmg_group_lloyd_create)
 *   _bind_batch     d_rgba / row0 / rows / d_labels are indexed [image * n_local + local device], widths / heights [image];
 *                   KMG_GROUP_CELLS is refused (it shards ONE image's cube pass; a batch's cube passes already fill the ranks)
 *   _set / _get_centroids_image     one image's centroid table
 *   _init / _prime / _step / _sync  as above, over all images: one collective per exchange
 *   _run_batch      ChooseCentroidModule::compute for every image: an image whose convergence count reaches k at one of its
 *                   every-check_period checks stops being updated (its rows stay in the collective, zero); iterations[image] as
 *                   kmg_lloyd_run.  Results per image are those of kmg_lloyd_run on the whole image, bit for bit.              */
KMG_API int kmg_group_lloyd_create_batch(kmg_group *g, uint32_t k, uint32_t n_images, kmg_group_lloyd **out);
KMG_API int kmg_group_lloyd_bind_batch(kmg_group_lloyd *gl, const uint8_t *const *d_rgba, const uint32_t *row0, const uint32_t *rows,
                                       const uint32_t *widths, const uint32_t *heights, uint32_t *const *d_labels, uint32_t flags);
KMG_API int kmg_group_lloyd_set_centroids_image(kmg_group_lloyd *gl, uint32_t image, const float *centroids4);
KMG_API int kmg_group_lloyd_get_centroids_image(kmg_group_lloyd *gl, uint32_t image, float *centroids4);
KMG_API int kmg_group_lloyd_run_batch(kmg_group_lloyd *gl, uint32_t *iterations);

#ifdef __cplusplus
}
#endif
#endif /* KMEANS_HIP_H */
