/*
 * kmg_oracle.h -- CPU ORACLE for the Lloyd-iteration hot path of redwarp/kmeans-gpu.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load or call it.  The shipped library
 * (libkmeans_hip.so) never links, loads or falls back to anything in oracle/.
 *
 * What it is: a plain-C, single-source restatement of the reference's WGSL compute
 * shaders and of the host loop that sequences them, in IEEE-754 binary32 arithmetic
 * with every operation order written out (compile with -ffp-contract=off; the only
 * fused operations are the explicit fmaf() calls).  The reference itself (Rust + wgpu
 * + WGSL on Vulkan) cannot be built or run in this environment, so the oracle is
 * pinned against the reference's committed golden images and its shader-test
 * known-answer values instead (tests/test_oracle_golden.py):
 *   - the three `find` goldens reproduce bit-exactly,
 *   - the `reduce -c 8` goldens land on the same 8 colours within +-1 LSB,
 *   - cie94 KAT 19.094658 +- 0.01 (core/src/shader_tests.rs:180-186).
 *
 * Spec items S1..S12 refer to SURVEY.md section 8(a).  All reference citations are
 * relative to /root/reference.
 *
 * Where the reference leaves arithmetic implementation-defined (WGSL pow/sin/division
 * precision, summation order of the look-back scan, hardware bilinear weights) the
 * oracle fixes ONE definition, stated next to the function, and the HIP path is
 * required to match the oracle bit-for-bit:
 *   - pow(c, 2.4) for the 256 possible sRGB inputs: correctly rounded (double pow -> f32)
 *   - pow(t, 1/3): correctly rounded f32 cube root
 *   - per-cluster sums: exact integers of round-to-nearest-even(Lab * 2^20) per pixel
 *     (order independent, hence identical for any tiling / GPU count)
 *   - arg-min: over the LITERAL distance_cie94 with strict '<' (first minimum wins), exactly as
 *     find_centroid.wgsl:32-41 and mix_colors.wgsl:73-80 do.  (The squared, divide-free key the GPU
 *     kernels order by before settling near-ties with the literal distance is kept as
 *     orc_assign(literal = 2) / orc_cie94_key for tests; ordered by it alone, about one colour
 *     in 10^7 per 64 centroids would get another label -- tests/test_oracle_golden.py.)
 */
#ifndef KMG_ORACLE_H
#define KMG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FIX_SHIFT 20              /* Lab fixed point: q = rint(x * 2^20) */
#define ORC_FIX_SCALE 1048576.0f

/* modes, mirror core/src/lib.rs:215-253 */
enum { ORC_MODE_REPLACE = 0, ORC_MODE_DITHER = 1, ORC_MODE_MELD = 2 };

/* ---- S1: RGBA8 -> Lab (core/shaders/converters/rgb_to_lab.wgsl:11-80) ---- */
void  orc_srgb_lut(float lut100[256]);            /* (srgb decode) * 100 for each byte */
float orc_cbrt(float x);                          /* correctly rounded cube root     */
float orc_pow_inv_2p4(float c);                   /* lab_to_rgb.wgsl pow(c, 1/2.4), the fixed binary64 evaluation */
void  orc_rgb_to_lab(const uint8_t *rgba, uint64_t n, float *lab3);

/* ---- S3: CIE94 (core/shaders/functions/delta_e.wgsl:1-22) ---- */
float orc_cie94(const float one[3], const float second[3]);          /* literal form */
float orc_cie94_key(const float one[3], const float second[3]);      /* squared key  */

/* ---- S2: assign (core/shaders/find_centroid.wgsl:15-44) ----
 * centroids: k x 4 floats (L,a,b,pad) like the reference's vec4 array
 * (core/src/structures.rs:501-521).  literal: 0 = the literal distance with the per-pixel / per-centroid
 * terms hoisted (same floats, fast), 1 = orc_cie94 per pair (the un-hoisted restatement), 2 = ordered
 * by the squared key alone (NOT the reference's arg-min; for tests). */
void orc_assign(const float *lab3, uint64_t n, const float *centroids4, uint32_t k,
                int literal, uint32_t *labels);

/* ---- S4: per-cluster sums (core/shaders/choose_centroid.wgsl:75-178, semantics only)
 * acc: k x 4 int64 = (sum qL, sum qa, sum qb, count), q = rint(x * 2^20).           */
void orc_accumulate(const float *lab3, const uint32_t *labels, uint64_t n, uint32_t k,
                    int64_t *acc4);

/* ---- S5: pick (core/shaders/choose_centroid.wgsl:180-206) ----
 * Updates centroids4 in place from acc4; returns the number of clusters whose move
 * is < convergence (empty clusters count as not converged).                         */
uint32_t orc_finalize(const int64_t *acc4, uint32_t k, float convergence, float *centroids4);

/* ---- S6: Lloyd host loop (core/src/modules.rs:763-840) ----
 * On entry centroids4 holds the initial centroids.  labels (n) receives the final
 * assignment.  Returns the value of `iteration` at which the loop stopped
 * (max_iterations-1 when it ran out).                                                */
uint32_t orc_lloyd(const float *lab3, uint64_t n, uint32_t k, float *centroids4,
                   uint32_t *labels, uint32_t max_iterations, uint32_t check_period,
                   float convergence);

/* ---- S12: deterministic farthest-point init
 * (core/shaders/plus_plus_init.wgsl:58-186, core/shaders/kmeans++_calc_diff.wgsl:16-33,
 *  core/src/modules.rs:946-1246) ---- */
float orc_rand(float seed);
void  orc_init_centroids(const float *lab3, uint32_t w, uint32_t h, uint32_t k,
                         float *centroids4);

/* ---- S11: shrink (core/src/structures.rs:67-182, core/shaders/resize.wgsl:7-18) ---- */
void orc_resized_dims(uint32_t w, uint32_t h, uint32_t max_size, uint32_t *nw, uint32_t *nh);
void orc_resize(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh,
                uint8_t *out_rgba);

/* ---- S7: ordered dither (core/shaders/mix_colors.wgsl:21-27,50-83,94-113) ----
 * out_index[n]: index of the chosen centroid, or k when the (1e4,1e4,1e4) sentinel
 * of mix_colors.wgsl:70 survives.                                                    */
float orc_dither_threshold(const float *centroids4, uint32_t k);
void  orc_dither(const float *lab3, uint32_t w, uint32_t h, const float *centroids4,
                 uint32_t k, uint32_t *out_index);

/* ---- meld (core/shaders/mix_colors.wgsl:29-48,85-90,117-135) ---- */
void  orc_meld(const float *lab3, uint32_t w, uint32_t h, const float *centroids4,
               uint32_t k, float *out_lab3);

/* ---- S9: Lab -> RGBA8 unorm (core/shaders/converters/lab_to_rgb.wgsl:11-81) ---- */
void orc_lab_to_rgba8(const float *lab3, uint64_t n, uint8_t *rgba);

/* ---- S10: palette-crate 0.7.3 conversions used on the HOST by the reference
 * (core/src/structures.rs:523-553 fixed_centroids, :581-617 pull_values)  ---- */
void orc_palette_srgb8_to_lab(const uint8_t rgb[3], float lab[3]);
void orc_palette_lab_to_srgb8(const float lab[3], uint8_t rgb[3]);

/* ---- end-to-end operations (core/src/lib.rs:67-164, core/src/operations.rs:15-271) ---- */
/* extract_palette_kmeans: shrink -> Lab -> init -> assign -> Lloyd.  centroids4: k x 4. */
uint32_t orc_extract_palette_kmeans(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k,
                                    uint32_t shrink_max_dim, float *centroids4);
/* find_colors / dither_colors / meld_colors at full resolution with given Lab centroids */
void orc_apply(const uint8_t *rgba, uint32_t w, uint32_t h, const float *centroids4,
               uint32_t k, int mode, uint8_t *out_rgba);
/* ImageProcessor::find */
void orc_find(const uint8_t *rgba, uint32_t w, uint32_t h, const uint8_t *palette_rgba,
              uint32_t n_colors, int mode, uint8_t *out_rgba);
/* ImageProcessor::reduce with Algorithm::Kmeans */
void orc_reduce(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, int mode,
                uint8_t *out_rgba);
/* ImageProcessor::palette with Algorithm::Kmeans: k RGBA8 colours sorted by Lab L */
void orc_palette(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, uint8_t *out_rgba);

/* ---- Algorithm::Octree (core/src/octree.rs, core/src/lib.rs:288-331).  out_rgba: capacity 4*k bytes
 * (the pixel count when k is larger); returns the number of colours (<= k).                       */
uint32_t orc_octree_palette(const uint8_t *rgba, uint64_t n, uint32_t color_count, uint8_t *out_rgba);
uint32_t orc_palette_octree(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, uint8_t *out_rgba);
void     orc_reduce_octree(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, int mode, uint8_t *out_rgba);

/* ---- synthetic inputs: splitmix64, R=r&255, G=(r>>8)&255, B=(r>>16)&255 */
void orc_synth_uniform(uint64_t seed, uint64_t n, uint8_t *rgba);

/* one fused assign + accumulate pass straight from RGBA8 (what bench.py times on the CPU) */
void orc_assign_accumulate_rgba(const uint8_t *rgba, uint64_t n, const float *centroids4,
                                uint32_t k, uint32_t *labels, int64_t *acc4);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
