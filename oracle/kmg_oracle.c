/*
 * kmg_oracle.c -- CPU ORACLE (test infrastructure only; see kmg_oracle.h).
 *
 * Plain C99 restatement of the reference's WGSL kernels + host sequencing for the
 * Lloyd-iteration hot path.  Compile with:  gcc -O2 -ffp-contract=off -fno-fast-math
 * (-mfma only makes the explicit fmaf() calls fast; it does not change results).
 * Citations are relative to /root/reference.
 */
#include "kmg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

/* ------------------------------------------------------------------------------------ */
/* S1  RGBA8 -> Lab          core/shaders/converters/rgb_to_lab.wgsl                     */
/* ------------------------------------------------------------------------------------ */

/* rgb_to_lab.wgsl:16-33: sRGB decode of c = byte/255 (textureLoad of rgba8unorm), then
 * "* 100.0".  Only 256 inputs exist, so the table IS the function.  WGSL pow precision is
 * implementation defined; the oracle fixes it as the correctly rounded result.          */
void orc_srgb_lut(float lut100[256])
{
    for (int v = 0; v < 256; ++v) {
        float c = (float)v / 255.0f;
        float lin;
        if (c > 0.04045f)
            lin = (float)pow((double)((c + 0.055f) / 1.055f), 2.4);
        else
            lin = c / 12.92f;
        lut100[v] = lin * 100.0f;
    }
}

/* rgb_to_lab.wgsl:45,50,55: pow(t, 1.0/3.0).  Fixed as the correctly rounded f32 cube
 * root (double cbrt rounded once; exhaustively equal to the x87 long-double result on
 * [1e-3, 2], tests/native/check_math.cpp via tests/test_host_math.py).                                              */
float orc_cbrt(float x) { return (float)cbrt((double)x); }

static inline float lab_f(float t)
{
    /* rgb_to_lab.wgsl:44-58 */
    if (t > 0.008856f) return orc_cbrt(t);
    return FMA(7.787f, t, 16.0f / 116.0f);
}

static const float *srgb_lut(void)
{
    static float lut[256];
    static int ready = 0;
    if (!ready) {
#pragma omp critical(orc_lut)
        {
            if (!ready) { orc_srgb_lut(lut); ready = 1; }
        }
    }
    return lut;
}

static inline void pixel_to_lab(const float *lut, const uint8_t *px, float *lab)
{
    float r = lut[px[0]], g = lut[px[1]], b = lut[px[2]];      /* alpha ignored, :78 */
    /* rgb_to_lab.wgsl:5-9,38: column-major mat3x3 * vec3 = col0*r + col1*g + col2*b,
     * evaluated as a multiply followed by two fused multiply-adds.                     */
    float X = FMA(0.1804375f, b, FMA(0.3575761f, g, 0.4124564f * r));
    float Y = FMA(0.0721750f, b, FMA(0.7151522f, g, 0.2126729f * r));
    float Z = FMA(0.9503041f, b, FMA(0.1191920f, g, 0.0193339f * r));
    /* rgb_to_lab.wgsl:42-44 */
    float fx = lab_f(X / 95.0489f);
    float fy = lab_f(Y / 100.0f);
    float fz = lab_f(Z / 108.8840f);
    /* rgb_to_lab.wgsl:59-61 */
    lab[0] = FMA(116.0f, fy, -16.0f);
    lab[1] = 500.0f * (fx - fy);
    lab[2] = 200.0f * (fy - fz);
}

void orc_rgb_to_lab(const uint8_t *rgba, uint64_t n, float *lab3)
{
    const float *lut = srgb_lut();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) pixel_to_lab(lut, rgba + 4 * i, lab3 + 3 * i);
}

/* ------------------------------------------------------------------------------------ */
/* S3  CIE94                 core/shaders/functions/delta_e.wgsl:1-22                    */
/* ------------------------------------------------------------------------------------ */

static inline float chroma(float a, float b) { return sqrtf(a * a + b * b); } /* :8-9 */

/* Literal form, operation for operation (no fused ops).  Asymmetric: SC/SH use C1 of
 * `one`.                                                                                */
float orc_cie94(const float one[3], const float second[3])
{
    const float K1 = 0.045f, K2 = 0.015f;
    float dL = one[0] - second[0];
    float da = one[1] - second[1];
    float db = one[2] - second[2];
    float C1 = chroma(one[1], one[2]);
    float C2 = chroma(second[1], second[2]);
    float dC = C1 - C2;
    float dH = sqrtf(fmaxf((da * da) + (db * db) - (dC * dC), 0.0f));
    float SL = 1.0f;
    float SC = 1.0f + K1 * C1;
    float SH = 1.0f + K2 * C1;
    float tL = dL / SL, tC = dC / SC, tH = dH / SH;
    return sqrtf(tL * tL + tC * tC + tH * tH);
}

/* The arg-min of the reference is over the LITERAL distance above (find_centroid.wgsl:32-41,
 * mix_colors.wgsl:73-80).  lit_terms() is that distance with what depends on `one` only (C1, SC, SH)
 * and C2 hoisted out of the centroid loop: the same operations on the same operands, hence the same
 * float as orc_cie94() -- orc_assign(literal = 1) runs the un-hoisted form and the tests compare.
 *
 * key_terms() is the squared form the GPU kernels ORDER by before they settle near-ties with the
 * literal distance:  key = dL^2 + dC^2 * (1/SC)^2 + max(da^2 + db^2 - dC^2, 0) * (1/SH)^2.
 * It is kept here (orc_assign(literal = 2), orc_cie94_key) only so that tests can show where ordering
 * by the key alone would differ from the reference.                                             */
typedef struct { float L, a, b, C, wC, wH; } orc_px;

static inline orc_px px_terms(const float lab[3])
{
    orc_px p;
    p.L = lab[0]; p.a = lab[1]; p.b = lab[2];
    p.C = chroma(lab[1], lab[2]);
    float SC = 1.0f + 0.045f * p.C;
    float SH = 1.0f + 0.015f * p.C;
    float iSC = 1.0f / SC, iSH = 1.0f / SH;
    p.wC = iSC * iSC;
    p.wH = iSH * iSH;
    return p;
}

static inline float lit_terms(const orc_px *p, float L2, float a2, float b2, float C2)
{
    /* delta_e.wgsl:4-21 with C1 = p->C and C2 supplied */
    float dL = p->L - L2, da = p->a - a2, db = p->b - b2, dC = p->C - C2;
    float dH = sqrtf(fmaxf((da * da) + (db * db) - (dC * dC), 0.0f));
    float SC = 1.0f + 0.045f * p->C;
    float SH = 1.0f + 0.015f * p->C;
    float tL = dL / 1.0f, tC = dC / SC, tH = dH / SH;
    return sqrtf(tL * tL + tC * tC + tH * tH);
}

static inline float key_terms(const orc_px *p, float L2, float a2, float b2, float C2)
{
    float dL = p->L - L2, da = p->a - a2, db = p->b - b2, dC = p->C - C2;
    float dC2 = dC * dC;
    float t = FMA(db, db, da * da);
    float h = fmaxf(t - dC2, 0.0f);
    return FMA(h, p->wH, FMA(dC2, p->wC, dL * dL));
}

float orc_cie94_key(const float one[3], const float second[3])
{
    orc_px p = px_terms(one);
    return key_terms(&p, second[0], second[1], second[2], chroma(second[1], second[2]));
}

/* ------------------------------------------------------------------------------------ */
/* S2  assign                core/shaders/find_centroid.wgsl:15-44                       */
/* ------------------------------------------------------------------------------------ */

static inline uint32_t argmin_lit(const orc_px *p, const float *cent5, uint32_t k)
{
    /* find_centroid.wgsl:29-41: min_distance = 100000.0, found_index = 0, strict '<' */
    float best = 100000.0f;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < k; ++j) {
        float d = lit_terms(p, cent5[5 * j], cent5[5 * j + 1], cent5[5 * j + 2], cent5[5 * j + 3]);
        if (d < best) { best = d; idx = j; }
    }
    return idx;
}

static inline uint32_t argmin_key(const orc_px *p, const float *cent5, uint32_t k)
{
    /* ordering by the squared key alone (NOT the reference's definition; see above) */
    float best = 100000.0f * 100000.0f;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < k; ++j) {
        float d = key_terms(p, cent5[5 * j], cent5[5 * j + 1], cent5[5 * j + 2], cent5[5 * j + 3]);
        if (d < best) { best = d; idx = j; }
    }
    return idx;
}

static float *make_cent5(const float *centroids4, uint32_t k)
{
    float *c5 = (float *)malloc(sizeof(float) * 5 * (k ? k : 1));
    for (uint32_t j = 0; j < k; ++j) {
        c5[5 * j] = centroids4[4 * j];
        c5[5 * j + 1] = centroids4[4 * j + 1];
        c5[5 * j + 2] = centroids4[4 * j + 2];
        c5[5 * j + 3] = chroma(centroids4[4 * j + 1], centroids4[4 * j + 2]);
        c5[5 * j + 4] = 0.0f;
    }
    return c5;
}

void orc_assign(const float *lab3, uint64_t n, const float *centroids4, uint32_t k,
                int literal, uint32_t *labels)
{
    if (literal == 1) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            float best = 100000.0f;
            uint32_t idx = 0;
            for (uint32_t j = 0; j < k; ++j) {
                float d = orc_cie94(lab3 + 3 * i, centroids4 + 4 * j);
                if (d < best) { best = d; idx = j; }
            }
            labels[i] = idx;
        }
        return;
    }
    float *c5 = make_cent5(centroids4, k);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        orc_px p = px_terms(lab3 + 3 * i);
        labels[i] = literal == 2 ? argmin_key(&p, c5, k) : argmin_lit(&p, c5, k);
    }
    free(c5);
}

/* ------------------------------------------------------------------------------------ */
/* S4  per-cluster sums      core/shaders/choose_centroid.wgsl:75-178 (semantics)        */
/* ------------------------------------------------------------------------------------ */

static inline int64_t fix(float x) { return (int64_t)(int32_t)rintf(x * ORC_FIX_SCALE); }

void orc_accumulate(const float *lab3, const uint32_t *labels, uint64_t n, uint32_t k,
                    int64_t *acc4)
{
    /* choose_centroid.wgsl:97-104: sum of the Lab of pixels whose label == k, plus a
     * count.  The reference adds f32 values in a scheduling-dependent order; the oracle
     * adds exact integers, so the order (and the tiling) cannot matter.                 */
    memset(acc4, 0, sizeof(int64_t) * 4 * k);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = labels[i];
        if (c >= k) continue;
        acc4[4 * c + 0] += fix(lab3[3 * i + 0]);
        acc4[4 * c + 1] += fix(lab3[3 * i + 1]);
        acc4[4 * c + 2] += fix(lab3[3 * i + 2]);
        acc4[4 * c + 3] += 1;
    }
}

/* ------------------------------------------------------------------------------------ */
/* S5  pick                  core/shaders/choose_centroid.wgsl:180-206                   */
/* ------------------------------------------------------------------------------------ */

uint32_t orc_finalize(const int64_t *acc4, uint32_t k, float convergence, float *centroids4)
{
    uint32_t converged = 0;
    for (uint32_t c = 0; c < k; ++c) {
        int64_t count = acc4[4 * c + 3];
        if (count > 0) {                                         /* :185 */
            float nw[3], prev[3];
            for (int j = 0; j < 3; ++j) {
                double mean = ((double)acc4[4 * c + j] / (double)count) * (1.0 / 1048576.0);
                nw[j] = (float)mean;                             /* :186 sum / count */
                prev[j] = centroids4[4 * c + j];
            }
            centroids4[4 * c + 0] = nw[0];
            centroids4[4 * c + 1] = nw[1];
            centroids4[4 * c + 2] = nw[2];
            centroids4[4 * c + 3] = 1.0f;
            if (orc_cie94(nw, prev) < convergence) converged += 1;   /* :191 */
        }
        /* else: centroid unchanged, convergence[k] = 0  (:192-194) */
    }
    return converged;                                            /* :196-202 */
}

/* ------------------------------------------------------------------------------------ */
/* S6  Lloyd loop            core/src/modules.rs:763-840, operations.rs:75-85            */
/* ------------------------------------------------------------------------------------ */

uint32_t orc_lloyd(const float *lab3, uint64_t n, uint32_t k, float *centroids4,
                   uint32_t *labels, uint32_t max_iterations, uint32_t check_period,
                   float convergence)
{
    int64_t *acc = (int64_t *)malloc(sizeof(int64_t) * 4 * k);
    orc_assign(lab3, n, centroids4, k, 0, labels);               /* operations.rs:75-83 */
    uint32_t it = 0;
    for (it = 0; it < max_iterations; ++it) {                    /* modules.rs:769 */
        orc_accumulate(lab3, labels, n, k, acc);                 /* modules.rs:773-788 */
        uint32_t converged = orc_finalize(acc, k, convergence, centroids4);
        orc_assign(lab3, n, centroids4, k, 0, labels);           /* modules.rs:793-800 */
        if (it > 0 && it % check_period == 0) {                  /* modules.rs:802 */
            if (converged >= k) break;                           /* modules.rs:826-831 */
        }
    }
    free(acc);
    return it < max_iterations ? it : max_iterations - 1;
}

/* ------------------------------------------------------------------------------------ */
/* S12 init                  core/shaders/plus_plus_init.wgsl, kmeans++_calc_diff.wgsl   */
/* ------------------------------------------------------------------------------------ */

/* plus_plus_init.wgsl:58-60: fract(sin(dot(vec2(seed), vec2(12.9898,78.233))) * 43758.5453)
 * in f32.  Only seeds 42 and 12 are ever used (:163-164); in IEEE f32 they give exactly
 * 0.5625 and 0.93359375.                                                                */
float orc_rand(float seed)
{
    float d = seed * 12.9898f + seed * 78.233f;
    float p = sinf(d) * 43758.5453f;
    return p - floorf(p);
}

void orc_init_centroids(const float *lab3, uint32_t w, uint32_t h, uint32_t k,
                        float *centroids4)
{
    uint64_t n = (uint64_t)w * h;
    /* plus_plus_init.wgsl:161-168 `initial` */
    int32_t x0 = (int32_t)((float)w * orc_rand(42.0f));
    int32_t y0 = (int32_t)((float)h * orc_rand(12.0f));
    uint64_t i0 = (uint64_t)y0 * w + (uint64_t)x0;
    centroids4[0] = lab3[3 * i0]; centroids4[1] = lab3[3 * i0 + 1];
    centroids4[2] = lab3[3 * i0 + 2]; centroids4[3] = 1.0f;
    if (k == 1) return;

    float *dist = (float *)malloc(sizeof(float) * n);
    for (uint64_t i = 0; i < n; ++i) dist[i] = 1000000.0f;      /* calc_diff.wgsl:26 */

    /* The arg-max below is a fold over blocks of 16 pixels (one per GPU thread of the reference) in ascending
     * order with "a later block wins a tie"; its result is the LAST block attaining the largest block value.
     * Folding contiguous ranges of blocks separately (one per OpenMP thread, each from Candidate(0, 0.0)) and
     * then the range results in ascending order with the same rule gives exactly that block: same tie rule,
     * any number of threads.                                                                              */
    const int64_t n_blocks = (int64_t)((n + 15) / 16);
    int n_thr = omp_get_max_threads();
    if (n_thr < 1) n_thr = 1;
    float *part_d = (float *)malloc(sizeof(float) * (size_t)n_thr);
    uint32_t *part_i = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n_thr);

    for (uint32_t j = 1; j < k; ++j) {
        /* kmeans++_calc_diff.wgsl:27-30: min over centroids 0..j-1 of cie94(pixel, c).
         * min is exact, so keeping the running minimum equals recomputing it.           */
        const float *c = centroids4 + 4 * (j - 1);
        /* plus_plus_init.wgsl:62-68,84-143: arg-max.  selectCandidate(a,b) returns b only
         * when a.distance < b.distance.  Each thread folds its N_SEQ=16 consecutive pixels
         * starting from Candidate(0, 0.0) with the accumulator as `a` (earliest maximum
         * wins inside a thread); threads / workgroups are folded with the LATER one as `a`
         * (latest maximum wins across threads).                                          */
        for (int t = 0; t < n_thr; ++t) { part_d[t] = -1.0f; part_i[t] = 0; }   /* -1: range without blocks */
#pragma omp parallel num_threads(n_thr)
        {
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
            const int64_t b0 = n_blocks * t / nt, b1 = n_blocks * (t + 1) / nt;
            uint32_t best_idx = 0; float best_d = 0.0f;
            for (int64_t b = b0; b < b1; ++b) {
                uint64_t s0 = (uint64_t)b * 16, e = s0 + 16 < n ? s0 + 16 : n;
                uint32_t l_idx = 0; float l_d = 0.0f;
                for (uint64_t i = s0; i < e; ++i) {
                    float d = fminf(dist[i], orc_cie94(lab3 + 3 * i, c));
                    dist[i] = d;
                    if (l_d < d) { l_d = d; l_idx = (uint32_t)i; }
                }
                if (!(l_d < best_d)) { best_d = l_d; best_idx = l_idx; }
            }
            if (b1 > b0) { part_d[t] = best_d; part_i[t] = best_idx; }
        }
        uint32_t best_idx = 0; float best_d = 0.0f;
        for (int t = 0; t < n_thr; ++t)
            if (part_d[t] >= 0.0f && !(part_d[t] < best_d)) { best_d = part_d[t]; best_idx = part_i[t]; }
        /* plus_plus_init.wgsl:172-181 `pick` */
        centroids4[4 * j + 0] = lab3[3 * (uint64_t)best_idx + 0];
        centroids4[4 * j + 1] = lab3[3 * (uint64_t)best_idx + 1];
        centroids4[4 * j + 2] = lab3[3 * (uint64_t)best_idx + 2];
        centroids4[4 * j + 3] = 1.0f;
    }
    free(part_d);
    free(part_i);
    free(dist);
}

/* ------------------------------------------------------------------------------------ */
/* S11 shrink                core/src/structures.rs:67-89, core/shaders/resize.wgsl      */
/* ------------------------------------------------------------------------------------ */

void orc_resized_dims(uint32_t w, uint32_t h, uint32_t max_size, uint32_t *nw, uint32_t *nh)
{
    /* structures.rs:79-89 (f32 arithmetic, truncating cast, max(1)) */
    if (w > h) {
        uint32_t v = (uint32_t)((float)h * (float)max_size / (float)w);
        *nw = max_size; *nh = v > 1 ? v : 1;
    } else {
        uint32_t v = (uint32_t)((float)w * (float)max_size / (float)h);
        *nw = v > 1 ? v : 1; *nh = max_size;
    }
}

static inline uint8_t unorm8(float v)
{
    /* rgba8unorm store: clamp, scale, round to nearest even */
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)rintf(v * 255.0f);
}

void orc_resize(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh,
                uint8_t *out)
{
    /* resize.wgsl:15-16: uv = gid / dims (no half-texel offset); linear filter, clamp to
     * edge (structures.rs:121-131).  Filter weights are implementation defined in the
     * reference; the oracle uses exact f32 weights and lerps x first, then y.           */
#pragma omp parallel for schedule(static)
    for (int64_t gy = 0; gy < (int64_t)nh; ++gy) {
        float v = (float)gy / (float)nh;
        float ty = v * (float)h - 0.5f;
        float fy0 = floorf(ty);
        float wy = ty - fy0;
        int64_t y0 = (int64_t)fy0, y1 = y0 + 1;
        if (y0 < 0) y0 = 0; if (y1 < 0) y1 = 0;
        if (y0 > (int64_t)h - 1) y0 = h - 1; if (y1 > (int64_t)h - 1) y1 = h - 1;
        for (uint32_t gx = 0; gx < nw; ++gx) {
            float u = (float)gx / (float)nw;
            float tx = u * (float)w - 0.5f;
            float fx0 = floorf(tx);
            float wx = tx - fx0;
            int64_t x0 = (int64_t)fx0, x1 = x0 + 1;
            if (x0 < 0) x0 = 0; if (x1 < 0) x1 = 0;
            if (x0 > (int64_t)w - 1) x0 = w - 1; if (x1 > (int64_t)w - 1) x1 = w - 1;
            const uint8_t *p00 = rgba + 4 * ((uint64_t)y0 * w + x0);
            const uint8_t *p10 = rgba + 4 * ((uint64_t)y0 * w + x1);
            const uint8_t *p01 = rgba + 4 * ((uint64_t)y1 * w + x0);
            const uint8_t *p11 = rgba + 4 * ((uint64_t)y1 * w + x1);
            uint8_t *o = out + 4 * ((uint64_t)gy * nw + gx);
            for (int ch = 0; ch < 4; ++ch) {
                float t00 = (float)p00[ch] / 255.0f, t10 = (float)p10[ch] / 255.0f;
                float t01 = (float)p01[ch] / 255.0f, t11 = (float)p11[ch] / 255.0f;
                float top = FMA(wx, t10 - t00, t00);
                float bot = FMA(wx, t11 - t01, t01);
                o[ch] = unorm8(FMA(wy, bot - top, top));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* S7  ordered dither        core/shaders/mix_colors.wgsl                                */
/* ------------------------------------------------------------------------------------ */

static const uint32_t BAYER[16] = {0, 8, 2, 10, 12, 4, 14, 6, 3, 11, 1, 9, 15, 7, 13, 5}; /* :13-16 */

float orc_dither_threshold(const float *centroids4, uint32_t k)
{
    /* mix_colors.wgsl:53-67 (recomputed per pixel in the reference; it only depends on the
     * centroid table).  Requires k >= 2.                                                */
    const float *A = centroids4, *B = centroids4 + 4;
    float dAB = orc_cie94(A, B);
    for (uint32_t i = 2; i < k; ++i) {
        const float *ci = centroids4 + 4 * i;
        float dA = orc_cie94(ci, A);
        float dB = orc_cie94(ci, B);
        if (dA > dB && dA > dAB) { dAB = dA; B = ci; }
        else if (dB > dAB)       { dAB = dB; A = ci; }
    }
    return dAB / sqrtf((float)k);
}

void orc_dither(const float *lab3, uint32_t w, uint32_t h, const float *centroids4,
                uint32_t k, uint32_t *out_index)
{
    uint64_t n = (uint64_t)w * h;
    if (k == 1) {                                              /* :104-108 */
        for (uint64_t i = 0; i < n; ++i) out_index[i] = 0;
        return;
    }
    float thr = orc_dither_threshold(centroids4, k);
    float *c5 = make_cent5(centroids4, k);
    const float sentinel[3] = {10000.0f, 10000.0f, 10000.0f};  /* :73 */
    float sC = chroma(sentinel[1], sentinel[2]);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)h; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            uint64_t i = (uint64_t)y * w + x;
            /* :21-27 index_value, :70 "- 0.5" */
            float iv = (float)BAYER[(x % 4u) + ((uint32_t)y % 4u) * 4u] / 16.0f - 0.5f;
            float off = thr * iv;                               /* :72 threshold * index_value */
            float adj[3] = {lab3[3 * i] + off, lab3[3 * i + 1] + off, lab3[3 * i + 2] + off};
            orc_px p = px_terms(adj);
            /* :73-80 running minimum starting from the distance to the sentinel */
            float best = lit_terms(&p, sentinel[0], sentinel[1], sentinel[2], sC);
            uint32_t idx = k;
            for (uint32_t j = 0; j < k; ++j) {
                float d = lit_terms(&p, c5[5 * j], c5[5 * j + 1], c5[5 * j + 2], c5[5 * j + 3]);
                if (d < best) { best = d; idx = j; }
            }
            out_index[i] = idx;
        }
    }
    free(c5);
}

void orc_meld(const float *lab3, uint32_t w, uint32_t h, const float *centroids4,
              uint32_t k, float *out_lab3)
{
    uint64_t n = (uint64_t)w * h;
    if (k == 1) {                                              /* :127-131 */
        for (uint64_t i = 0; i < n; ++i) memcpy(out_lab3 + 3 * i, centroids4, 3 * sizeof(float));
        return;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float *color = lab3 + 3 * i;
        /* :29-48 two_closest_colors, literal distances, vec4(10000.0) sentinels */
        float closest[3] = {10000.0f, 10000.0f, 10000.0f};
        float second[3] = {10000.0f, 10000.0f, 10000.0f};
        float d_closest = orc_cie94(color, closest);
        float d_second = d_closest;
        for (uint32_t j = 0; j < k; ++j) {
            const float *t = centroids4 + 4 * j;
            float d = orc_cie94(color, t);
            if (d < d_closest) {
                memcpy(second, closest, sizeof second); d_second = d_closest;
                memcpy(closest, t, sizeof closest);     d_closest = d;
            } else if (d < d_second) {
                memcpy(second, t, sizeof second);       d_second = d;
            }
        }
        /* :85-90 */
        float factor = orc_cie94(color, second) / orc_cie94(closest, second);
        for (int c = 0; c < 3; ++c)
            out_lab3[3 * i + c] = factor * closest[c] + (1.0f - factor) * second[c];
    }
}

/* ------------------------------------------------------------------------------------ */
/* S9  Lab -> RGBA8          core/shaders/converters/lab_to_rgb.wgsl                     */
/* ------------------------------------------------------------------------------------ */

/* lab_to_rgb.wgsl:21-35 `pow(c, 1.0 / 2.4)`, c in (0.0031308, 1).  WGSL leaves pow's precision to the driver;
 * the definition fixed here is the following binary64 evaluation (IEEE +, -, *, /, sqrt; relative error ~1e-15
 * before the single rounding to binary32; equal to (float)pow((double)c, (double)(1.0f / 2.4f)) of a correctly
 * rounded libm except on ~1e-8 of the inputs, tests/test_oracle_golden.py), which the GPU library restates
 * operation by operation so that the bytes of meld outputs agree:
 *   y = f32(1 / 2.4) = 5/12 - d;  c^y = c^(5/12) exp(-d ln c);  c^(1/12) = sqrt(sqrt(cbrt c));
 *   ln c = 24 atanh((s - 1) / (s + 1)), s = c^(1/12)                                                   */
float orc_pow_inv_2p4(float c)
{
    if (!(c < 1.0f)) return 1.0f;                         /* clamps to 255 in the caller anyway */
    const double x = (double)c;
    double y = (double)orc_cbrt(c < 1.0e-3f ? 1.0e-3f : c);
    for (int i = 0; i < 2; ++i) y = y - (((y * y) * y) - x) / ((3.0 * y) * y);
    const double s = sqrt(sqrt(y));
    const double s2 = s * s, s4 = s2 * s2, p = s4 * s;
    const double z = (s - 1.0) / (s + 1.0), z2 = z * z;
    const double series = 1.0 + z2 * (1.0 / 3.0 + z2 * (1.0 / 5.0 + z2 * (1.0 / 7.0 + z2 * (1.0 / 9.0 + z2 * (1.0 / 11.0 + z2 * (1.0 / 13.0 + z2 * (1.0 / 15.0)))))));
    const double ln_c = 24.0 * (z * series);
    const double d = 5.0 / 12.0 - (double)(1.0f / 2.4f);
    const double t = -(d * ln_c);
    return (float)(p * (1.0 + t + (0.5 * t) * t));
}

static inline float srgb_encode(float c)
{
    if (c > 0.0031308f) return 1.055f * orc_pow_inv_2p4(c) - 0.055f;
    return 12.92f * c;
}

static inline float lab_finv(float t)
{
    /* lab_to_rgb.wgsl:45-59; pow(t, 3.0) restated as t*t*t */
    float t3 = t * t * t;
    if (t3 > 0.008856f) return t3;
    return (t - 16.0f / 116.0f) / 7.787f;
}

static void lab_to_rgba8_one(const float *lab, uint8_t *out)
{
    /* lab_to_rgb.wgsl:41-43 */
    float y = (lab[0] + 16.0f) / 116.0f;
    float x = lab[1] / 500.0f + y;
    float z = y - lab[2] / 200.0f;
    /* :61-63 */
    float X = lab_finv(x) * 95.0489f;
    float Y = lab_finv(y) * 100.0f;
    float Z = lab_finv(z) * 108.8840f;
    /* :12-14 */
    x = X / 100.0f; y = Y / 100.0f; z = Z / 100.0f;
    /* :5-9,16: column-major mat3x3 * vec3 */
    float r = FMA(-0.4985314f, z, FMA(-1.5371385f, y, 3.2404542f * x));
    float g = FMA(0.0415560f, z, FMA(1.8760108f, y, -0.9692660f * x));
    float b = FMA(1.0572252f, z, FMA(-0.2040259f, y, 0.0556434f * x));
    out[0] = unorm8(srgb_encode(r));
    out[1] = unorm8(srgb_encode(g));
    out[2] = unorm8(srgb_encode(b));
    out[3] = 255;                                              /* :37 alpha 1.0 */
}

void orc_lab_to_rgba8(const float *lab3, uint64_t n, uint8_t *rgba)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) lab_to_rgba8_one(lab3 + 3 * i, rgba + 4 * i);
}

/* ------------------------------------------------------------------------------------ */
/* S10 palette crate 0.7.3 (Cargo.lock:846-847), used on the host by the reference at    */
/*     core/src/structures.rs:533-536 and :601-607.  The crate is not vendored; this is   */
/*     its published algorithm (sRGB transfer function, sRGB/D65 matrices, CIE L*a*b*     */
/*     with epsilon = 216/24389, kappa = 24389/27, white = (0.95047, 1, 1.08883)).        */
/*     Pinned only end-to-end by the three `find` goldens.                                */
/* ------------------------------------------------------------------------------------ */

void orc_palette_srgb8_to_lab(const uint8_t rgb[3], float lab[3])
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

void orc_palette_lab_to_srgb8(const float lab[3], uint8_t rgb[3])
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
        float s = l <= 0.0031308f ? 12.92f * l
                                  : 1.055f * (float)pow((double)l, 1.0 / 2.4) - 0.055f;
        rgb[i] = unorm8(s);
    }
}

/* ------------------------------------------------------------------------------------ */
/* end-to-end operations     core/src/lib.rs, core/src/operations.rs                     */
/* ------------------------------------------------------------------------------------ */

uint32_t orc_extract_palette_kmeans(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k,
                                    uint32_t shrink_max_dim, float *centroids4)
{
    /* operations.rs:15-88 */
    const uint8_t *src = rgba; uint8_t *small = NULL;
    uint32_t sw = w, sh = h;
    if (shrink_max_dim && (w > shrink_max_dim || h > shrink_max_dim)) {  /* structures.rs:67-74 */
        orc_resized_dims(w, h, shrink_max_dim, &sw, &sh);
        small = (uint8_t *)malloc((uint64_t)sw * sh * 4);
        orc_resize(rgba, w, h, sw, sh, small);
        src = small;
    }
    uint64_t n = (uint64_t)sw * sh;
    float *lab = (float *)malloc(sizeof(float) * 3 * n);
    uint32_t *labels = (uint32_t *)malloc(sizeof(uint32_t) * n);
    orc_rgb_to_lab(src, n, lab);                                /* operations.rs:63-71 */
    for (uint32_t j = 0; j < 4 * k; ++j) centroids4[j] = 0.0f;  /* structures.rs:501-521 */
    orc_init_centroids(lab, sw, sh, k, centroids4);            /* operations.rs:73 */
    /* modules.rs:764-766, lib.rs:189-194 */
    uint32_t it = orc_lloyd(lab, n, k, centroids4, labels, 128, 8, 1.0f);
    free(labels); free(lab); free(small);
    return it;
}

void orc_apply(const uint8_t *rgba, uint32_t w, uint32_t h, const float *centroids4,
               uint32_t k, int mode, uint8_t *out_rgba)
{
    uint64_t n = (uint64_t)w * h;
    float *lab = (float *)malloc(sizeof(float) * 3 * n);
    orc_rgb_to_lab(rgba, n, lab);
    if (mode == ORC_MODE_MELD) {                                /* operations.rs:157-213 */
        float *out_lab = (float *)malloc(sizeof(float) * 3 * n);
        orc_meld(lab, w, h, centroids4, k, out_lab);
        orc_lab_to_rgba8(out_lab, n, out_rgba);
        free(out_lab); free(lab);
        return;
    }
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * n);
    if (mode == ORC_MODE_DITHER) orc_dither(lab, w, h, centroids4, k, idx);  /* operations.rs:99-155 */
    else orc_assign(lab, n, centroids4, k, 0, idx);            /* operations.rs:215-271 */
    /* swap.wgsl:12-25 + lab_to_rgb: only k (+ sentinel) distinct outputs exist */
    uint8_t *pal = (uint8_t *)malloc(4 * (k + 1));
    for (uint32_t j = 0; j < k; ++j) lab_to_rgba8_one(centroids4 + 4 * j, pal + 4 * j);
    const float sentinel[3] = {10000.0f, 10000.0f, 10000.0f};
    lab_to_rgba8_one(sentinel, pal + 4 * k);
    for (uint64_t i = 0; i < n; ++i) memcpy(out_rgba + 4 * i, pal + 4 * idx[i], 4);
    free(pal); free(idx); free(lab);
}

void orc_find(const uint8_t *rgba, uint32_t w, uint32_t h, const uint8_t *palette_rgba,
              uint32_t n_colors, int mode, uint8_t *out_rgba)
{
    /* lib.rs:79-114, structures.rs:523-553 */
    float *c4 = (float *)malloc(sizeof(float) * 4 * n_colors);
    for (uint32_t j = 0; j < n_colors; ++j) {
        orc_palette_srgb8_to_lab(palette_rgba + 4 * j, c4 + 4 * j);
        c4[4 * j + 3] = 1.0f;
    }
    orc_apply(rgba, w, h, c4, n_colors, mode, out_rgba);
    free(c4);
}

void orc_reduce(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, int mode,
                uint8_t *out_rgba)
{
    /* lib.rs:116-164 with Algorithm::Kmeans; structures.rs:23 MAX_IMAGE_DIMENSION = 256 */
    float *c4 = (float *)malloc(sizeof(float) * 4 * k);
    orc_extract_palette_kmeans(rgba, w, h, k, 256, c4);
    orc_apply(rgba, w, h, c4, k, mode, out_rgba);
    free(c4);
}

void orc_palette(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, uint8_t *out_rgba)
{
    /* lib.rs:255-286 kmeans_palette */
    float *c4 = (float *)malloc(sizeof(float) * 4 * k);
    orc_extract_palette_kmeans(rgba, w, h, k, 256, c4);
    float *keyL = (float *)malloc(sizeof(float) * k);
    for (uint32_t j = 0; j < k; ++j) {
        uint8_t rgb[3]; float lab[3];
        orc_palette_lab_to_srgb8(c4 + 4 * j, rgb);             /* structures.rs:601-607 */
        out_rgba[4 * j] = rgb[0]; out_rgba[4 * j + 1] = rgb[1];
        out_rgba[4 * j + 2] = rgb[2]; out_rgba[4 * j + 3] = 255;
        orc_palette_srgb8_to_lab(rgb, lab);                    /* lib.rs:276-284 sort key */
        keyL[j] = lab[0];
    }
    /* sort ascending by L (insertion sort = stable; the reference's sort_unstable_by may
     * order equal-L colours arbitrarily)                                                 */
    for (uint32_t i = 1; i < k; ++i) {
        float kl = keyL[i]; uint8_t px[4]; memcpy(px, out_rgba + 4 * i, 4);
        int64_t j = (int64_t)i - 1;
        while (j >= 0 && keyL[j] > kl) {
            keyL[j + 1] = keyL[j]; memcpy(out_rgba + 4 * (j + 1), out_rgba + 4 * j, 4); --j;
        }
        keyL[j + 1] = kl; memcpy(out_rgba + 4 * (j + 1), px, 4);
    }
    free(keyL); free(c4);
}

/* ------------------------------------------------------------------------------------ */
/* Algorithm::Octree         core/src/octree.rs, core/src/lib.rs:288-331                 */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    uint32_t level; int32_t parent; uint32_t color_index; int32_t children[8]; uint32_t child_count;
    uint64_t count, r, g, b; int in_leaves;
} orc_node;

/* Node::partial_cmp (octree.rs:214-233): returns <0, 0, >0 */
static int node_cmp(const orc_node *nodes, int32_t a, int32_t b)
{
    if (a == b) return 0;
    if (nodes[a].child_count != nodes[b].child_count) return nodes[a].child_count < nodes[b].child_count ? -1 : 1;
    uint64_t ac = nodes[a].count >> nodes[a].level, bc = nodes[b].count >> nodes[b].level;
    if (ac != bc) return ac < bc ? -1 : 1;
    return a < b ? -1 : 1;
}

uint32_t orc_octree_palette(const uint8_t *rgba, uint64_t n, uint32_t color_count, uint8_t *out_rgba)
{
    if (color_count == 0) return 0;                                   /* octree.rs:67-69 */
    uint64_t cap = 1 + 8 * n;
    orc_node *nodes = (orc_node *)calloc(cap, sizeof(orc_node));
    uint64_t n_nodes = 1;
    nodes[0].parent = -1;
    for (int i = 0; i < 8; ++i) nodes[0].children[i] = -1;
    for (uint64_t i = 0; i < n; ++i) {                                /* add_color, :41-64 */
        const uint8_t *px = rgba + 4 * i;
        int32_t cur = 0;
        for (uint32_t level = 0; level < 8; ++level) {
            uint8_t mask = (uint8_t)(0x80u >> level);
            uint32_t ci = 0;
            if (px[0] & mask) ci |= 4u;
            if (px[1] & mask) ci |= 2u;
            if (px[2] & mask) ci |= 1u;
            if (nodes[cur].children[ci] < 0) {
                orc_node *nd = &nodes[n_nodes];
                nd->level = level; nd->parent = cur; nd->color_index = ci;
                for (int q = 0; q < 8; ++q) nd->children[q] = -1;
                nodes[cur].children[ci] = (int32_t)n_nodes;
                nodes[cur].child_count += 1;
                n_nodes += 1;
            }
            cur = nodes[cur].children[ci];
        }
        nodes[cur].r += px[0]; nodes[cur].g += px[1]; nodes[cur].b += px[2]; nodes[cur].count += 1;
    }
    uint64_t n_leaves = 0;
    for (uint64_t i = 0; i < n_nodes; ++i)
        if (nodes[i].count > 0) { nodes[i].in_leaves = 1; n_leaves += 1; }   /* :71-78 */
    while (n_leaves > color_count) {                                  /* :80-103: smallest leaf merges into its parent */
        int32_t m = -1;
        for (uint64_t i = 0; i < n_nodes; ++i)
            if (nodes[i].in_leaves && (m < 0 || node_cmp(nodes, (int32_t)i, m) < 0)) m = (int32_t)i;
        nodes[m].in_leaves = 0; n_leaves -= 1;
        int32_t pid = nodes[m].parent;
        if (pid >= 0) {
            if (nodes[pid].in_leaves) { nodes[pid].in_leaves = 0; n_leaves -= 1; }
            nodes[pid].r += nodes[m].r; nodes[pid].g += nodes[m].g; nodes[pid].b += nodes[m].b;
            nodes[pid].count += nodes[m].count;
            nodes[pid].child_count -= 1;
            nodes[pid].children[nodes[m].color_index] = -1;
            nodes[m].parent = -1;
            nodes[pid].in_leaves = 1; n_leaves += 1;
        }
    }
    uint32_t cnt = 0;
    for (uint64_t i = 0; i < n_nodes; ++i)
        if (nodes[i].in_leaves) {                                     /* :106-109 output_color */
            out_rgba[4 * cnt + 0] = (uint8_t)(nodes[i].r / nodes[i].count);
            out_rgba[4 * cnt + 1] = (uint8_t)(nodes[i].g / nodes[i].count);
            out_rgba[4 * cnt + 2] = (uint8_t)(nodes[i].b / nodes[i].count);
            out_rgba[4 * cnt + 3] = 255;
            cnt += 1;
        }
    free(nodes);
    /* :110-111 sort (RGBA8 derives Ord: r, g, b, a) and dedup */
    for (uint32_t i = 1; i < cnt; ++i) {
        uint8_t px[4]; memcpy(px, out_rgba + 4 * i, 4);
        int64_t j = (int64_t)i - 1;
        while (j >= 0 && memcmp(out_rgba + 4 * j, px, 4) > 0) { memcpy(out_rgba + 4 * (j + 1), out_rgba + 4 * j, 4); --j; }
        memcpy(out_rgba + 4 * (j + 1), px, 4);
    }
    uint32_t u = 0;
    for (uint32_t i = 0; i < cnt; ++i)
        if (u == 0 || memcmp(out_rgba + 4 * (u - 1), out_rgba + 4 * i, 4) != 0) { memmove(out_rgba + 4 * u, out_rgba + 4 * i, 4); u += 1; }
    return u;
}

/* octree_palette (lib.rs:288-331): shrink to <= 128, octree, sort ascending by palette-crate Lab L */
uint32_t orc_palette_octree(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, uint8_t *out_rgba)
{
    const uint8_t *src = rgba; uint8_t *small = NULL;
    uint32_t sw = w, sh = h;
    if (w > 128 || h > 128) {                                         /* lib.rs:293-308 */
        orc_resized_dims(w, h, 128, &sw, &sh);
        small = (uint8_t *)malloc((uint64_t)sw * sh * 4);
        orc_resize(rgba, w, h, sw, sh, small);
        src = small;
    }
    uint32_t cnt = orc_octree_palette(src, (uint64_t)sw * sh, k, out_rgba);
    free(small);
    float *keyL = (float *)malloc(sizeof(float) * (cnt ? cnt : 1));
    for (uint32_t j = 0; j < cnt; ++j) { float lab[3]; orc_palette_srgb8_to_lab(out_rgba + 4 * j, lab); keyL[j] = lab[0]; }
    for (uint32_t i = 1; i < cnt; ++i) {                              /* lib.rs:320-328 (stable here) */
        float kl = keyL[i]; uint8_t px[4]; memcpy(px, out_rgba + 4 * i, 4);
        int64_t j = (int64_t)i - 1;
        while (j >= 0 && keyL[j] > kl) { keyL[j + 1] = keyL[j]; memcpy(out_rgba + 4 * (j + 1), out_rgba + 4 * j, 4); --j; }
        keyL[j + 1] = kl; memcpy(out_rgba + 4 * (j + 1), px, 4);
    }
    free(keyL);
    return cnt;
}

/* ImageProcessor::reduce with Algorithm::Octree (lib.rs:133-136) */
void orc_reduce_octree(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t k, int mode, uint8_t *out_rgba)
{
    uint8_t *pal = (uint8_t *)malloc(4 * (size_t)(k ? k : 1));
    uint32_t cnt = orc_palette_octree(rgba, w, h, k, pal);
    orc_find(rgba, w, h, pal, cnt, mode, out_rgba);
    free(pal);
}

/* ------------------------------------------------------------------------------------ */
/* synthetic inputs and the CPU-baseline kernel                                          */
/* ------------------------------------------------------------------------------------ */

void orc_synth_uniform(uint64_t seed, uint64_t n, uint8_t *rgba)
{
    uint64_t s = seed;
    for (uint64_t i = 0; i < n; ++i) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        rgba[4 * i + 0] = (uint8_t)(z & 255);
        rgba[4 * i + 1] = (uint8_t)((z >> 8) & 255);
        rgba[4 * i + 2] = (uint8_t)((z >> 16) & 255);
        rgba[4 * i + 3] = 255;
    }
}

void orc_assign_accumulate_rgba(const uint8_t *rgba, uint64_t n, const float *centroids4,
                                uint32_t k, uint32_t *labels, int64_t *acc4)
{
    const float *lut = srgb_lut();
    float *c5 = make_cent5(centroids4, k);
    memset(acc4, 0, sizeof(int64_t) * 4 * k);
#pragma omp parallel
    {
        int64_t *loc = (int64_t *)calloc(4 * (size_t)k, sizeof(int64_t));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            float lab[3];
            pixel_to_lab(lut, rgba + 4 * i, lab);
            orc_px p = px_terms(lab);
            uint32_t c = argmin_lit(&p, c5, k);
            labels[i] = c;
            loc[4 * c + 0] += fix(lab[0]);
            loc[4 * c + 1] += fix(lab[1]);
            loc[4 * c + 2] += fix(lab[2]);
            loc[4 * c + 3] += 1;
        }
#pragma omp critical(orc_acc)
        for (uint32_t j = 0; j < 4 * k; ++j) acc4[j] += loc[j];
        free(loc);
    }
    free(c5);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
