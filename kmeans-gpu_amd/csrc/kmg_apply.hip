// kmg_apply.hip -- the output passes of the C ABI (find_colors / dither_colors / meld_colors, core/src/operations.rs:99-155,
// 215-271): kmg_apply_plan_*, kmg_dev_apply, their cost models and the exhaustive checks of their candidate masks / lists.

#include "kmg_state.h"

// Same model for the replace-mode output pass (no histogram: every cell is labelled; scratch from the processor's
// blocks).
static bool replace_table_pays(const kmg_processor *p, uint64_t n, uint32_t k)
{
    if (const int f = forced_strategy(p)) return f > 0;
    const double N = (double)n;
    const double brute = N * (8.1e-12 + 2.45e-13 * k);
    const double table = 9.0e-5 + 1.85e-7 * k + N * (k <= 256 ? 1.8e-12 + 2.0e-15 * k : 6.6e-12);
    return table < brute;
}

// Dither output pass: per-pixel scan of all k centroids, or of the candidates of the pixel's cell only (k <= 256: byte lists per
// cell of a grid over Lab, kmg_lists.hip; above: mask words per (RGB cell, Bayer index)).  Measured on MI355X, noise images,
// random palettes (tools/dither_crossover.py -> profiles/r03_dither_crossover.txt): the scan costs 6.5 + 0.275 k ps per pixel,
// the list pass 6.8 + 0.02 k ps per pixel after ~45 us for the lists and their launch; with k >= 128 the scan's own latency
// (one wave walks all k) makes the lists win on any image.
static bool dither_pruning_pays(const kmg_processor *p, uint64_t n, uint32_t k)
{
    if (const int f = forced_strategy(p)) return f > 0;
    if (k > kLabListMaxK) return n >= 4000000ull;          // mask words: 0.55 ms of masks at k = 512
    if (k >= 128u) return n >= 16384ull;
    return (double)n * (0.255 * k - 0.3) > 45.0e6;          // ps saved per pixel x pixels > 45 us
}

// Meld output pass: ordered scan of all k centroids per pixel, or of the candidates of the pixel's colour
// cell only (k_meld_candidates: ~0.03 ms per 64 centroids; tools/dither_probe.py).
static bool meld_pruning_pays(const kmg_processor *p, uint64_t n, uint32_t k)
{
    if (const int f = forced_strategy(p)) return f > 0;
    return k >= 16 && n >= (1ull << 20);
}

// test support: exhaustive validation of the dither candidate masks (kmg_table.hip) for a centroid
// table: over all 2^24 colours x 16 Bayer offsets, the arg-min over the candidates must equal the
// brute-force arg-min.  *violations must come back 0.
// Which pruned dither / meld pass?  k <= 512: byte lists per cell of a grid over Lab (kmg_lists.hip); larger k: mask words per (RGB
// cell, Bayer index) (kmg_table.hip).  KMG_STRATEGY_MASK_WORDS (kmg_options.strategy) sends every k to the mask words.
static bool dither_takes_lists(const kmg_processor *p, uint32_t k)
{
    return !(p->strategy.load(std::memory_order_relaxed) & KMG_STRATEGY_MASK_WORDS) && k <= kLabListMaxK;
}

extern "C" int kmg_debug_check_dither_masks(kmg_processor *p, const float *c4, uint32_t k, uint64_t *violations, void *stream)
try {
    if (!p || !c4 || !violations || k < 2 || k > KMG_MAX_K) return fail(KMG_ERR_INVALID_ARGUMENT, "bad check_dither_masks arguments");
    HIP_TRY(hipSetDevice(p->device));
    int rc;
    if ((rc = ensure_bounds(p, S(stream))) != KMG_OK) return rc;
    std::vector<Centroid> hc(k);
    for (uint32_t i = 0; i < k; ++i) {
        hc[i].L = c4[4 * i]; hc[i].a = c4[4 * i + 1]; hc[i].b = c4[4 * i + 2];
        hc[i].C = chroma(hc[i].a, hc[i].b);
    }
    const float thr = dither_threshold(c4, k);
    DevBuf cent, masks, viol;
    HIP_TRY(cent.alloc(sizeof(Centroid) * k));
    HIP_TRY(masks.alloc(sizeof(uint64_t) * (size_t)kCells * 16u * mask_words(k)));
    HIP_TRY(viol.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyAsync(cent.ptr, hc.data(), sizeof(Centroid) * k, hipMemcpyHostToDevice, S(stream)));
    HIP_TRY(hipMemsetAsync(viol.ptr, 0, sizeof(unsigned long long), S(stream)));
    HIP_TRY(launch_offset_candidates(p->d_bounds, (const Centroid *)cent.ptr, k, thr, (uint64_t *)masks.ptr, S(stream)));
    HIP_TRY(launch_check_offset_masks((const Centroid *)cent.ptr, k, (const uint64_t *)masks.ptr, p->d_lut, thr,
                                      (unsigned long long *)viol.ptr, S(stream)));
    if (k <= kLabListMaxK) {                                // the byte lists over Lab cells (kmg_lists.hip), same counter
        DevBuf lists;
        HIP_TRY(lists.alloc(lab_list_bytes(k)));
        HIP_TRY(launch_lab_candidates((const Centroid *)cent.ptr, k, thr, false, (uint8_t *)lists.ptr, S(stream)));
        HIP_TRY(launch_check_lab_lists((const Centroid *)cent.ptr, k, (const uint8_t *)lists.ptr, p->d_lut, thr,
                                       (unsigned long long *)viol.ptr, S(stream)));
        HIP_TRY(hipStreamSynchronize(S(stream)));
    }
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, viol.ptr, sizeof h, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    *violations = h;
    if (KMG_TOOLS_ENV("KMG_DITHER_STATS")) {
        // distribution of the candidate counts: per (cell, Bayer index) slot, and per cell over its 16 slots together
        const uint32_t words = mask_words(k);
        std::vector<uint64_t> hm((size_t)kCells * 16u * words);
        HIP_TRY(hipMemcpy(hm.data(), masks.ptr, hm.size() * 8, hipMemcpyDeviceToHost));
        uint64_t hs[34] = {0}, hu[34] = {0};
        for (uint32_t c = 0; c < kCells; ++c) {
            std::vector<uint64_t> u(words, 0);
            for (uint32_t b = 0; b < 16; ++b) {
                uint32_t n = 0;
                for (uint32_t w = 0; w < words; ++w) { const uint64_t m = hm[((size_t)c * 16 + b) * words + w]; u[w] |= m; n += (uint32_t)__builtin_popcountll(m); }
                hs[n < 33 ? n : 33]++;
            }
            uint32_t n = 0;
            for (uint32_t w = 0; w < words; ++w) n += (uint32_t)__builtin_popcountll(u[w]);
            hu[n < 33 ? n : 33]++;
        }
        fprintf(stderr, "dither candidates per slot :");
        for (int i = 0; i < 34; ++i) fprintf(stderr, " %.4f", (double)hs[i] / (kCells * 16.0));
        fprintf(stderr, "\ndither candidates per cell (union of 16 slots):");
        for (int i = 0; i < 34; ++i) fprintf(stderr, " %.4f", (double)hu[i] / kCells);
        fprintf(stderr, "\n");
    }
    return KMG_OK;
}
KMG_ABI_CATCH

#ifdef KMG_TOOLS
namespace kmg {
hipError_t launch_dither_list_stats(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0, const float *lut, float threshold,
                                    const uint8_t *lists, unsigned long long *out68, hipStream_t st);
}
// tools build: list lengths of the dither pass over an image (k <= 256): out[68] as k_dither_list_stats leaves them
extern "C" KMG_API int kmg_tools_dither_list_stats(kmg_processor *p, const uint8_t *d_rgba, uint32_t w, uint32_t rows, const float *c4, uint32_t k,
                                                  unsigned long long *out68)
try {
    if (!p || !d_rgba || !c4 || !out68 || k < 2 || k > 256) return fail(KMG_ERR_INVALID_ARGUMENT, "bad dither_list_stats arguments");
    HIP_TRY(hipSetDevice(p->device));
    std::vector<Centroid> hc(k);
    for (uint32_t i = 0; i < k; ++i) hc[i] = {c4[4 * i], c4[4 * i + 1], c4[4 * i + 2], chroma(c4[4 * i + 1], c4[4 * i + 2])};
    const float thr = dither_threshold(c4, k);
    DevBuf cent, lists, acc;
    HIP_TRY(cent.alloc(sizeof(Centroid) * k));
    HIP_TRY(lists.alloc(lab_list_bytes(k)));
    HIP_TRY(acc.alloc(sizeof(unsigned long long) * 68));
    HIP_TRY(hipMemcpy(cent.ptr, hc.data(), sizeof(Centroid) * k, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(acc.ptr, 0, sizeof(unsigned long long) * 68));
    HIP_TRY(launch_lab_candidates((const Centroid *)cent.ptr, k, thr, false, (uint8_t *)lists.ptr, nullptr));
    HIP_TRY(launch_dither_list_stats((const uint32_t *)d_rgba, w, rows, 0u, p->d_lut, thr, (const uint8_t *)lists.ptr, (unsigned long long *)acc.ptr, nullptr));
    HIP_TRY(hipMemcpy(out68, acc.ptr, sizeof(unsigned long long) * 68, hipMemcpyDeviceToHost));
    return KMG_OK;
}
KMG_ABI_CATCH
#endif

// test support: exhaustive validation of the meld candidate masks for a centroid table (k >= 2): over all
// 2^24 colours the two closest centroids found among the cell's candidates must be those of the full scan.
extern "C" int kmg_debug_check_meld_masks(kmg_processor *p, const float *c4, uint32_t k, uint64_t *violations, void *stream)
try {
    if (!p || !c4 || !violations || k < 2 || k > KMG_MAX_K) return fail(KMG_ERR_INVALID_ARGUMENT, "bad check_meld_masks arguments");
    HIP_TRY(hipSetDevice(p->device));
    int rc;
    if ((rc = ensure_bounds(p, S(stream))) != KMG_OK) return rc;
    std::vector<Centroid> hc(k);
    for (uint32_t i = 0; i < k; ++i) {
        hc[i].L = c4[4 * i]; hc[i].a = c4[4 * i + 1]; hc[i].b = c4[4 * i + 2];
        hc[i].C = chroma(hc[i].a, hc[i].b);
    }
    DevBuf cent, masks, viol;
    HIP_TRY(cent.alloc(sizeof(Centroid) * k));
    HIP_TRY(masks.alloc(sizeof(uint64_t) * (size_t)kCells * mask_words(k)));
    HIP_TRY(viol.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyAsync(cent.ptr, hc.data(), sizeof(Centroid) * k, hipMemcpyHostToDevice, S(stream)));
    HIP_TRY(hipMemsetAsync(viol.ptr, 0, sizeof(unsigned long long), S(stream)));
    HIP_TRY(launch_meld_candidates(p->d_bounds, (const Centroid *)cent.ptr, k, (uint64_t *)masks.ptr, S(stream)));
    HIP_TRY(launch_check_meld_masks((const Centroid *)cent.ptr, k, (const uint64_t *)masks.ptr, p->d_lut,
                                    (unsigned long long *)viol.ptr, S(stream)));
    DevBuf lists;
    if (k <= kLabListMaxK) {                                // the byte lists over Lab cells (kmg_lists.hip), same counter
        HIP_TRY(lists.alloc(lab_list_bytes(k)));
        HIP_TRY(launch_lab_candidates((const Centroid *)cent.ptr, k, 0.0f, true, (uint8_t *)lists.ptr, S(stream)));
        HIP_TRY(launch_check_lab_lists_two((const Centroid *)cent.ptr, k, (const uint8_t *)lists.ptr, p->d_lut,
                                           (unsigned long long *)viol.ptr, S(stream)));
    }
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, viol.ptr, sizeof h, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    *violations = h;
    return KMG_OK;
}
KMG_ABI_CATCH

// ---- the output pass as a PLAN: everything that depends on the centroid table only -- the device copy of centroids and palette,
// the threshold, the candidate lists / masks or the label tables of the colour cube -- is built once (asynchronously on the
// creating call's stream); kmg_apply_plan_run then launches the per-pixel kernel for any band of rows on any stream, without a
// host synchronisation, so a caller that streams an image in bands (or runs bands on several streams) overlaps copies and
// kernels.  kmg_dev_apply = create + run + synchronise + destroy.
struct kmg_apply_plan {
    kmg_processor *p = nullptr;
    uint32_t k = 0;
    int mode = 0;
    bool dither = false;
    float thr = 0.0f;
    enum Route { kScan, kMeldScan, kMeldMasks, kMeldLists, kReplaceTable, kDitherLists, kDitherMasks } route = kScan;
    ArenaGuard arena;
    std::vector<uint8_t> staged;        // host copy of the tables: lives as long as the asynchronous upload may
    Centroid *d_cent = nullptr;
    uint32_t *d_pal = nullptr;
    void *aux = nullptr;                // candidate lists / masks, or the per-colour labels (kReplaceTable)
    uint16_t *sub = nullptr;            // kReplaceTable: the label pass's first-level tables
    hipEvent_t ready = nullptr;         // the tables are built (recorded on the creating stream)
    hipStream_t built_on = nullptr;
};

extern "C" int kmg_apply_plan_create(kmg_processor *p, const float *c4, uint32_t k, int mode, uint64_t n_pixels_hint, void *stream,
                                     kmg_apply_plan **out)
try {
    if (!p || !c4 || !out || k == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad apply_plan arguments");
    *out = nullptr;
    if (k > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", k, KMG_MAX_K);
    if (mode != KMG_MODE_REPLACE && mode != KMG_MODE_DITHER && mode != KMG_MODE_MELD)
        return fail(KMG_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    HIP_TRY(hipSetDevice(p->device));
    kmg_apply_plan *pl = new (std::nothrow) kmg_apply_plan();
    if (!pl) return fail(KMG_ERR_OUT_OF_MEMORY, "host allocation failed");
    struct Undo { kmg_apply_plan *pl; ~Undo() { if (pl) { if (pl->ready) (void)hipEventDestroy(pl->ready); (void)hipStreamSynchronize(pl->built_on); delete pl; } } } undo{pl};
    pl->p = p; pl->k = k; pl->mode = mode; pl->built_on = S(stream);

    // per-centroid work on the host: (L,a,b,C) table, RGBA8 palette (lab_to_rgb.wgsl), threshold
    std::vector<Centroid> hc(k);
    std::vector<uint32_t> pal(k + 1);
    for (uint32_t i = 0; i < k; ++i) {
        hc[i].L = c4[4 * i]; hc[i].a = c4[4 * i + 1]; hc[i].b = c4[4 * i + 2];
        hc[i].C = chroma(hc[i].a, hc[i].b);
        uint8_t px[4];
        shader_lab_to_rgba8(c4 + 4 * i, px);
        memcpy(&pal[i], px, 4);
    }
    {
        const float sentinel[3] = {10000.0f, 10000.0f, 10000.0f};     // mix_colors.wgsl:73
        uint8_t px[4];
        shader_lab_to_rgba8(sentinel, px);
        memcpy(&pal[k], px, 4);
    }
    const bool dither = (mode == KMG_MODE_DITHER) && k > 1;           // mix_colors.wgsl:104-108
    const float thr = dither ? dither_threshold(c4, k) : 0.0f;
    pl->dither = dither; pl->thr = thr;

    // which route (by the number of pixels the plan is made for), and how much scratch it needs: one block per plan
    const uint64_t n_px = n_pixels_hint;
    const bool meld_masks_pay = mode == KMG_MODE_MELD && k >= 2 && meld_pruning_pays(p, n_px, k);
    const bool replace_table = mode != KMG_MODE_MELD && !dither && replace_table_pays(p, n_px, k);
    const bool dither_pruned = mode != KMG_MODE_MELD && dither && dither_pruning_pays(p, n_px, k);
    const size_t tables_bytes = sizeof(Centroid) * k + sizeof(uint32_t) * (k + 1);
    const size_t sub_bytes = sizeof(uint16_t) * (kSubCells + kCells) + sizeof(uint32_t) * kCells;
    const size_t labels_bytes = (size_t)(k <= 256 ? 1 : 2) << 24;
    const size_t masks_bytes = sizeof(uint64_t) * (size_t)kCells * mask_words(k) * (dither_pruned ? 16u : 1u);
    size_t need = ArenaGuard::padded(tables_bytes);
    const bool dither_lists = dither_pruned && dither_takes_lists(p, k);      // byte lists over Lab cells instead of mask words
    const bool meld_lists = meld_masks_pay && dither_takes_lists(p, k);       // the same for the meld pass's two closest
    if ((meld_masks_pay && !meld_lists) || (dither_pruned && !dither_lists)) need += ArenaGuard::padded(masks_bytes);
    if (dither_lists || meld_lists) need += ArenaGuard::padded(lab_list_bytes(k));
    if (replace_table) need += ArenaGuard::padded(labels_bytes) + ArenaGuard::padded(sub_bytes) + ArenaGuard::padded(cube_masks_bytes(k)) +
                               ArenaGuard::padded(cube_work_bytes());
    ArenaGuard &arena = pl->arena;
    hipError_t e = arena.acquire(p, need);
    // centroid table and palette travel in one block (one copy)
    pl->staged.resize(tables_bytes);
    memcpy(pl->staged.data(), hc.data(), sizeof(Centroid) * k);
    memcpy(pl->staged.data() + sizeof(Centroid) * k, pal.data(), sizeof(uint32_t) * (k + 1));
    Centroid *d_cent = nullptr;
    if (e == hipSuccess) {
        d_cent = (Centroid *)arena.take(tables_bytes);
        pl->d_cent = d_cent;
        pl->d_pal = (uint32_t *)((uint8_t *)d_cent + sizeof(Centroid) * k);
        e = hipMemcpyAsync(d_cent, pl->staged.data(), pl->staged.size(), hipMemcpyHostToDevice, S(stream));
    }
    int rc = KMG_OK;
    if (e != hipSuccess) {
        // fall through to the error report
    } else if (mode == KMG_MODE_MELD) {
        pl->route = kmg_apply_plan::kMeldScan;
        if (meld_lists) {
            uint8_t *lst = (uint8_t *)arena.take(lab_list_bytes(k));
            e = launch_lab_candidates(d_cent, k, 0.0f, true, lst, S(stream));
            pl->aux = lst; pl->route = kmg_apply_plan::kMeldLists;
        } else if (meld_masks_pay) {
            // large image: per colour cell, the centroids that can be one of a pixel's two closest
            if ((rc = ensure_bounds(p, S(stream))) == KMG_OK) {
                uint64_t *m = (uint64_t *)arena.take(masks_bytes);
                e = launch_meld_candidates(p->d_bounds, d_cent, k, m, S(stream));
                pl->aux = m; pl->route = kmg_apply_plan::kMeldMasks;
            }
        }
    } else if (replace_table) {
        // replace mode on a large image: the label of a pixel depends on its colour only, so label the
        // colour cube once (candidate masks + cube pass without sums) and emit pal[label] through the
        // label tables -- the same bit-exact machinery as the Lloyd label pass
        if ((rc = ensure_bounds(p, S(stream))) == KMG_OK) {
            void *colour_labels = arena.take(labels_bytes);
            uint16_t *sub = (uint16_t *)arena.take(sub_bytes);
            uint64_t *m = (uint64_t *)arena.take(cube_masks_bytes(k));
            void *cwork = arena.take(cube_work_bytes());
            e = launch_cube(nullptr, nullptr, nullptr, nullptr, nullptr, p->d_bounds, p->d_sub_bounds, d_cent, k, p->d_lab_table,
                            m, cwork, colour_labels, sub, nullptr, 0, 0u, nullptr, S(stream), nullptr, affine_for(p, k, S(stream)));
            pl->aux = colour_labels; pl->sub = sub; pl->route = kmg_apply_plan::kReplaceTable;
        }
    } else if (dither_pruned) {
        // dither on a large image: candidate lists per cell of a grid over Lab (mask words per (colour cell, Bayer index)
        // above k = 512), then a scan of the pixel's candidates only
        if (dither_lists) {
            uint8_t *lst = (uint8_t *)arena.take(lab_list_bytes(k));
            e = launch_lab_candidates(d_cent, k, thr, false, lst, S(stream));
            pl->aux = lst; pl->route = kmg_apply_plan::kDitherLists;
        } else if ((rc = ensure_bounds(p, S(stream))) == KMG_OK) {
            uint64_t *m = (uint64_t *)arena.take(masks_bytes);
            e = launch_offset_candidates(p->d_bounds, d_cent, k, thr, m, S(stream));
            pl->aux = m; pl->route = kmg_apply_plan::kDitherMasks;
        }
    }
    if (rc != KMG_OK) return rc;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(pl->ready, S(stream));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP, "apply plan failed: %s", hipGetErrorString(e));
    undo.pl = nullptr;
    *out = pl;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_apply_plan_run(kmg_apply_plan *pl, const uint8_t *d_rgba, uint32_t w, uint32_t rows, uint32_t row0, uint8_t *d_out,
                                  void *stream)
try {
    if (!pl || !d_rgba || !d_out || !w || !rows) return fail(KMG_ERR_INVALID_ARGUMENT, "bad apply_plan_run arguments");
    // the output kernels keep the pixel index (and from it the Bayer coordinates) in 32 bits
    if ((uint64_t)w * rows > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "band has more than 2^32-1 pixels");
    kmg_processor *p = pl->p;
    HIP_TRY(hipSetDevice(p->device));
    if (S(stream) != pl->built_on) HIP_TRY(hipStreamWaitEvent(S(stream), pl->ready, 0));      // (another stream: after the tables)
    const uint64_t n_px = (uint64_t)w * rows;
    const uint32_t k = pl->k;
    hipError_t e = hipSuccess;
    switch (pl->route) {
    case kmg_apply_plan::kMeldLists:
        e = launch_meld_lists((const uint32_t *)d_rgba, n_px, pl->d_cent, k, p->d_lut, (const uint8_t *)pl->aux, (uint32_t *)d_out, S(stream));
        break;
    case kmg_apply_plan::kMeldMasks:
    case kmg_apply_plan::kMeldScan:
        e = launch_meld((const uint32_t *)d_rgba, n_px, pl->d_cent, k, p->d_lut, (const uint64_t *)pl->aux, (uint32_t *)d_out, S(stream));
        break;
    case kmg_apply_plan::kReplaceTable:
        e = launch_labels((const uint32_t *)d_rgba, n_px, pl->aux, pl->sub, k, pl->d_pal, (uint32_t *)d_out, S(stream));
        break;
    case kmg_apply_plan::kDitherLists:
        e = launch_dither_lists((const uint32_t *)d_rgba, w, rows, row0, pl->d_cent, k, p->d_lut, pl->d_pal, pl->thr, (const uint8_t *)pl->aux,
                                (uint32_t *)d_out, S(stream));
        break;
    case kmg_apply_plan::kDitherMasks:
        e = launch_dither_pruned((const uint32_t *)d_rgba, w, rows, row0, pl->d_cent, k, p->d_lut, pl->d_pal, pl->thr, (const uint64_t *)pl->aux,
                                 (uint32_t *)d_out, S(stream));
        break;
    default:
        e = launch_apply((const uint32_t *)d_rgba, w, rows, row0, pl->d_cent, k, p->d_lut, pl->d_pal, pl->dither, pl->thr, (uint32_t *)d_out,
                         S(stream));
    }
    if (e != hipSuccess) return fail(KMG_ERR_HIP, "apply failed: %s", hipGetErrorString(e));
    return KMG_OK;
}
KMG_ABI_CATCH

// The plan's scratch block goes back to the processor: every stream that ran the plan must have been synchronised by the caller,
// or `synchronise` != 0 makes this call wait for the whole device first.
extern "C" void kmg_apply_plan_destroy(kmg_apply_plan *pl, int synchronise)
try {
    if (!pl) return;
    (void)hipSetDevice(pl->p->device);
    if (synchronise) (void)hipDeviceSynchronize();
    if (pl->ready) (void)hipEventDestroy(pl->ready);
    delete pl;                                                        // (~ArenaGuard returns the block)
}
KMG_ABI_CATCH_VOID

extern "C" int kmg_dev_apply(kmg_processor *p, const uint8_t *d_rgba, uint32_t w, uint32_t rows, uint32_t row0,
                             const float *c4, uint32_t k, int mode, uint8_t *d_out, void *stream)
try {
    if (!p || !d_rgba || !d_out || !c4 || !w || !rows || k == 0)
        return fail(KMG_ERR_INVALID_ARGUMENT, "bad apply arguments");
    if ((uint64_t)w * rows > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "band has more than 2^32-1 pixels");
    kmg_apply_plan *pl = nullptr;
    int rc = kmg_apply_plan_create(p, c4, k, mode, (uint64_t)w * rows, stream, &pl);
    if (rc != KMG_OK) return rc;
    rc = kmg_apply_plan_run(pl, d_rgba, w, rows, row0, d_out, stream);
    const hipError_t e2 = hipStreamSynchronize(S(stream));            // the call returns when the band is written; the block is idle again
    kmg_apply_plan_destroy(pl, 0);
    if (rc != KMG_OK) return rc;
    if (e2 != hipSuccess) return fail(KMG_ERR_HIP, "apply failed: %s", hipGetErrorString(e2));
    return KMG_OK;
}
KMG_ABI_CATCH

