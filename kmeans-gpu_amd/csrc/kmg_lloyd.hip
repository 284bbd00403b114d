// kmg_lloyd.hip -- one Lloyd problem of the C ABI (kmg_lloyd_*, include/kmeans_hip.h): binding an image to its colour table
// (kmg_table.h), the strategy choice, the farthest-point initialisation (modules.rs:946-1246), the assign / accumulate / update
// passes and the loop of ChooseCentroidModule::compute (modules.rs:763-840), with their test and statistics support.  Every
// per-pixel and per-colour step is a kernel launch.

#include <math.h>

#include "kmg_state.h"

// ---------------------------------------------------------------------------------------------
// colour-table strategy (kmg_table.h): binding an image, strategy choice
// ---------------------------------------------------------------------------------------------
static void drop_events(kmg_lloyd *s);
static void destroy_events(kmg_lloyd *s);
static int debug_refresh(kmg_lloyd *s, hipStream_t st, unsigned long long stage[8]);
static int side_flush(kmg_lloyd *s, hipStream_t st);
// The table's blocks go back to the processor for the next image.  The caller has made sure that no kernel still uses them.
static void free_table(kmg_processor *p, ColourTable &t)
{
    block_give(p, t.blk, t.blk_cap);
    block_give(p, t.blk_alt, t.blk_alt_cap);
    block_give(p, t.blk_init, t.blk_init_cap);
    t = ColourTable();
}


// carve `bytes` (256-byte granules) off a block
static inline void *carve(void *base, size_t &off, size_t bytes)
{
    void *r = (uint8_t *)base + off;
    off += pad256(bytes);
    return r;
}

// Cost model of one Lloyd iteration (seconds on MI355X), refitted in round 6 to tools/costmodel_sweep.py
// (profiles/r06_costmodel_sweep.txt: noise, Gaussian blobs and the tiled photograph, 2^18 .. 2^24 pixels, k = 8 .. 256, the
// one-launch cube pass).  It is asked TWICE: before anything is known about the image (`facts` = NULL) it answers with the
// cheapest cube pass an image of this size can have -- "is a binding worth trying" -- and after the binding, whose histogram
// says how many of the 32^3 cells the image occupies and whether it has hot cells, it answers for THIS image (the binding
// is paid by then: it no longer counts).  Rounds 2-5 asked once, blind, and leaned towards the photograph: a 1 Mpx noise
// image at k = 256 got the slower strategy by 37 %.
//   per-pixel scan : 12 us (k < 16: 18) + 0.04 k us + n (6 + 0.36 k - 0.00052 k^2) ps   k <= 256 (above: 1.1e-5 + 9.3e-8 k + n (7.5 + 0.245 k) ps);
//                    x 1.3 for k >= 64 on an image with hot cells (crowded centroids: more near-tie repairs)
//   cube pass      : k <= 32 (one launch)            4.3 + 0.15 k + (22 + 0.35 k) occ / 32768 us, + 10 us with hot cells
//                    32 < k <= 256, no hot cells     48.5 + 0.07 k us (k_cube_one: noise and blobs alike, whatever they occupy)
//                    32 < k <= 256, hot cells        26 + 0.2 k us (three launches; 1.9 K .. 12 K occupied cells alike)
//                    k > 256                         86 + 0.23 k us
//   label pass     : k <= 256: 3 us + 4.4 ps per pixel up to 2^22 pixels (one 1024-thread workgroup per CU stages the 128 KiB pair
//                    table whatever the image), 2.2 ps per pixel beyond; u16 labels (k > 256): 6.7 ps per pixel
//   binding        : bind_seconds(n) / 16            one-off histogram + cell sums, spread over ~16 passes
struct ImageFacts { uint32_t occupied, hot; };

static double bind_seconds(uint64_t n)
{
    const double N = (double)n;
    return n >= (1ull << 21) ? 2.5e-4 + N * 4.1e-12 : 2.0e-4 + N * 4.0e-11;
}

static double scan_seconds(uint64_t n, uint32_t k, bool hot)
{
    const double N = (double)n, K = (double)k;
    const double t = k <= 256u ? (k < 16u ? 1.8e-5 : 1.2e-5) + 4.0e-8 * K + N * (6.0e-12 + 3.6e-13 * K - 5.2e-16 * K * K)
                               : 1.1e-5 + 9.3e-8 * K + N * (7.5e-12 + 2.45e-13 * K);
    return t * ((hot && k >= 64u) ? 1.3 : 1.0);
}

static double cube_seconds(uint32_t k, const ImageFacts *facts)
{
    const double K = (double)k;
    if (k > 256u) return 8.6e-5 + 2.3e-7 * K;
    if (k <= kCubeSmallMaxK) {
        // (unknown image: a sparse one -- a third of the cells)
        const double occ = facts ? (double)facts->occupied / (double)kCells : 0.33;
        return 4.3e-6 + 1.5e-7 * K + (2.2e-5 + 3.5e-7 * K) * occ + ((facts && facts->hot) ? 1.0e-5 : 0.0);
    }
    const double one_launch = 4.85e-5 + 7.0e-8 * K, three_launches = 2.6e-5 + 2.0e-7 * K;
    if (!facts) return one_launch < three_launches ? one_launch : three_launches;
    return facts->hot ? three_launches : one_launch;
}

// facts == NULL: before a binding (its cost counts); else: the image is bound (its cost is spent)
static bool table_pays(const kmg_processor *p, uint64_t n, uint32_t k, bool labels, const ImageFacts *facts)
{
    if (const int f = forced_strategy(p)) return f > 0;
    const double N = (double)n;
    const double N22 = (double)(1u << 22);
    const double label_pass = !labels ? 0.0 : (k > 256u ? N * 6.7e-12 : (N <= N22 ? 3.0e-6 + N * 4.4e-12 : 2.15e-5 + (N - N22) * 2.2e-12));
    const double table = cube_seconds(k, facts) + label_pass + (facts ? 0.0 : bind_seconds(n) / 16.0);
    // (blind: the scan at its dearest -- an image with hot cells -- against the table at its cheapest)
    const double scan = scan_seconds(n, k, facts ? facts->hot != 0u : true);
    return table < scan;
}

int ensure_bounds(kmg_processor *p, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(p->mu);
    if (p->d_bounds) return KMG_OK;
    CellBounds *b = nullptr, *sb = nullptr;
    float4 *lab = nullptr;
    HIP_TRY(hipMalloc((void **)&b, sizeof(CellBounds) * kCells));
    hipError_t e = hipMalloc((void **)&sb, sizeof(CellBounds) * kSubCells);
    if (e == hipSuccess) e = hipMalloc((void **)&lab, sizeof(float4) << 24);
    if (e == hipSuccess) e = launch_cell_bounds(p->d_lut, b, sb, lab, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(b);
        if (sb) (void)hipFree(sb);
        if (lab) (void)hipFree(lab);
        return fail(e == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP, "colour tables failed: %s", hipGetErrorString(e));
    }
    p->d_lab_table = lab;
    p->d_sub_bounds = sb;
    p->d_bounds = b;
    return KMG_OK;
}

// The dominance test's table (kmg_table.h), image independent, for passes with k <= 256: built once, after ensure_bounds.
// NULL when it cannot be had -- the pass is exact without it, only slower.
const float *affine_for(kmg_processor *p, uint32_t k, hipStream_t st)
{
    if (k > 256u) return nullptr;                                      // (k_cube_small and k_cube_one make the test)
    std::lock_guard<std::mutex> lock(p->mu);
    if (p->d_sub_affine || p->affine_failed || !p->d_lab_table) return p->d_sub_affine;
    float *a = nullptr;
    hipError_t e = hipMalloc((void **)&a, sub_affine_bytes());
    if (e == hipSuccess) e = launch_sub_affine(p->d_lab_table, p->d_sub_bounds, a, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipFree(a);
        p->affine_failed = true;
        return nullptr;
    }
    p->d_sub_affine = a;
    return a;
}

// everything a binding derives from the histogram: per-cell and per-sub-cell sums, occupancy bits, the list of occupied
// cells and the hot cells (n_pixels = the pixels the histogram counts)
static int tables_from_histogram(kmg_lloyd *s, uint64_t n_pixels, hipStream_t st)
{
    ColourTable &t = s->tab;
    HIP_TRY(launch_cell_aggregates(t.d_hist, s->p->d_lab_table, t.d_agg, t.d_sub_agg, t.d_occ, st));
    // dense list of the occupied cells and the hot cells (static for this image); the label pass has a kernel variant for
    // images with hot cells, so the host needs their number: 4 bytes back, the one synchronisation of a binding
    HIP_TRY(launch_work_list(t.d_agg, t.d_work, s->k <= 256 ? n_pixels : 0, st));
    HIP_TRY(hipMemcpyAsync(&t.n_hot, t.d_work + kCells + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&t.n_occ, t.d_work, sizeof(uint32_t), hipMemcpyDeviceToHost, st));      // (the cost model wants it)
    HIP_TRY(hipStreamSynchronize(st));
    if (const char *e = KMG_TOOLS_ENV("KMG_HOT_CELLS")) { if (e[0] == '0') t.n_hot = 0; }      // (tools build only)
    t.d_work_share = nullptr;
    t.balance_work = nullptr;                                        // a new work list: the cube pass deals its tasks afresh
    return KMG_OK;
}

// want_tie: also build the init tie keys (ColourTable::d_tie, allocated by the caller) for an image whose
// first pixel has the image-wide index first_index
static int bind_image_impl(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, void *stream, bool want_tie, uint64_t first_index)
{
    if (!s || !d_rgba || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad bind_image arguments");
    if (s->side) HIP_TRY(hipStreamSynchronize(s->side));             // label passes of the previous binding
    s->lab_pending[0] = s->lab_pending[1] = false;
    // the histogram counts, partition totals and prefix sums are u32 (kmg_table.hip)
    if (n > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image has more than 2^32-1 pixels");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc;
    if ((rc = ensure_bounds(s->p, S(stream))) != KMG_OK) return rc;
    ColourTable &t = s->tab;
    if (!t.d_hist) {
        // one block for all tables, from the processor's idle blocks when one fits (no hipMalloc on a warm processor)
        const size_t sizes[11] = {sizeof(uint32_t) << 24, sizeof(int64_t) * 4ull * kCells, sizeof(int64_t) * 4ull * kSubCells,
                                  (size_t)1 << 21, cube_work_bytes(), cube_masks_bytes(s->k),
                                  sizeof(uint32_t) * kWorkWords, (size_t)(s->k <= 256 ? 1 : 2) << 24, sub_table_bytes(),
                                  sizeof(uint32_t) * (kCells + 1), cube_balance_bytes()};
        size_t need = 0;
        for (size_t b : sizes) need += pad256(b);
        const hipError_t e = block_take(s->p, need, &t.blk, &t.blk_cap);
        if (e != hipSuccess) {
            t.blk = nullptr; t.blk_cap = 0;
            return fail(e == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP,
                        "colour table allocation failed: %s", hipGetErrorString(e));
        }
        size_t off = 0;
        t.d_hist = (uint32_t *)carve(t.blk, off, sizes[0]);
        t.d_agg = (int64_t *)carve(t.blk, off, sizes[1]);
        t.d_sub_agg = (int64_t *)carve(t.blk, off, sizes[2]);
        t.d_occ = (uint8_t *)carve(t.blk, off, sizes[3]);
        t.d_cell_work = carve(t.blk, off, sizes[4]);
        t.d_masks = (uint64_t *)carve(t.blk, off, sizes[5]);
        t.d_work = (uint32_t *)carve(t.blk, off, sizes[6]);
        t.d_colour_labels = carve(t.blk, off, sizes[7]);
        t.d_sub = (uint16_t *)carve(t.blk, off, sizes[8]);
        t.share_buf = (uint32_t *)carve(t.blk, off, sizes[9]);
        t.d_balance = (uint16_t *)carve(t.blk, off, sizes[10]);
    }
    t.d_work_share = nullptr;                                        // a new image: the whole work list again
    t.balance_work = nullptr;
    t.rgba = nullptr;
    t.tables_valid = false;
    t.tie_valid = false;
    t.bound_by_init = false;
    t.bound_by_caller = false;
    // entries of cells no pixel falls into are never read by the label pass; 0xFF.. = "empty"
    HIP_TRY(hipMemsetAsync(t.d_sub, 0xFF, sub_table_bytes(), S(stream)));
    if (t.d_sub_alt) HIP_TRY(hipMemsetAsync(t.d_sub_alt, 0xFF, sub_table_bytes(), S(stream)));
    if (n >= (1ull << 21)) {
        // partition + per-partition LDS histograms: no global atomic per pixel (kmg_table.hip)
        StreamBuf small, elems, keys;
        hipError_t e = small.alloc(s->p, sizeof(uint32_t) * (4 * 1024 + 1), S(stream));
        if (e == hipSuccess) e = elems.alloc(s->p, sizeof(uint16_t) * n, S(stream));
        if (e == hipSuccess && want_tie) e = keys.alloc(s->p, sizeof(uint32_t) * n, S(stream));
        if (e == hipSuccess)
            e = launch_partitioned_histogram((const uint32_t *)d_rgba, n, first_index, (uint32_t *)small.ptr, (uint16_t *)elems.ptr,
                                             (uint32_t *)keys.ptr, t.d_hist, t.d_tie, S(stream));
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP, "histogram failed: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(hipMemsetAsync(t.d_hist, 0, sizeof(uint32_t) << 24, S(stream)));
        HIP_TRY(launch_histogram((const uint32_t *)d_rgba, n, t.d_hist, S(stream)));
        if (want_tie) {
            HIP_TRY(hipMemsetAsync(t.d_tie, 0, sizeof(uint32_t) << 24, S(stream)));
            HIP_TRY(launch_tie_keys((const uint32_t *)d_rgba, n, first_index, t.d_tie, S(stream)));
        }
    }
    if (want_tie) {
        t.tie_valid = true;
        t.tie_first = first_index;
    }
    int rc_agg;
    if ((rc_agg = tables_from_histogram(s, n, S(stream))) != KMG_OK) return rc_agg;
    if (want_tie) HIP_TRY(launch_init_records(t.d_work, s->p->d_bounds, t.d_init_cells, S(stream)));   // an initialisation follows
    t.rgba = d_rgba;
    t.n = n;
    return KMG_OK;
}

extern "C" int kmg_lloyd_bind_image(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, void *stream)
try {
    const int rc = bind_image_impl(s, d_rgba, n, stream, false, 0);
    if (rc == KMG_OK) s->tab.bound_by_caller = true;
    return rc;
}
KMG_ABI_CATCH

// caller = true: the public entry point (the binding then lasts until the caller unbinds or binds again);
// false: made on behalf of one kmg_lloyd_run, which drops it before it returns
static int prepare_impl(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, int want_labels, int *strategy, void *stream, bool caller)
{
    if (!s || !d_rgba || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad prepare arguments");
    int chosen = 0;
    // the initialisation of this problem may have bound the image a moment ago: that binding is kept, and its facts are known
    const bool fresh = s->tab.rgba == d_rgba && s->tab.n == n && s->tab.bound_by_init;
    s->tab.bound_by_init = false;
    const ImageFacts before = {s->tab.n_occ, s->tab.n_hot};
    // Where the blind model's two ends disagree -- the table wins on the cheapest image of this size and loses on the dearest --
    // 16384 sampled pixels say which kind this one is (one workgroup + 12 bytes back: ~20 us, a tenth of a binding).
    bool try_table = n <= 0xFFFFFFFFull && table_pays(s->p, n, s->k, want_labels != 0, fresh ? &before : nullptr);
    if (try_table && !fresh && forced_strategy(s->p) == 0) {
        const ImageFacts dense = {kCells, 0u};
        const double N = (double)n, N22 = (double)(1u << 22);
        // (the dearest table of this size: every cell occupied, no hot cells to make the scan dearer -- with the binding on top)
        const double label_pass = !want_labels ? 0.0 : (s->k > 256u ? N * 6.7e-12 : (N <= N22 ? 3.0e-6 + N * 4.4e-12 : 2.15e-5 + (N - N22) * 2.2e-12));
        const bool sure = cube_seconds(s->k, &dense) + label_pass + bind_seconds(n) / 16.0 < scan_seconds(n, s->k, false);
        if (!sure) {
            uint32_t *d_probe = reinterpret_cast<uint32_t *>(s->d_nconv) + 1;     // (three words behind the convergence count)
            uint32_t h[3] = {0u, 0u, 1u};
            HIP_TRY(launch_sparsity_probe((const uint32_t *)d_rgba, n, d_probe, S(stream)));
            HIP_TRY(hipMemcpyAsync(h, d_probe, sizeof h, hipMemcpyDeviceToHost, S(stream)));
            HIP_TRY(hipStreamSynchronize(S(stream)));
            // occupied cells of the IMAGE from those of the sample: d = M (1 - exp(-S / M)) solved for M by fixed-point steps
            double M = (double)h[0];
            for (int it = 0; it < 8 && h[0] < h[2]; ++it) M = (double)h[0] / (1.0 - exp(-(double)h[2] / (M > 1.0 ? M : 1.0)));
            const ImageFacts guess = {(uint32_t)(M < (double)kCells ? M : (double)kCells), h[1] * 10u >= h[2] ? 1u : 0u};
            const double table = cube_seconds(s->k, &guess) + label_pass + bind_seconds(n) / 16.0;
            try_table = table < scan_seconds(n, s->k, guess.hot != 0u);
            if (log_debug()) fprintf(stderr, "[kmeans_hip] prepare: %u of %u samples' cells, %u in crowded cells -> ~%u occupied%s: %s\n", h[0], h[2], h[1],
                                     guess.occupied, guess.hot ? ", hot" : "", try_table ? "bind" : "per-pixel scan");
        }
    }
    if (try_table) {
        if (!fresh) {
            int rc = bind_image_impl(s, d_rgba, n, stream, false, 0);
            if (rc != KMG_OK) return rc;
        }
        // now the histogram has been seen: occupied cells, hot cells -- the model answers for THIS image
        const ImageFacts facts = {s->tab.n_occ, s->tab.n_hot};
        if (fresh || table_pays(s->p, n, s->k, want_labels != 0, &facts)) {
            s->tab.bound_by_caller = caller;
            chosen = 1;
        } else {
            s->tab.rgba = nullptr;   // (the blocks stay with the object: the next image may want them)
            if (log_debug()) fprintf(stderr, "[kmeans_hip] prepare: %u occupied cells, %u hot cells -- the per-pixel scan after all\n", facts.occupied, facts.hot);
        }
    } else if (s->tab.rgba == d_rgba) {
        s->tab.rgba = nullptr;   // the cost model prefers the per-pixel scan for this problem
    }
    if (strategy) *strategy = chosen;
    return KMG_OK;
}

extern "C" int kmg_lloyd_prepare(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, int want_labels, int *strategy, void *stream)
try {
    return prepare_impl(s, d_rgba, n, want_labels, strategy, stream, true);
}
KMG_ABI_CATCH

extern "C" int kmg_debug_bound_image(kmg_lloyd *s, uint64_t out[2])
try {
    if (!s || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad bound_image arguments");
    if (!s->tab.rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "no image is bound");
    out[0] = s->tab.n_occ;
    out[1] = s->tab.n_hot;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_unbind_image(kmg_lloyd *s)
try {
    if (!s) return fail(KMG_ERR_INVALID_ARGUMENT, "bad unbind_image arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipDeviceSynchronize());                                 // nothing uses the tables any more
    s->lab_pending[0] = s->lab_pending[1] = false;                   // (their label passes included)
    free_table(s->p, s->tab);
    return KMG_OK;
}
KMG_ABI_CATCH

// test support: exhaustive validation, over all 2^24 colours, of the cube pass for the current centroid
// table (run without an image: every colour counts).  out[0] = (colour, centroid) pairs whose key lies outside
// the cell's or the sub-cell's interval bounds, out[1] = colours whose brute-force arg-min is missing from the
// cell's candidate mask, out[2] = colours whose label in the per-colour table differs from the brute-force
// arg-min (this covers the sub-cell stage and the near-tie repair).
extern "C" int kmg_debug_check_table(kmg_lloyd *s, uint64_t out[3], void *stream)
try {
    if (!s || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad check_table arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc;
    if ((rc = ensure_bounds(s->p, S(stream))) != KMG_OK) return rc;
    DevBuf masks, viol, labels, sub, cwork;
    HIP_TRY(masks.alloc(cube_masks_bytes(s->k)));
    HIP_TRY(cwork.alloc(cube_work_bytes()));
    HIP_TRY(viol.alloc(3 * sizeof(unsigned long long)));
    HIP_TRY(labels.alloc((size_t)(s->k <= 256 ? 1 : 2) << 24));
    HIP_TRY(sub.alloc(sizeof(uint16_t) * (kSubCells + kCells) + sizeof(uint32_t) * kCells));
    HIP_TRY(hipMemsetAsync(viol.ptr, 0, 3 * sizeof(unsigned long long), S(stream)));
    HIP_TRY(launch_cube(nullptr, nullptr, nullptr, nullptr, nullptr, s->p->d_bounds, s->p->d_sub_bounds, s->d_cent, s->k,
                        s->p->d_lab_table, (uint64_t *)masks.ptr, cwork.ptr, labels.ptr, (uint16_t *)sub.ptr, nullptr, 0, 1u, nullptr,
                        S(stream), nullptr, affine_for(s->p, s->k, S(stream))));
    HIP_TRY(launch_check_bounds(s->p->d_bounds, s->p->d_sub_bounds, s->d_cent, s->k, (const uint64_t *)masks.ptr, labels.ptr,
                                s->p->d_lut, (unsigned long long *)viol.ptr, S(stream)));
    unsigned long long h[3];
    HIP_TRY(hipMemcpyAsync(h, viol.ptr, sizeof h, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return KMG_OK;
}
KMG_ABI_CATCH

// test / tuning support: statistics of the last colour-table pass of the bound image.
// out[0] occupied cells, [1] sum of candidate counts over occupied cells, [2] occupied cells with one
// candidate, [3] largest candidate count, [4] cells whose occupied colours share one label,
// [5] occupied sub-cells, [6] sub-cells whose occupied colours share one label, [7] distinct colours,
// [8] sub-cells the cube pass decided from their bounds, [9] sub-cells whose colours it scanned,
// [10] candidates summed over the scanned sub-cells, [11] cells with too many candidates for the sub-cell stage,
// [12] candidates the dominance phase removed from scanned sub-cells, [13] scanned sub-cells it left with one candidate
extern "C" int kmg_debug_table_stats(kmg_lloyd *s, uint64_t out[14], void *stream)
try {
    if (!s || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad table_stats arguments");
    if (!s->tab.rgba || !s->tab.tables_valid) return fail(KMG_ERR_INVALID_ARGUMENT, "no current colour table");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    int rc_;
    unsigned long long stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((rc_ = debug_refresh(s, S(stream), stage)) != KMG_OK) return rc_;
    const uint32_t words = mask_words(s->k);
    std::vector<uint64_t> masks((size_t)kCells * words);
    std::vector<int64_t> agg(4ull * kCells);
    std::vector<uint32_t> hist(1u << 24);
    std::vector<uint16_t> labels(1u << 24);
    HIP_TRY(hipMemcpy(masks.data(), s->tab.d_masks, masks.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(agg.data(), s->tab.d_agg, agg.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hist.data(), s->tab.d_hist, hist.size() * 4, hipMemcpyDeviceToHost));
    if (s->k <= 256) {
        std::vector<uint8_t> l8(1u << 24);
        HIP_TRY(hipMemcpy(l8.data(), s->tab.d_colour_labels, l8.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < l8.size(); ++i) labels[i] = l8[i];
    } else {
        HIP_TRY(hipMemcpy(labels.data(), s->tab.d_colour_labels, labels.size() * 2, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[8] = stage[2]; out[9] = stage[3]; out[10] = stage[4]; out[11] = stage[5]; out[12] = stage[6]; out[13] = stage[7];
    for (uint32_t c = 0; c < kCells; ++c) {
        if (agg[4ull * c + 3] == 0) continue;
        uint64_t pop = 0;
        for (uint32_t w = 0; w < words; ++w) pop += (uint64_t)__builtin_popcountll(masks[(size_t)c * words + w]);
        out[0] += 1; out[1] += pop; out[2] += pop == 1; out[3] = std::max<uint64_t>(out[3], pop);
        int cell_first = -1;
        bool cell_one = true;
        for (uint32_t q = 0; q < 8; ++q) {
            int first = -1;
            bool one = true;
            for (uint32_t i = 0; i < 64; ++i) {
                const uint32_t col = c * kCellColours + q * 64 + i;
                if (!hist[col]) continue;
                if (first < 0) first = labels[col];
                one = one && first == (int)labels[col];
                if (cell_first < 0) cell_first = labels[col];
                cell_one = cell_one && cell_first == (int)labels[col];
            }
            out[5] += first >= 0;
            out[6] += first >= 0 && one;
        }
        out[4] += cell_one;
    }
    for (uint32_t v : hist) out[7] += v != 0;
    return KMG_OK;
}
KMG_ABI_CATCH

// test / tuning support (k <= 256): validates the pair entries of the last colour-table pass against
// the per-colour labels.  out[0] = occupied colours whose pair entry gives a label different from the
// per-colour table (must be 0), out[1] = pixels the label pass resolves from the pair entry alone,
// out[2] = pixels of the bound image.
extern "C" int kmg_debug_check_pairs(kmg_lloyd *s, uint64_t out[3], void *stream)
try {
    if (!s || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad check_pairs arguments");
    if (!s->tab.rgba || !s->tab.tables_valid) return fail(KMG_ERR_INVALID_ARGUMENT, "no current colour table");
    if (s->k > 256) return fail(KMG_ERR_INVALID_ARGUMENT, "pair entries exist for k <= 256 only");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    int rc_;
    if ((rc_ = debug_refresh(s, S(stream), nullptr)) != KMG_OK) return rc_;
    std::vector<uint32_t> hist(1u << 24), pairs(kCells);
    std::vector<uint8_t> labels(1u << 24);
    HIP_TRY(hipMemcpy(hist.data(), s->tab.d_hist, hist.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(labels.data(), s->tab.d_colour_labels, labels.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pairs.data(), reinterpret_cast<const uint32_t *>(s->tab.d_sub + kSubCells + kCells),
                      pairs.size() * 4, hipMemcpyDeviceToHost));
    // the label pass answers the colours of a hot cell from a copy of the per-colour table in LDS (unless the cell's entry
    // already resolves all of them): resolved, and right by construction
    std::vector<uint32_t> hot(1 + kHotMax, 0u);
    if (s->tab.n_hot) HIP_TRY(hipMemcpy(hot.data(), s->tab.d_work + kCells + 1, sizeof(uint32_t) * (1 + kHotMax), hipMemcpyDeviceToHost));
    std::vector<uint8_t> in_lds(kCells, 0);
    for (uint32_t h = 0; h < s->tab.n_hot && h < kHotMax; ++h) {
        const uint32_t e = pairs[hot[1 + h]];
        const bool one_label = (e & 0xFFu) == ((e >> 8) & 0xFFu) && (e >> 23) == 0u;
        if (!one_label) in_lds[hot[1 + h]] = 1;
    }
    out[0] = out[1] = out[2] = 0;
    for (uint32_t c = 0; c < (1u << 24); ++c) {
        if (!hist[c]) continue;
        if (in_lds[c >> 9]) { out[1] += hist[c]; out[2] += hist[c]; continue; }
        uint32_t r, g, b;
        index_to_rgb(c, r, g, b);
        const uint32_t e = pairs[c >> 9];
        const uint32_t px = r | (g << 8) | (b << 16);
        const uint32_t got = pair_decode(e, pair_project(pair_dir_word((e >> 16) & 127u), px));
        out[2] += hist[c];
        if (got == kPairFine) continue;
        out[1] += hist[c];
        out[0] += got != labels[c];
    }
    return KMG_OK;
}
KMG_ABI_CATCH

// statistics / checks read the cell masks and the per-colour labels of EVERY cell, which the normal pass does
// not store: repeat the cube pass of the bound image for the current centroids with both switched on
// The balancing state of the next cube pass of a loop (kmg_table.h CubeBalance): the passes count on as long as the same work list is
// walked -- the image's, or one share of it; a new list (another binding, another share) starts over.
static CubeBalance next_balance(ColourTable &t, const uint32_t *work)
{
    if (t.balance_work != work) { t.balance_work = work; t.balance_pass = 0; }
    CubeBalance b;
    b.state = t.d_balance;
    b.pass = t.balance_pass++;
    return b;
}

static int debug_refresh(kmg_lloyd *s, hipStream_t st, unsigned long long stage[8] = nullptr)
{
    ColourTable &t = s->tab;
    int rc_;
    if (t.d_work_share) return fail(KMG_ERR_INVALID_ARGUMENT, "statistics / checks of a bound image need the whole cube: a cell share is set");
    if ((rc_ = side_flush(s, st)) != KMG_OK) return rc_;
    // d_partials is scratch here: the sums of this repeat pass and, behind them, the stage counters
    unsigned long long *d_stage = reinterpret_cast<unsigned long long *>(s->d_partials) + 4ull * s->k;
    HIP_TRY(hipMemsetAsync(s->d_partials, 0, sizeof(int64_t) * (4ull * s->k + 8ull), st));
    HIP_TRY(launch_cube(t.d_hist, t.d_agg, t.d_sub_agg, t.d_occ, t.d_work, s->p->d_bounds, s->p->d_sub_bounds, s->d_cent, s->k,
                        s->p->d_lab_table, t.d_masks, t.d_cell_work, t.d_colour_labels, t.d_sub, s->d_partials, 1u, 1u | (t.n_hot ? kCubeHot : 0u), d_stage, st,
                        nullptr, affine_for(s->p, s->k, st)));
    t.entries_valid = true;
    if (stage) HIP_TRY(hipMemcpyAsync(stage, d_stage, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return KMG_OK;
}

// label passes that kmg_lloyd_iterate left running on the side stream: `st` waits for them (no host sync)
static int side_flush(kmg_lloyd *s, hipStream_t st)
{
    for (int i = 0; i < 2; ++i)
        if (s->lab_pending[i]) {
            HIP_TRY(hipStreamWaitEvent(st, s->ev_lab[i], 0));
            s->lab_pending[i] = false;
        }
    return KMG_OK;
}

static bool table_bound(const kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n)
{
    return s->tab.rgba != nullptr && s->tab.rgba == d_rgba && s->tab.n == n;
}

// labels (optional) + sums through the colour table.  The cube workgroups add their sums into `rows` shared
// rows of `d_sums` (k x 4 int64 each): the caller's accumulators directly (rows = 1, no reduction pass), or
// the partial slab for the two-step entry points.
// the label pass's first-level tables (pair entries / cell summaries) of the current per-colour labels, if a pass deferred them
static int ensure_entries(kmg_lloyd *s, hipStream_t st)
{
    ColourTable &t = s->tab;
    if (t.entries_valid) return KMG_OK;
    HIP_TRY(launch_cube_entries(t.d_work_share ? t.d_work_share : t.d_work, t.d_occ, t.d_colour_labels, t.d_sub, s->k, st));
    t.entries_valid = true;
    return KMG_OK;
}

static int table_assign(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, int64_t *d_sums,
                        uint32_t rows, hipStream_t st, bool update_after = false, bool defer_entries = false, bool add_in_place = false)
{
    if (d_labels) defer_entries = false;
    ColourTable &t = s->tab;
    if (add_in_place) {
        // the cube pass ADDS its sums to d_sums as it stands (the caller keeps it zero between passes, or carries other sums in
        // it): no hand-over, no clearing -- no tail launch (kmg_lloyd_accumulate_into)
        if (d_labels || update_after || rows != 1u) return fail(KMG_ERR_INVALID_ARGUMENT, "table_assign: in-place sums are sums only");
        t.bound_by_init = false;
        int rc2;
        if ((rc2 = side_flush(s, st)) != KMG_OK) return rc2;
        const CubeBalance bal = next_balance(t, t.d_work_share ? t.d_work_share : t.d_work);
        PROF_LAUNCH(s, KMG_K_CUBE, st, launch_cube(t.d_hist, t.d_agg, t.d_sub_agg, t.d_occ, t.d_work_share ? t.d_work_share : t.d_work,
                                                   s->p->d_bounds, s->p->d_sub_bounds, s->d_cent, s->k, s->p->d_lab_table, t.d_masks, t.d_cell_work,
                                                   t.d_colour_labels, t.d_sub, d_sums, 1u, t.n_hot ? kCubeHot : 0u, nullptr, st, nullptr,
                                                   affine_for(s->p, s->k, st), &bal));
        t.entries_valid = true;
        t.tables_valid = t.d_work_share == nullptr;
        return KMG_OK;
    }
    // With a cell share set (kmg_lloyd_set_cell_share) a pass labels one share of the cube and returns ITS sums: the label
    // tables are complete only after the caller's all-gather, the sums only after its all-reduce.  A label map, a centroid
    // update or the two-step partial sums from such a pass would silently be those of a fraction of the image.
    if (t.d_work_share && (d_labels || update_after || rows != 1u))
        return fail(KMG_ERR_INVALID_ARGUMENT, "a cell share is set: this pass returns one share's sums only -- no label map, no "
                    "update, no partial rows (all-reduce the sums, all-gather the tables, then kmg_lloyd_labels_from_tables)");
    t.bound_by_init = false;      // only a prepare() that directly follows the initialisation may reuse its binding
    int rc_;
    if ((rc_ = side_flush(s, st)) != KMG_OK) return rc_;
    if (rows == 1u) {
        // the sums accumulate in the state's own buffer, which is zero between passes: the last launch of the cube pass
        // hands them over to d_sums, clears the buffer again and -- update_after -- updates the centroids from them
        // (CubeTail): neither a memset nor a k_update launch
        if (s->acc_int_dirty) HIP_TRY(hipMemsetAsync(s->d_acc_int, 0, sizeof(int64_t) * 4ull * s->k, st));
        s->acc_int_dirty = true;
        CubeTail tail;
        tail.acc_out = d_sums;
        tail.do_update = update_after ? 1 : 0;
        tail.convergence = s->p->opt.convergence;
        tail.cent = s->d_cent;
        tail.n_converged = s->d_nconv;
        // (k <= 32, and 32 < k <= 256 without hot cells: the cube pass is one launch, and when a label pass follows, its tail rides on that one)
        static const bool tail_rides = tools_env_int(KMG_TOOLS_ENV("KMG_TAIL_ON_LABELS"), 1) != 0;   // (tools build: 0 = a launch of its own)
        const bool tail_on_labels = tail_rides && d_labels != nullptr && s->k <= 256u && cube_single_launch(s->k, t.n_hot ? kCubeHot : 0u);
        const CubeBalance bal = next_balance(t, t.d_work_share ? t.d_work_share : t.d_work);
        PROF_LAUNCH(s, KMG_K_CUBE, st, launch_cube(t.d_hist, t.d_agg, t.d_sub_agg, t.d_occ, t.d_work_share ? t.d_work_share : t.d_work,
                                                   s->p->d_bounds, s->p->d_sub_bounds,
                                                   s->d_cent, s->k, s->p->d_lab_table, t.d_masks, t.d_cell_work, t.d_colour_labels, t.d_sub,
                                                   s->d_acc_int, 1u, (defer_entries ? kCubeNoEntries : 0u) | (t.n_hot ? kCubeHot : 0u), nullptr, st,
                                                   tail_on_labels ? nullptr : &tail, affine_for(s->p, s->k, st), &bal));
        t.entries_valid = !defer_entries;
        if (tail_on_labels) {
            t.tables_valid = true;
            PROF_LAUNCH(s, KMG_K_LABELS, st, launch_labels((const uint32_t *)d_rgba, n, t.d_colour_labels, t.d_sub, s->k, nullptr, d_labels, st,
                                                           s->reserve_cus, t.n_hot ? t.d_work + kCells + 1 : nullptr, &tail, s->d_acc_int));
            s->acc_int_dirty = false;
            if (update_after) t.tables_valid = false;
            return KMG_OK;
        }
        s->acc_int_dirty = false;
    } else {
        if (update_after) return fail(KMG_ERR_INVALID_ARGUMENT, "table_assign: update_after needs the final sums");
        HIP_TRY(hipMemsetAsync(d_sums, 0, sizeof(int64_t) * 4ull * s->k * rows, st));
        PROF_LAUNCH(s, KMG_K_CUBE, st, launch_cube(t.d_hist, t.d_agg, t.d_sub_agg, t.d_occ, t.d_work, s->p->d_bounds, s->p->d_sub_bounds,
                                                   s->d_cent, s->k, s->p->d_lab_table, t.d_masks, t.d_cell_work, t.d_colour_labels, t.d_sub,
                                                   d_sums, rows, t.n_hot ? kCubeHot : 0u, nullptr, st, nullptr, affine_for(s->p, s->k, st)));
        t.entries_valid = true;
    }
    // (a share's pass leaves the tables current for ITS cells only: kmg_lloyd_labels refuses them, _labels_from_tables -- after
    // the caller's all-gather -- takes them as they stand)
    t.tables_valid = t.d_work_share == nullptr;
    if (d_labels)
        PROF_LAUNCH(s, KMG_K_LABELS, st, launch_labels((const uint32_t *)d_rgba, n, t.d_colour_labels, t.d_sub, s->k, nullptr, d_labels, st, s->reserve_cus, t.n_hot ? t.d_work + kCells + 1 : nullptr));
    // (after an update the tables still describe the assignment just made, not the new centroids)
    if (update_after) t.tables_valid = false;
    return KMG_OK;
}

// pool_stream != NULL: the small workspace buffers are stream-ordered allocations on that stream (the
// per-call objects of the host-buffer API, which would otherwise pay ~6 hipMalloc + hipFree per call)
int lloyd_create_impl(kmg_processor *p, uint32_t k, kmg_lloyd **out, hipStream_t pool_stream)
{
    if (!p || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad lloyd_create arguments");
    *out = nullptr;
    if (k == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "k must be an integer higher than 0");  // args.rs:160-171
    if (k > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", k, KMG_MAX_K);
    HIP_TRY(hipSetDevice(p->device));
    kmg_lloyd *s = new (std::nothrow) kmg_lloyd();
    if (!s) return fail(KMG_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->p = p;
    s->k = k;
    s->d_cent = nullptr; s->d_partials = nullptr; s->d_acc = nullptr; s->d_nconv = nullptr;
    s->d_key = nullptr; s->d_dist = nullptr; s->dist_cap = 0; s->last_rows = 0; s->prof = 0; s->init_colours = false;
    s->side = nullptr; s->ev_cube = nullptr; s->ev_lab[0] = s->ev_lab[1] = nullptr;
    s->lab_pending[0] = s->lab_pending[1] = false; s->set = 0;
    s->pooled = pool_stream != nullptr;
    s->pool_stream = pool_stream;
    s->ws = nullptr; s->ws_cap = 0; s->dist_blk_cap = 0;
    s->h_slot = host_slot_take(p);
    // one block from the processor's idle blocks (a warm processor creates a kmg_lloyd without a hipMalloc)
    const size_t sizes[6] = {sizeof(Centroid) * k, sizeof(int64_t) * 4ull * k * 2048ull, sizeof(int64_t) * 4ull * k, sizeof(uint32_t),
                             sizeof(unsigned long long), sizeof(int64_t) * 4ull * k};
    size_t need = 0;
    for (size_t b : sizes) need += pad256(b);
    hipError_t e = block_take(p, need, &s->ws, &s->ws_cap);
    if (e == hipSuccess) {
        size_t off = 0;
        s->d_cent = (Centroid *)carve(s->ws, off, sizes[0]);
        s->d_partials = (int64_t *)carve(s->ws, off, sizes[1]);
        s->d_acc = (int64_t *)carve(s->ws, off, sizes[2]);
        s->d_nconv = (uint32_t *)carve(s->ws, off, sizes[3]);
        s->d_key = (unsigned long long *)carve(s->ws, off, sizes[4]);
        s->d_acc_int = (int64_t *)carve(s->ws, off, sizes[5]);
        s->acc_int_dirty = true;                                      // cleared in stream order by the first pass
        auto zero = [&](void *ptr, size_t bytes) {
            return s->pooled ? hipMemsetAsync(ptr, 0, bytes, pool_stream) : hipMemset(ptr, 0, bytes);
        };
        e = zero(s->d_cent, sizeof(Centroid) * k);                    // structures.rs:501-521
        if (e == hipSuccess) e = zero(s->d_nconv, sizeof(uint32_t));
    } else {
        s->ws = nullptr; s->ws_cap = 0;
    }
    if (e != hipSuccess) {
        kmg_lloyd_destroy(s);
        return fail(e == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP,
                    "lloyd workspace allocation failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return KMG_OK;
}

extern "C" int kmg_lloyd_create(kmg_processor *p, uint32_t k, kmg_lloyd **out)
try {
    return lloyd_create_impl(p, k, out, nullptr);
}
KMG_ABI_CATCH

extern "C" void kmg_lloyd_destroy(kmg_lloyd *s)
try {
    if (!s) return;
    (void)hipSetDevice(s->p->device);
    if (s->side) {
        (void)hipStreamSynchronize(s->side);
        (void)hipStreamDestroy(s->side);
        (void)hipEventDestroy(s->ev_cube); (void)hipEventDestroy(s->ev_lab[0]); (void)hipEventDestroy(s->ev_lab[1]);
    }
    // The blocks go back to the processor; nothing may still be using them.  Per-call objects of the host-buffer API live
    // on one private stream; for a caller's object every stream counts, as with the hipFree this replaces.
    if (s->pooled) (void)hipStreamSynchronize(s->pool_stream); else (void)hipDeviceSynchronize();
    block_give(s->p, s->ws, s->ws_cap);
    block_give(s->p, s->d_dist, s->dist_blk_cap);
    free_table(s->p, s->tab);
    destroy_events(s);
    host_slot_give(s->p, s->h_slot);
    delete s;
}
KMG_ABI_CATCH_VOID

extern "C" int kmg_lloyd_set_centroids(kmg_lloyd *s, const float *c4, void *stream)
try {
    if (!s || !c4) return fail(KMG_ERR_INVALID_ARGUMENT, "bad set_centroids arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    std::vector<Centroid> h(s->k);
    for (uint32_t i = 0; i < s->k; ++i) {
        h[i].L = c4[4 * i]; h[i].a = c4[4 * i + 1]; h[i].b = c4[4 * i + 2];
        h[i].C = chroma(h[i].a, h[i].b);
    }
    s->tab.tables_valid = false;
    HIP_TRY(hipMemcpyAsync(s->d_cent, h.data(), sizeof(Centroid) * s->k, hipMemcpyHostToDevice, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_get_centroids(kmg_lloyd *s, float *c4, void *stream)
try {
    if (!s || !c4) return fail(KMG_ERR_INVALID_ARGUMENT, "bad get_centroids arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    std::vector<Centroid> own;
    Centroid *h = static_cast<Centroid *>(s->h_slot);
    if (!h || sizeof(Centroid) * s->k > kHostSlotBytes) { own.resize(s->k); h = own.data(); }
    HIP_TRY(hipMemcpyAsync(h, s->d_cent, sizeof(Centroid) * s->k, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    for (uint32_t i = 0; i < s->k; ++i) {
        c4[4 * i] = h[i].L; c4[4 * i + 1] = h[i].a; c4[4 * i + 2] = h[i].b; c4[4 * i + 3] = 1.0f;
    }
    return KMG_OK;
}
KMG_ABI_CATCH

// Farthest-point init: k - 1 passes over the pixels (two launches, ~8e-6 s, + n * 7.0e-12 s each: sRGB->Lab + literal
// CIE94 per pixel) or over the image's colours (one launch per pass: ~1.05e-5 s once most cells are skipped, ~3e-5 s
// more for each of the first ~16 passes, which reach every cell) after binding the image (bind_seconds, x 1.5 with the
// tie keys that come with the partitioned histogram, or another atomic per pixel on small images); MI355X,
// tools/cfg3_probe.py / tools/init_phases.sh, round 2.
static bool init_table_pays(const kmg_processor *p, uint64_t n, uint32_t k)
{
    if (const int f = forced_strategy(p)) return f > 0;
    const double N = (double)n, passes = (double)(k - 1);
    // (k >= 32: k_init_multi picks up to four centroids per launch -- ~0.3 k + 18 launches -- on a grid of at most 256
    // workgroups, four literal distances per pixel and launch: 9 us + 30 ps per pixel; tools/init_crossover.py,
    // profiles/r05_init_crossover.txt)
    const double pixels = k >= 32u ? (0.3 * k + 18.0) * (9.0e-6 + N * 30.0e-12) : passes * (8.0e-6 + N * 7.0e-12);
    // (passes over the colours: 9 us each, the first 16 visit every cell: + 30 us; binding with its tie keys 1.5 x a plain bind,
    // below 2^21 pixels 0.05 ms + 0.2 ns per pixel -- refitted in round 6, tests/test_gpu_costmodel.py prints the rows)
    const double colours = passes * 0.9e-5 + (passes < 16.0 ? passes : 16.0) * 3.0e-5 +
                           (n >= (1ull << 21) ? 1.5 * bind_seconds(n) : 5.0e-5 + N * 2.0e-10);
    return colours < pixels;
}

// decides the init strategy for (d_rgba, n, first_index) and, for the colour strategy, makes sure the
// image is bound and its tie keys and per-colour distance map exist
static int init_over_colours(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint64_t first_index, bool *colours,
                             void *stream)
{
    *colours = false;
    // an initialisation starts a new problem: the image is (re)bound from the buffer's current contents,
    // so the loop that follows never works from the histogram of an earlier image in the same buffer
    if (first_index + n > 0xFFFFFFF0ull || !init_table_pays(s->p, n, s->k)) {
        if (s->tab.rgba == d_rgba) s->tab.rgba = nullptr;
        return KMG_OK;
    }
    ColourTable &t = s->tab;
    if (!t.d_tie) {
        const size_t need = pad256(sizeof(uint32_t) << 24) + pad256(sizeof(float) << 24) + pad256(init_scratch_bytes());
        const hipError_t e = block_take(s->p, need, &t.blk_init, &t.blk_init_cap);
        if (e != hipSuccess) {
            t.blk_init = nullptr; t.blk_init_cap = 0;
            return fail(e == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP, "init tables allocation failed: %s", hipGetErrorString(e));
        }
        size_t off = 0;
        t.d_tie = (uint32_t *)carve(t.blk_init, off, sizeof(uint32_t) << 24);
        t.d_cdist = (float *)carve(t.blk_init, off, sizeof(float) << 24);
        t.d_init_cells = carve(t.blk_init, off, init_scratch_bytes());
    }
    int rc;
    if ((rc = bind_image_impl(s, d_rgba, n, stream, true, first_index)) != KMG_OK) return rc;
    t.bound_by_init = true;
    *colours = true;
    return KMG_OK;
}

extern "C" int kmg_lloyd_init_centroids(kmg_lloyd *s, const uint8_t *d_rgba, uint32_t w, uint32_t h, void *stream)
try {
    if (!s || !d_rgba || !w || !h) return fail(KMG_ERR_INVALID_ARGUMENT, "bad init_centroids arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    const uint64_t n = (uint64_t)w * h;
    if (n > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image has more than 2^32-1 pixels");
    s->tab.tables_valid = false;
    // plus_plus_init.wgsl:161-168: rand(42) = 0.5625, rand(12) = 0.93359375 in IEEE binary32
    const int32_t x0 = (int32_t)((float)w * 0.5625f);
    const int32_t y0 = (int32_t)((float)h * 0.93359375f);
    const uint64_t i0 = (uint64_t)y0 * w + (uint64_t)x0;
    const uint32_t *rgba = (const uint32_t *)d_rgba;
    HIP_TRY(launch_init_first(rgba, i0, s->p->d_lut, s->d_cent, s->d_key, S(stream)));
    if (s->k > 1) {
        int rc;
        bool colours = false;
        if ((rc = init_over_colours(s, d_rgba, n, 0, &colours, stream)) != KMG_OK) return rc;
        if (!colours && s->dist_cap < n) {
            if (s->d_dist) {
                // (too small for this image: freed, not parked -- a distance map serves nothing else)
                HIP_TRY(hipStreamSynchronize(S(stream)));
                HIP_TRY(hipFree(s->d_dist));
                s->d_dist = nullptr; s->dist_cap = 0; s->dist_blk_cap = 0;
            }
            HIP_TRY(block_take(s->p, sizeof(float) * n, (void **)&s->d_dist, &s->dist_blk_cap));
            s->dist_cap = n;
        }
        // Per-pixel passes of a whole image, k >= 32: several centroids per launch (kmg_kernels.h
        // launch_init_multi).  Launches go out in chunks -- the reference submits its passes 32 at a time and polls,
        // modules.rs:949,1211-1246 -- and the number of centroids chosen so far comes back in between (a launch picks one to
        // four; one that finds the table complete does nothing).
        if (!colours && s->k >= 32u && init_multi_bytes(n) <= sizeof(int64_t) * 4ull * s->k * 2048ull) {
            uint32_t have = 1u, launch = 1u, launches = 0u;
            while (have < s->k) {
                // (launch 1 only sweeps; early launches pick ~1.3 centroids each, late ones ~3: a third of what is missing, then look)
                const uint32_t chunk = (launch == 1u ? 1u : 0u) + (s->k - have + 2u) / 3u;
                for (uint32_t q = 0; q < chunk; ++q, ++launch)
                    HIP_TRY(launch_init_multi(rgba, n, s->p->d_lut, s->d_cent, s->k, launch, s->d_dist, s->d_partials, S(stream)));
                launches += chunk;
                uint32_t *h = s->h_slot ? static_cast<uint32_t *>(s->h_slot) : &have;
                HIP_TRY(hipMemcpyAsync(h, init_multi_count(s->d_partials, n, launch - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, S(stream)));
                HIP_TRY(hipStreamSynchronize(S(stream)));
                have = *static_cast<volatile uint32_t *>(h);
                if (have == 0u || have > s->k) return fail(KMG_ERR_HIP, "initialisation: centroid count %u out of range", have);
            }
            if (log_debug()) fprintf(stderr, "[kmeans_hip] initialisation: %u centroids in %u launches\n", s->k, launches);
            return KMG_OK;
        }
        // Passes over the colours, k >= 32: several centroids per launch as well (kmg_table.h launch_init_cells_multi; early
        // launches pick ~2, late ones ~3.6 of at most 4: a third of what is missing, then look)
        if (colours && s->k >= 32u) {
            uint32_t have = 1u, launch = 1u, launches = 0u;
            while (have < s->k) {
                const uint32_t chunk = (launch == 1u ? 1u : 0u) + (s->k - have + 2u) / 3u;
                for (uint32_t q = 0; q < chunk; ++q, ++launch)
                    HIP_TRY(launch_init_cells_multi(s->tab.d_tie, s->tab.d_occ, s->p->d_lab_table, s->d_cent, s->k, launch, s->tab.d_cdist,
                                                    s->tab.d_init_cells, rgba, s->p->d_lut, S(stream)));
                launches += chunk;
                uint32_t *h = s->h_slot ? static_cast<uint32_t *>(s->h_slot) : &have;
                HIP_TRY(hipMemcpyAsync(h, init_cells_multi_count(s->tab.d_init_cells, launch - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost,
                                       S(stream)));
                HIP_TRY(hipStreamSynchronize(S(stream)));
                have = *static_cast<volatile uint32_t *>(h);
                if (have == 0u || have > s->k) return fail(KMG_ERR_HIP, "initialisation: centroid count %u out of range", have);
            }
            if (log_debug()) fprintf(stderr, "[kmeans_hip] initialisation over the colours: %u centroids in %u launches\n", s->k, launches);
            return KMG_OK;
        }
        for (uint32_t j = 1; j < s->k + (colours ? 1u : 0u); ++j) {   // modules.rs:1211-1246
            if (colours) {
                // launch j picks centroid j - 1 and runs pass j; launch k only picks
                HIP_TRY(launch_init_pass_cells(s->tab.d_tie, s->tab.d_occ, s->p->d_lab_table, s->d_cent, j,
                                               j < s->k ? 1 : 0, s->tab.d_cdist, s->tab.d_init_cells, nullptr, rgba, s->p->d_lut,
                                               S(stream)));
            } else {
                // launch j picks centroid j - 1 (j >= 2) and runs pass j; the workgroups' keys travel through two slot sets
                // in the partial-sum slab, which nothing else uses during the initialisation (kmg_kernels.h)
                HIP_TRY(launch_init_pass(rgba, n, s->p->d_lut, s->d_cent, j, s->d_dist, (unsigned long long *)s->d_partials, 0, S(stream), true));
            }
        }
        if (!colours)       // the last centroid, from the last pass's slots
            HIP_TRY(launch_init_pick_slots(rgba, n, s->p->d_lut, (const unsigned long long *)s->d_partials, s->d_cent, s->k - 1u, S(stream)));
    }
    return KMG_OK;
}
KMG_ABI_CATCH

// ---- the same initialisation for an image sharded in row bands (last row) ----
// One step = local pass (running min-distance map of this band, arg-max key over IMAGE-wide pixel
// indices) -> caller all-reduces the key (max) -> kmg_lloyd_init_pick_band publishes the winning
// pixel's colour from the band that owns it -> caller all-reduces {colour, 1} (sum) ->
// kmg_lloyd_set_centroid_rgba.  No host synchronisation is involved.
extern "C" int kmg_lloyd_init_step(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_local, uint64_t first_index,
                                   uint32_t j, uint64_t *d_key, void *stream)
try {
    if (!s || !d_key || j == 0 || j >= s->k || (n_local && !d_rgba))
        return fail(KMG_ERR_INVALID_ARGUMENT, "bad init_step arguments");
    if (first_index + n_local > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image has more than 2^32-1 pixels");
    HIP_TRY(hipSetDevice(s->p->device));
    s->tab.tables_valid = false;
    HIP_TRY(hipMemsetAsync(d_key, 0, sizeof(uint64_t), S(stream)));
    if (n_local == 0) return KMG_OK;
    if (j == 1) {
        int rc;
        if ((rc = init_over_colours(s, d_rgba, n_local, first_index, &s->init_colours, stream)) != KMG_OK) return rc;
    }
    if (s->init_colours) {
        const ColourTable &t = s->tab;
        if (t.rgba != d_rgba || t.n != n_local || !t.tie_valid || t.tie_first != first_index)
            return fail(KMG_ERR_INVALID_ARGUMENT, "init_step: the band changed since step j = 1");
        HIP_TRY(launch_init_pass_cells(t.d_tie, t.d_occ, s->p->d_lab_table, s->d_cent, j, 1, t.d_cdist,
                                       t.d_init_cells, (unsigned long long *)d_key, nullptr, nullptr, S(stream)));
        return KMG_OK;
    }
    if (s->dist_cap < n_local) {
        if (j != 1) return fail(KMG_ERR_INVALID_ARGUMENT, "init_step: the distance map of this band was never started (j = 1)");
        if (s->d_dist) {
            HIP_TRY(hipStreamSynchronize(S(stream)));
            HIP_TRY(hipFree(s->d_dist));
            s->d_dist = nullptr; s->dist_cap = 0; s->dist_blk_cap = 0;
        }
        HIP_TRY(block_take(s->p, sizeof(float) * n_local, (void **)&s->d_dist, &s->dist_blk_cap));
        s->dist_cap = n_local;
    }
    HIP_TRY(launch_init_pass((const uint32_t *)d_rgba, n_local, s->p->d_lut, s->d_cent, j, s->d_dist,
                             (unsigned long long *)d_key, first_index, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_init_pick_band(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n_local, uint64_t first_index,
                                        const uint64_t *d_key, uint32_t *d_colour2, void *stream)
try {
    if (!s || !d_key || !d_colour2 || (n_local && !d_rgba)) return fail(KMG_ERR_INVALID_ARGUMENT, "bad init_pick_band arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    HIP_TRY(launch_init_pick_band((const uint32_t *)d_rgba, n_local, first_index, (const unsigned long long *)d_key,
                                  d_colour2, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_set_centroid_rgba(kmg_lloyd *s, uint32_t j, const uint32_t *d_colour, void *stream)
try {
    if (!s || !d_colour || j >= s->k) return fail(KMG_ERR_INVALID_ARGUMENT, "bad set_centroid_rgba arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    s->tab.tables_valid = false;
    HIP_TRY(launch_set_centroid_rgba(d_colour, s->p->d_lut, s->d_cent, j, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" uint64_t kmg_init_first_key(uint32_t width, uint32_t height)
{
    // plus_plus_init.wgsl:161-168 `initial`: the key that names pixel (floor(w rand(42)), floor(h rand(12)))
    const int32_t x0 = (int32_t)((float)width * 0.5625f);
    const int32_t y0 = (int32_t)((float)height * 0.93359375f);
    const uint64_t i0 = (uint64_t)y0 * width + (uint64_t)x0;
    return (1ull << 32) | (uint64_t)((((uint32_t)(i0 >> 4)) << 4) | (15u - (uint32_t)(i0 & 15u)));
}

// One pass: labels and/or the partial sums of the current centroid table.  Uses the colour table
// when this image is bound (kmg_lloyd_bind_image), the per-pixel scan otherwise; both give the
// same labels and the same integer sums.
static int assign_pass(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, bool sums, hipStream_t st)
{
    if (table_bound(s, d_rgba, n)) {
        s->last_rows = kMergeRows;
        return table_assign(s, d_rgba, n, d_labels, s->d_partials, kMergeRows, st);
    }
    // a label pass kmg_lloyd_iterate left on the side stream may still be writing d_labels
    int rc_;
    if ((rc_ = side_flush(s, st)) != KMG_OK) return rc_;
    s->last_rows = assign_grid(n);
    PROF_LAUNCH(s, KMG_K_ASSIGN, st, launch_assign((const uint32_t *)d_rgba, n, s->d_cent, s->k, s->p->d_lut, d_labels,
                                                  sums ? s->d_partials : nullptr, st));
    return KMG_OK;
}

extern "C" int kmg_lloyd_assign_accumulate(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels,
                                           int64_t *d_acc4, void *stream)
try {
    if (!s || !d_rgba || n == 0 || (!d_labels && !d_acc4))
        return fail(KMG_ERR_INVALID_ARGUMENT, "bad assign_accumulate arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    if (d_acc4 && table_bound(s, d_rgba, n))        // the cube pass adds straight into d_acc4: no reduction pass
        return table_assign(s, d_rgba, n, d_labels, d_acc4, 1u, S(stream));
    int rc;
    if ((rc = assign_pass(s, d_rgba, n, d_labels, d_acc4 != nullptr, S(stream))) != KMG_OK) return rc;
    if (d_acc4) PROF_LAUNCH(s, KMG_K_REDUCE, S(stream), launch_reduce_partials(s->d_partials, s->last_rows, s->k, d_acc4, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_accumulate_into(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, int64_t *d_acc4, void *stream)
try {
    if (!s || !d_rgba || !d_acc4 || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad accumulate_into arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    if (!table_bound(s, d_rgba, n)) return fail(KMG_ERR_INVALID_ARGUMENT, "accumulate_into: the image is not bound (kmg_lloyd_bind_image / _prepare)");
    return table_assign(s, d_rgba, n, nullptr, d_acc4, 1u, S(stream), false, false, true);
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_assign_partials(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, void *stream)
try {
    if (!s || !d_rgba || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad assign_partials arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    return assign_pass(s, d_rgba, n, d_labels, true, S(stream));
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_reserve_cus(kmg_lloyd *s, uint32_t n_cus)
try {
    if (!s || n_cus > 128u) return fail(KMG_ERR_INVALID_ARGUMENT, "bad reserve_cus arguments");
    s->reserve_cus = n_cus;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_labels(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, void *stream)
try {
    if (!s || !d_rgba || !d_labels || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad labels arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc_;
    if ((rc_ = side_flush(s, S(stream))) != KMG_OK) return rc_;      // both branches write d_labels
    if (table_bound(s, d_rgba, n) && s->tab.tables_valid) {
        if ((rc_ = ensure_entries(s, S(stream))) != KMG_OK) return rc_;
        PROF_LAUNCH(s, KMG_K_LABELS, S(stream), launch_labels((const uint32_t *)d_rgba, n, s->tab.d_colour_labels,
                                                              s->tab.d_sub, s->k, nullptr, d_labels, S(stream), s->reserve_cus,
                                                              s->tab.n_hot ? s->tab.d_work + kCells + 1 : nullptr));
        return KMG_OK;
    }
    PROF_LAUNCH(s, KMG_K_ASSIGN, S(stream), launch_assign((const uint32_t *)d_rgba, n, s->d_cent, s->k, s->p->d_lut,
                                                          d_labels, nullptr, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

// Cell-sharded cube pass (include/kmeans_hip.h)
extern "C" int kmg_lloyd_set_cell_share(kmg_lloyd *s, uint32_t part, uint32_t parts, void *stream)
try {
    if (!s || parts == 0 || part >= parts) return fail(KMG_ERR_INVALID_ARGUMENT, "bad set_cell_share arguments");
    if (!s->tab.rgba || !s->tab.d_hist) return fail(KMG_ERR_INVALID_ARGUMENT, "set_cell_share: no bound image");
    HIP_TRY(hipSetDevice(s->p->device));
    ColourTable &t = s->tab;
    t.balance_work = nullptr;                                        // (the share buffer is reused: the same pointer, another list)
    if (parts == 1u) { t.d_work_share = nullptr; return KMG_OK; }
    HIP_TRY(launch_work_share(t.d_work, part, parts, t.share_buf, S(stream)));
    t.d_work_share = t.share_buf;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_labels_from_tables(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, void *stream)
try {
    if (!s || !d_rgba || !d_labels || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad labels_from_tables arguments");
    if (!s->tab.rgba || !s->tab.d_hist) return fail(KMG_ERR_INVALID_ARGUMENT, "labels_from_tables: no bound image");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc_;
    if ((rc_ = side_flush(s, S(stream))) != KMG_OK) return rc_;
    if ((rc_ = ensure_entries(s, S(stream))) != KMG_OK) return rc_;
    PROF_LAUNCH(s, KMG_K_LABELS, S(stream), launch_labels((const uint32_t *)d_rgba, n, s->tab.d_colour_labels, s->tab.d_sub, s->k,
                                                          nullptr, d_labels, S(stream), s->reserve_cus,
                                                          s->tab.n_hot ? s->tab.d_work + kCells + 1 : nullptr));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_labels_from_tables_update(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, int64_t *d_acc4,
                                                   void *stream)
try {
    if (!s || !d_rgba || !d_labels || !d_acc4 || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad labels_from_tables_update arguments");
    if (!s->tab.rgba || !s->tab.d_hist) return fail(KMG_ERR_INVALID_ARGUMENT, "labels_from_tables_update: no bound image");
    if (s->k > 256u) return fail(KMG_ERR_UNSUPPORTED, "labels_from_tables_update: k <= 256 (the label pass that carries a tail)");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc_;
    if ((rc_ = side_flush(s, S(stream))) != KMG_OK) return rc_;
    if ((rc_ = ensure_entries(s, S(stream))) != KMG_OK) return rc_;
    // the label pass's last workgroup copies the sums to the object's own k x 4 buffer, updates the centroids from them
    // (choose_centroid.wgsl:180-206) and clears d_acc4 (kmg_table.h CubeTail)
    CubeTail tail;
    tail.acc_out = s->d_acc;
    tail.do_update = 1;
    tail.convergence = s->p->opt.convergence;
    tail.cent = s->d_cent;
    tail.n_converged = s->d_nconv;
    PROF_LAUNCH(s, KMG_K_LABELS, S(stream), launch_labels((const uint32_t *)d_rgba, n, s->tab.d_colour_labels, s->tab.d_sub, s->k,
                                                          nullptr, d_labels, S(stream), s->reserve_cus,
                                                          s->tab.n_hot ? s->tab.d_work + kCells + 1 : nullptr, &tail, d_acc4));
    s->tab.tables_valid = false;                                 // (the tables describe the assignment just made, not the new centroids)
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_histogram_buffer(kmg_lloyd *s, void **hist, uint64_t *bytes)
try {
    if (!s || !s->tab.d_hist || !s->tab.rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "histogram_buffer: no bound image");
    if (hist) *hist = s->tab.d_hist;
    if (bytes) *bytes = sizeof(uint32_t) << 24;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_rebuild_from_histogram(kmg_lloyd *s, uint64_t n_pixels, void *stream)
try {
    if (!s || !s->tab.d_hist || !s->tab.rgba || n_pixels == 0 || n_pixels > 0xFFFFFFFFull)
        return fail(KMG_ERR_INVALID_ARGUMENT, "bad rebuild_from_histogram arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc_;
    if ((rc_ = side_flush(s, S(stream))) != KMG_OK) return rc_;
    s->tab.tables_valid = false;
    s->tab.tie_valid = false;                                        // the init's tie keys belong to the band's own histogram
    return tables_from_histogram(s, n_pixels, S(stream));
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_table_buffers(kmg_lloyd *s, void **colour_labels, uint64_t *colour_label_bytes, void **entries,
                                       uint64_t *entry_bytes)
try {
    if (!s || !s->tab.d_hist) return fail(KMG_ERR_INVALID_ARGUMENT, "table_buffers: no bound image");
    if (colour_labels) *colour_labels = s->tab.d_colour_labels;
    if (colour_label_bytes) *colour_label_bytes = (uint64_t)(s->k <= 256 ? 1 : 2) << 24;
    if (entries) *entries = s->tab.d_sub;
    if (entry_bytes) *entry_bytes = sub_table_bytes();
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_reduce_partials(kmg_lloyd *s, uint64_t n, int64_t *d_acc4, void *stream)
try {
    if (!s || !d_acc4 || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad reduce_partials arguments");
    if (s->last_rows == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "reduce_partials without a preceding assign_partials");
    HIP_TRY(hipSetDevice(s->p->device));
    PROF_LAUNCH(s, KMG_K_REDUCE, S(stream), launch_reduce_partials(s->d_partials, s->last_rows, s->k, d_acc4, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

static void drop_events(kmg_lloyd *s)
{
    for (ProfEvent &e : s->events) { s->pool.push_back(e.e0); s->pool.push_back(e.e1); }
    s->events.clear();
}

static void destroy_events(kmg_lloyd *s)
{
    drop_events(s);
    for (hipEvent_t e : s->pool) (void)hipEventDestroy(e);
    s->pool.clear();
}

extern "C" const char *kmg_kernel_name(int id)
{
    static const char *names[KMG_K_COUNT] = {"k_assign", "k_reduce_partials", "k_update", "k_cell_candidates", "k_cube", "k_labels"};
    return (id >= 0 && id < KMG_K_COUNT) ? names[id] : "?";
}

extern "C" int kmg_lloyd_profile(kmg_lloyd *s, int enable)
try {
    if (!s) return fail(KMG_ERR_INVALID_ARGUMENT, "bad profile arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    drop_events(s);
    s->prof = enable < 0 ? 0xFFFFFFFFu : (uint32_t)enable;
    if (s->prof)   // pre-create a pool so the timed region does not pay for event creation
        while (s->pool.size() < 512) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            s->pool.push_back(e);
        }
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_profile_read(kmg_lloyd *s, double total_ms[KMG_K_COUNT], uint32_t launches[KMG_K_COUNT])
try {
    if (!s || !total_ms || !launches) return fail(KMG_ERR_INVALID_ARGUMENT, "bad profile_read arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    for (int i = 0; i < KMG_K_COUNT; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
    for (ProfEvent &e : s->events) {
        HIP_TRY(hipEventSynchronize(e.e1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, e.e0, e.e1));
        total_ms[e.id] += ms;
        launches[e.id] += 1;
    }
    drop_events(s);
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_update(kmg_lloyd *s, const int64_t *d_acc4, void *stream)
try {
    if (!s || !d_acc4) return fail(KMG_ERR_INVALID_ARGUMENT, "bad update arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    s->tab.tables_valid = false;
    PROF_LAUNCH(s, KMG_K_UPDATE, S(stream), launch_update(d_acc4, s->k, s->p->opt.convergence, s->d_cent, s->d_nconv, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

// Assign, then update (include/kmeans_hip.h).  With a bound image the update rides on the last launch of the cube pass.
static int assign_update_impl(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, int64_t *d_acc4,
                              int do_update, void *stream, bool defer_entries)
{
    if (!s || !d_rgba || n == 0 || !d_acc4) return fail(KMG_ERR_INVALID_ARGUMENT, "bad assign_update arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    if (table_bound(s, d_rgba, n)) return table_assign(s, d_rgba, n, d_labels, d_acc4, 1u, S(stream), do_update != 0, defer_entries);
    int rc;
    // (a small slab of partial sums: reduction and update are one launch -- kmg_kernels.h reduce_update_fits)
    if ((rc = assign_pass(s, d_rgba, n, d_labels, true, S(stream))) != KMG_OK) return rc;
    if (reduce_update_fits(s->last_rows, s->k)) {
        if (do_update) s->tab.tables_valid = false;
        PROF_LAUNCH(s, KMG_K_REDUCE, S(stream), launch_reduce_update(s->d_partials, s->last_rows, s->k, d_acc4, do_update, s->p->opt.convergence,
                                                                    s->d_cent, s->d_nconv, S(stream)));
        return KMG_OK;
    }
    PROF_LAUNCH(s, KMG_K_REDUCE, S(stream), launch_reduce_partials(s->d_partials, s->last_rows, s->k, d_acc4, S(stream)));
    return do_update ? kmg_lloyd_update(s, d_acc4, stream) : KMG_OK;
}

extern "C" int kmg_lloyd_assign_update(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, int64_t *d_acc4,
                                       int do_update, void *stream)
try {
    return assign_update_impl(s, d_rgba, n, d_labels, d_acc4, do_update, stream, false);
}
KMG_ABI_CATCH

// One Lloyd iteration with the label pass taken off the critical path (modules.rs:769-800: update, then
// re-assign).  The loop only depends on the sums; with the colour table they come from the cube pass, and the
// label pass that turns the cube pass's tables into the per-pixel label map feeds nothing.  So the label pass
// of iteration t runs on a stream of its own while the main stream already updates the centroids and runs the
// cube pass of iteration t + 1 -- a memory-bound kernel beside an issue-bound one.  Two sets of label tables
// alternate; the cube pass that is about to overwrite a set first waits for the label pass that read it.
extern "C" int kmg_lloyd_iterate(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels, int64_t *d_acc4,
                                 int update_first, void *stream)
try {
    if (!s || !d_rgba || n == 0 || !d_acc4) return fail(KMG_ERR_INVALID_ARGUMENT, "bad iterate arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    int rc;
    if (!table_bound(s, d_rgba, n) || !d_labels) {
        // per-pixel scan (labels and sums come out of one kernel) or no label map wanted: nothing to overlap
        if (update_first && (rc = kmg_lloyd_update(s, d_acc4, stream)) != KMG_OK) return rc;
        return kmg_lloyd_assign_accumulate(s, d_rgba, n, d_labels, d_acc4, stream);
    }
    hipStream_t st = S(stream);
    ColourTable &t = s->tab;
    if (t.d_work_share) return fail(KMG_ERR_INVALID_ARGUMENT, "iterate: a cell share is set (kmg_lloyd_set_cell_share)");
    if (!s->side) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_cube, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_lab[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_lab[1], hipEventDisableTiming));
    }
    const size_t sub_bytes = sub_table_bytes();
    if (!t.d_colour_labels_alt) {
        const size_t lab_bytes = (size_t)(s->k <= 256 ? 1 : 2) << 24;
        HIP_TRY(block_take(s->p, pad256(lab_bytes) + pad256(sub_bytes), &t.blk_alt, &t.blk_alt_cap));
        size_t off = 0;
        t.d_colour_labels_alt = carve(t.blk_alt, off, lab_bytes);
        t.d_sub_alt = (uint16_t *)carve(t.blk_alt, off, sub_bytes);
        HIP_TRY(hipMemsetAsync(t.d_sub_alt, 0xFF, sub_bytes, st));     // as bind_image_impl does for the first set
    }
    // write the other set; its last reader (the label pass of two iterations ago) must be through.  Until the cube
    // pass has been issued the set holds the tables of two iterations ago (or nothing): not valid.  If a call in between
    // fails, the swap is undone, so a later label pass never gathers from that set.
    std::swap(t.d_colour_labels, t.d_colour_labels_alt);
    std::swap(t.d_sub, t.d_sub_alt);
    s->set ^= 1;
    t.tables_valid = false;
    t.bound_by_init = false;
    auto issue = [&]() -> int {
        if (s->lab_pending[s->set]) {
            HIP_TRY(hipStreamWaitEvent(st, s->ev_lab[s->set], 0));
            s->lab_pending[s->set] = false;
        }
        if (update_first)
            PROF_LAUNCH(s, KMG_K_UPDATE, st, launch_update(d_acc4, s->k, s->p->opt.convergence, s->d_cent, s->d_nconv, st));
        HIP_TRY(hipMemsetAsync(d_acc4, 0, sizeof(int64_t) * 4ull * s->k, st));
        const CubeBalance bal = next_balance(t, t.d_work);
        PROF_LAUNCH(s, KMG_K_CUBE, st, launch_cube(t.d_hist, t.d_agg, t.d_sub_agg, t.d_occ, t.d_work, s->p->d_bounds, s->p->d_sub_bounds,
                                                   s->d_cent, s->k, s->p->d_lab_table, t.d_masks, t.d_cell_work, t.d_colour_labels, t.d_sub,
                                                   d_acc4, 1u, t.n_hot ? kCubeHot : 0u, nullptr, st, nullptr, affine_for(s->p, s->k, st), &bal));
        return KMG_OK;
    };
    if ((rc = issue()) != KMG_OK) {
        std::swap(t.d_colour_labels, t.d_colour_labels_alt);
        std::swap(t.d_sub, t.d_sub_alt);
        s->set ^= 1;
        return rc;
    }
    t.tables_valid = true;
    t.entries_valid = true;
    HIP_TRY(hipEventRecord(s->ev_cube, st));
    HIP_TRY(hipStreamWaitEvent(s->side, s->ev_cube, 0));
    PROF_LAUNCH(s, KMG_K_LABELS, s->side, launch_labels((const uint32_t *)d_rgba, n, t.d_colour_labels, t.d_sub, s->k, nullptr,
                                                        d_labels, s->side, s->reserve_cus, t.n_hot ? t.d_work + kCells + 1 : nullptr));
    HIP_TRY(hipEventRecord(s->ev_lab[s->set], s->side));
    s->lab_pending[s->set] = true;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_flush(kmg_lloyd *s, void *stream)
try {
    if (!s) return fail(KMG_ERR_INVALID_ARGUMENT, "bad flush arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    return side_flush(s, S(stream));
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_converged_count(kmg_lloyd *s, uint32_t *count, void *stream)
try {
    if (!s || !count) return fail(KMG_ERR_INVALID_ARGUMENT, "bad converged_count arguments");
    HIP_TRY(hipSetDevice(s->p->device));
    if (s->h_slot) {
        HIP_TRY(hipMemcpyAsync(s->h_slot, s->d_nconv, sizeof(uint32_t), hipMemcpyDeviceToHost, S(stream)));
        HIP_TRY(hipStreamSynchronize(S(stream)));
        *count = *static_cast<const volatile uint32_t *>(s->h_slot);
        return KMG_OK;
    }
    HIP_TRY(hipMemcpyAsync(count, s->d_nconv, sizeof(uint32_t), hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_lloyd_run(kmg_lloyd *s, const uint8_t *d_rgba, uint64_t n, uint32_t *d_labels,
                             uint32_t *iterations, void *stream)
try {
    if (!s || !d_rgba || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad lloyd_run arguments");
    if (s->tab.d_work_share && table_bound(s, d_rgba, n))
        return fail(KMG_ERR_INVALID_ARGUMENT, "lloyd_run: a cell share is set (kmg_lloyd_set_cell_share): the loop would update from one share's sums");
    const kmg_options &o = s->p->opt;
    int rc;
    // large problems iterate over the image's colour table instead of its pixels (same results);
    // the loop itself only needs the sums, so with the table the per-pixel label map is written
    // once, after the last iteration (the per-pixel scan writes it in the same pass for free)
    // A binding made by the caller (kmg_lloyd_bind_image / _prepare) is trusted.  Any other one is (re)made
    // here from the buffer's CURRENT contents -- only the initialisation of this very problem may hand its
    // binding over (prepare keeps it) -- and dropped before returning: a later run on the same buffer with
    // new pixels must not meet the histogram of the old ones.
    const bool callers = table_bound(s, d_rgba, n) && s->tab.bound_by_caller;
    if (!callers)
        if ((rc = prepare_impl(s, d_rgba, n, 0, nullptr, stream, false)) != KMG_OK) return rc;
    const bool table = table_bound(s, d_rgba, n);
    uint32_t *loop_labels = table ? nullptr : d_labels;
    // operations.rs:75-83 initial assignment (fused with the sums the first update needs), then modules.rs:769-800:
    // update (:773-788), re-assign (:793-800), every check_period-th iteration read the convergence count (:802-836).
    // The update of iteration `it` rides on the assign pass before it (kmg_lloyd_assign_update) unless the loop may stop
    // in between -- i.e. unless that pass is the one a convergence check follows: after the last update nothing but the
    // re-assignment may happen.
    auto checked = [&](uint32_t it) { return it > 0 && it % o.check_period == 0; };
    if (!table && assign_loop_fits(n) && assign_loop_scratch_bytes(s->k) <= sizeof(int64_t) * 4ull * s->k * 2048ull) {
        // Small image (the reference's default working size): one launch per iteration (kmg_kernels.h launch_assign_loop).
        // The partial-sum slab, unused by this loop, holds its three sum buffers and the second centroid buffer.
        hipStream_t st = S(stream);
        if ((rc = side_flush(s, st)) != KMG_OK) return rc;
        int64_t *acc3[3] = {s->d_partials, s->d_partials + 4ull * s->k, s->d_partials + 8ull * s->k};
        Centroid *cur = s->d_cent, *alt = reinterpret_cast<Centroid *>(s->d_partials + 12ull * s->k);
        HIP_TRY(hipMemsetAsync(s->d_partials, 0, sizeof(int64_t) * 12ull * s->k, st));
        const uint32_t *px = (const uint32_t *)d_rgba;
        s->tab.tables_valid = false;
        // operations.rs:75-83: the initial assignment
        PROF_LAUNCH(s, KMG_K_ASSIGN, st, launch_assign_loop(px, n, cur, alt, s->k, s->p->d_lut, d_labels, nullptr, acc3[0], acc3[1], 0,
                                                          o.convergence, s->d_nconv, st));
        uint32_t it = 0;
        for (it = 0; it < o.max_iterations; ++it) {                   // modules.rs:769: update (:773-788), re-assign (:793-800)
            const uint32_t l = it + 1u;
            PROF_LAUNCH(s, KMG_K_ASSIGN, st, launch_assign_loop(px, n, cur, alt, s->k, s->p->d_lut, d_labels, acc3[(l + 2u) % 3u], acc3[l % 3u],
                                                              acc3[(l + 1u) % 3u], 1, o.convergence, s->d_nconv, st));
            std::swap(cur, alt);
            if (checked(it)) {                                       // :802
                uint32_t conv = 0;
                if ((rc = kmg_lloyd_converged_count(s, &conv, stream)) != KMG_OK) return rc;
                if (conv >= s->k) {                                  // :826-831
                    if (log_debug()) fprintf(stderr, "[kmeans_hip] We have convergence, checked at iteration %u\n", it);
                    break;
                }
            }
        }
        if (cur != s->d_cent) HIP_TRY(hipMemcpyAsync(s->d_cent, cur, sizeof(Centroid) * s->k, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (iterations) *iterations = it < o.max_iterations ? it : o.max_iterations - 1;
        return KMG_OK;
    }
    // pass(it) = the assign pass of iteration it (it = -1: the initial one); its update-after is iteration it + 1's update
    auto pass = [&](long it) -> int {
        const bool may_stop_here = it >= 0 && checked((uint32_t)it);          // a check follows this pass
        const bool last = it + 1 >= (long)o.max_iterations;
        const bool fuse = !may_stop_here && !last;
        // (with the table the loop needs only the sums: the pair entries of the label pass are derived once, before the
        // final label pass -- 19 us per iteration at k = 256)
        return assign_update_impl(s, d_rgba, n, loop_labels, s->d_acc, fuse ? 1 : 0, stream, true);
    };
    if ((rc = pass(-1)) != KMG_OK) return rc;
    uint32_t it = 0;
    for (it = 0; it < o.max_iterations; ++it) {                       // modules.rs:769
        // the update of this iteration: already done by the previous pass unless that pass was followed by a check
        const bool fused_before = !(it >= 1 && checked(it - 1));
        if (!fused_before && (rc = kmg_lloyd_update(s, s->d_acc, stream)) != KMG_OK) return rc;   // :773-788
        if ((rc = pass((long)it)) != KMG_OK) return rc;                                            // :793-800
        if (checked(it)) {                                           // :802
            uint32_t conv = 0;
            if ((rc = kmg_lloyd_converged_count(s, &conv, stream)) != KMG_OK) return rc;
            if (conv >= s->k) {                                      // :826-831
                if (log_debug()) fprintf(stderr, "[kmeans_hip] We have convergence, checked at iteration %u\n", it);
                break;
            }
        }
    }
    if (table && d_labels && (rc = ensure_entries(s, S(stream))) != KMG_OK) return rc;
    if (table && d_labels)   // the label tables of the last pass belong to the final centroids
        PROF_LAUNCH(s, KMG_K_LABELS, S(stream), launch_labels((const uint32_t *)d_rgba, n, s->tab.d_colour_labels,
                                                              s->tab.d_sub, s->k, nullptr, d_labels, S(stream), s->reserve_cus,
                                                              s->tab.n_hot ? s->tab.d_work + kCells + 1 : nullptr));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    if (table && !callers) { s->tab.rgba = nullptr; s->tab.tables_valid = false; }
    if (iterations) *iterations = it < o.max_iterations ? it : o.max_iterations - 1;
    return KMG_OK;
}
KMG_ABI_CATCH

