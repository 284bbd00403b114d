// kmg_processor.hip -- kmg_processor of the C ABI (include/kmeans_hip.h): the error channel, ImageProcessor::new
// (core/src/lib.rs:38-65) on one device, the device blocks and page-locked slots a processor keeps between calls, and the host
// colour helpers the reference takes from the `palette` crate.  There is no CPU data path: without a HIP device the processor
// cannot be created.

#include <stddef.h>

#include "kmg_state.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int kmg::fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    if (const char *lv = getenv("KMG_LOG"))
        if (!strcmp(lv, "debug") || !strcmp(lv, "error")) fprintf(stderr, "[kmeans_hip] error %d: %s\n", code, g_err);
    return code;
}

// called inside a catch (...) handler: rethrows to classify (kmg_internal.h)
int kmg::abi_trap() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(KMG_ERR_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)");
    } catch (const std::exception &e) {
        return fail(KMG_ERR_HIP, "internal error: %s", e.what());
    } catch (...) {
        return fail(KMG_ERR_HIP, "internal error: unknown C++ exception");
    }
}

bool kmg::log_debug()
{
    const char *lv = getenv("KMG_LOG");
    return lv && !strcmp(lv, "debug");
}

extern "C" const char *kmg_last_error(void) { return g_err; }
extern "C" const char *kmg_version(void) { return "kmeans_hip 0.1 (gfx950)"; }

extern "C" void kmg_default_options(kmg_options *opt)
try {
    if (!opt) return;
    opt->struct_size = sizeof(kmg_options);
    opt->device = -1;
    opt->shrink_max_dim = 256;   // structures.rs:23
    opt->max_iterations = 128;   // modules.rs:765
    opt->check_period = 8;       // modules.rs:766
    opt->convergence = 1.0f;     // lib.rs:189-194
    opt->strategy = KMG_STRATEGY_AUTO;
}
KMG_ABI_CATCH_VOID

// ---------------------------------------------------------------------------------------------
// processor
// ---------------------------------------------------------------------------------------------
void *host_slot_take(kmg_processor *p)
{
    std::lock_guard<std::mutex> lock(p->mu);
    if (!p->h_page_tried) {
        p->h_page_tried = true;
        if (hipHostMalloc(&p->h_page, kHostPageBytes, hipHostMallocDefault) != hipSuccess) { p->h_page = nullptr; (void)hipGetLastError(); }
        else for (uint16_t i = 0; i < kHostPageBytes / kHostSlotBytes; ++i) p->h_free.push_back(i);
    }
    if (!p->h_page || p->h_free.empty()) return nullptr;
    const uint16_t i = p->h_free.back();
    p->h_free.pop_back();
    return static_cast<char *>(p->h_page) + (size_t)i * kHostSlotBytes;
}

void host_slot_give(kmg_processor *p, void *slot)
{
    if (!slot) return;
    std::lock_guard<std::mutex> lock(p->mu);
    p->h_free.push_back((uint16_t)((static_cast<char *>(slot) - static_cast<char *>(p->h_page)) / kHostSlotBytes));
}

// The idle list is bounded: a block that would take it beyond kIdleMaxBlocks blocks or kIdleMaxBytes bytes pushes the OLDEST idle
// blocks out (hipFree), so a processor that meets images of ever growing size, or ever larger k, does not keep every block
// it once needed; and a hipMalloc that fails for lack of memory frees the whole list and tries once more.
constexpr size_t kIdleMaxBlocks = 24;
constexpr size_t kIdleMaxBytes = (size_t)3 << 30;

// smallest idle block that is large enough, else a fresh one
hipError_t block_take(kmg_processor *p, size_t bytes, void **ptr, size_t *cap)
{
    {
        std::lock_guard<std::mutex> lock(p->mu);
        size_t best = p->idle_arenas.size();
        // (not a block more than four times too large: a 256 MiB distance map must not end up as a 20 MiB workspace)
        const size_t too_large = 4u * bytes + ((size_t)1 << 20);
        for (size_t i = 0; i < p->idle_arenas.size(); ++i)
            if (p->idle_arenas[i].second >= bytes && p->idle_arenas[i].second <= too_large &&
                (best == p->idle_arenas.size() || p->idle_arenas[i].second < p->idle_arenas[best].second))
                best = i;
        if (best != p->idle_arenas.size()) {
            *ptr = p->idle_arenas[best].first;
            *cap = p->idle_arenas[best].second;
            p->idle_arenas.erase(p->idle_arenas.begin() + (long)best);
            p->n_block_reuse += 1;
            return hipSuccess;
        }
        p->n_block_malloc += 1;
    }
    hipError_t e = hipMalloc(ptr, bytes);
    if (e == hipErrorOutOfMemory) {
        // idle blocks may hold what this request needs: hand all of them back to the driver and try once more
        std::vector<std::pair<void *, size_t>> victims;
        {
            std::lock_guard<std::mutex> lock(p->mu);
            victims.swap(p->idle_arenas);
        }
        (void)hipGetLastError();
        for (auto &v : victims) (void)hipFree(v.first);
        if (!victims.empty()) e = hipMalloc(ptr, bytes);
    }
    if (e == hipSuccess) *cap = bytes;
    return e;
}

// the caller guarantees that no kernel still uses the block
void block_give(kmg_processor *p, void *ptr, size_t cap)
{
    if (!ptr) return;
    std::vector<void *> victims;
    {
        std::lock_guard<std::mutex> lock(p->mu);
        p->idle_arenas.emplace_back(ptr, cap);
        size_t total = 0;
        for (auto &a : p->idle_arenas) total += a.second;
        while (p->idle_arenas.size() > 1 && (p->idle_arenas.size() > kIdleMaxBlocks || total > kIdleMaxBytes)) {
            total -= p->idle_arenas.front().second;
            victims.push_back(p->idle_arenas.front().first);
            p->idle_arenas.erase(p->idle_arenas.begin());
        }
    }
    for (void *v : victims) (void)hipFree(v);                         // (outside the lock: hipFree synchronises the device)
}

extern "C" int kmg_processor_create(kmg_processor **out)
try { return kmg_processor_create_ex(nullptr, out); }
KMG_ABI_CATCH

extern "C" int kmg_processor_create_ex(const kmg_options *opt, kmg_processor **out)
try {
    if (!out) return fail(KMG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    kmg_options o;
    kmg_default_options(&o);
    if (opt) {
        // (the struct grew by `strategy` in round 6: a caller compiled against the previous header passes the old size)
        constexpr uint32_t kOldSize = (uint32_t)offsetof(kmg_options, strategy);
        if (opt->struct_size != sizeof(kmg_options) && opt->struct_size != kOldSize)
            return fail(KMG_ERR_INVALID_ARGUMENT, "kmg_options.struct_size mismatch");
        memcpy(&o, opt, opt->struct_size);
        o.struct_size = sizeof(kmg_options);
        if ((o.strategy & 3) == 3 || (o.strategy & ~7) != 0) return fail(KMG_ERR_INVALID_ARGUMENT, "kmg_options.strategy: unknown value %d", o.strategy);
        if (o.max_iterations == 0 || o.check_period == 0)
            return fail(KMG_ERR_INVALID_ARGUMENT, "max_iterations and check_period must be > 0");
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(KMG_ERR_NO_DEVICE, "no HIP device available (%s); libkmeans_hip has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    int dev = o.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= count) return fail(KMG_ERR_NO_DEVICE, "device %d out of range (%d devices)", dev, count);
    HIP_TRY(hipSetDevice(dev));
    kmg_processor *p = new (std::nothrow) kmg_processor();
    if (!p) return fail(KMG_ERR_OUT_OF_MEMORY, "host allocation failed");
    p->device = dev;
    p->opt = o;
    p->strategy.store(o.strategy);
    p->d_lut = nullptr;
    p->d_bounds = nullptr;
    p->d_sub_bounds = nullptr;
    p->d_lab_table = nullptr;
    p->d_sub_affine = nullptr;
    p->affine_failed = false;
    p->pool = nullptr;
    {
        // per-call scratch comes from a stream-ordered pool of the processor's own that keeps what has been freed instead
        // of handing it back to the driver at every synchronisation (with the default threshold of 0 each call of the
        // host-buffer API pays for fresh allocations again: find -m replace at 8192^2 0.65 -> 0.3 ms).  The device's
        // default pool is left alone: its settings belong to the host application.
        hipMemPoolProps props;
        memset(&props, 0, sizeof props);
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            p->pool = pool;
        } else {
            (void)hipGetLastError();      // no private pool on this runtime: per-call scratch falls back to the default pool
        }
    }
    float lut[256];
    build_srgb_lut100(lut);
    hipError_t e1 = hipMalloc((void **)&p->d_lut, 2 * sizeof lut);
    if (e1 == hipSuccess) e1 = hipMemcpy(p->d_lut, lut, sizeof lut, hipMemcpyHostToDevice);
    if (e1 == hipSuccess) e1 = launch_encode_thresholds(p->d_lut + 256, nullptr);
    if (e1 == hipSuccess) e1 = hipDeviceSynchronize();
    if (e1 != hipSuccess) {
        if (p->d_lut) (void)hipFree(p->d_lut);
        delete p;
        return fail(KMG_ERR_HIP, "processor setup failed: %s", hipGetErrorString(e1));
    }
    *out = p;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_host_alloc(size_t bytes, void **out)
try {
    if (!out || bytes == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad host_alloc arguments");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" void kmg_host_free(void *ptr)
try {
    if (ptr) (void)hipHostFree(ptr);
}
KMG_ABI_CATCH_VOID

extern "C" int kmg_debug_block_counts(kmg_processor *p, uint64_t out[2])
try {
    if (!p || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad block_counts arguments");
    std::lock_guard<std::mutex> lock(p->mu);
    out[0] = p->n_block_malloc;
    out[1] = p->n_block_reuse;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_debug_idle_blocks(kmg_processor *p, uint64_t out[2])
try {
    if (!p || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad idle_blocks arguments");
    std::lock_guard<std::mutex> lock(p->mu);
    out[0] = p->idle_arenas.size();
    out[1] = 0;
    for (auto &a : p->idle_arenas) out[1] += a.second;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_debug_encode_table_check(kmg_processor *p, uint64_t *mismatches)
try {
    if (!p || !mismatches) return fail(KMG_ERR_INVALID_ARGUMENT, "bad encode_table_check arguments");
    unsigned long long *d_bad = nullptr, bad = 0;
    hipError_t e = hipMalloc((void **)&d_bad, sizeof bad);
    if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof bad);
    if (e == hipSuccess) e = launch_encode_check(p->d_lut + 256, d_bad, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
    if (d_bad) (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(KMG_ERR_HIP, "encode table check failed: %s", hipGetErrorString(e));
    *mismatches = bad;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_debug_division_check(kmg_processor *p, float c, uint64_t out[3])
try {
    if (!p || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad division_check arguments");
    HIP_TRY(hipSetDevice(p->device));
    unsigned long long *d = nullptr, h[3] = {0ull, ~0ull, 0ull};
    hipError_t e = hipMalloc((void **)&d, sizeof h);
    if (e == hipSuccess) e = hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_division_check(c, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    if (e != hipSuccess) return fail(KMG_ERR_HIP, "division check failed: %s", hipGetErrorString(e));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_processor_set_strategy(kmg_processor *p, int strategy)
try {
    if (!p) return fail(KMG_ERR_INVALID_ARGUMENT, "processor is NULL");
    if ((strategy & 3) == 3 || (strategy & ~7) != 0) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown strategy %d", strategy);
    p->strategy.store(strategy);
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" void kmg_processor_destroy(kmg_processor *p)
try {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_lut) (void)hipFree(p->d_lut);
    if (p->d_bounds) (void)hipFree(p->d_bounds);
    if (p->d_sub_bounds) (void)hipFree(p->d_sub_bounds);
    if (p->d_lab_table) (void)hipFree(p->d_lab_table);
    if (p->d_sub_affine) (void)hipFree(p->d_sub_affine);
    for (hipStream_t st : p->idle_streams) (void)hipStreamDestroy(st);
    for (auto &a : p->idle_arenas) (void)hipFree(a.first);
    if (p->pool) (void)hipMemPoolDestroy(p->pool);
    if (p->h_page) (void)hipHostFree(p->h_page);
    delete p;
}
KMG_ABI_CATCH_VOID

// ---------------------------------------------------------------------------------------------
// host colour helpers
// ---------------------------------------------------------------------------------------------
extern "C" int kmg_palette_to_centroids(const uint8_t *palette_rgba, uint32_t n, float *c4)
try {
    if (!palette_rgba || !c4 || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad palette arguments");
    for (uint32_t i = 0; i < n; ++i) {
        crate_srgb8_to_lab(palette_rgba + 4 * i, c4 + 4 * i);
        c4[4 * i + 3] = 1.0f;
    }
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_centroids_to_palette(const float *c4, uint32_t k, uint8_t *out)
try {
    if (!c4 || !out || k == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad centroid arguments");
    for (uint32_t i = 0; i < k; ++i) {
        crate_lab_to_srgb8(c4 + 4 * i, out + 4 * i);
        out[4 * i + 3] = 255;
    }
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_dither_threshold(const float *c4, uint32_t k, float *thr)
try {
    if (!c4 || !thr || k < 2) return fail(KMG_ERR_INVALID_ARGUMENT, "dither threshold needs k >= 2");
    *thr = dither_threshold(c4, k);
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" void kmg_resized_dims(uint32_t w, uint32_t h, uint32_t max_size, uint32_t *nw, uint32_t *nh)
try {
    // structures.rs:79-89
    uint32_t a, b;
    if (w > h) {
        uint32_t v = (uint32_t)((float)h * (float)max_size / (float)w);
        a = max_size; b = v > 1 ? v : 1;
    } else {
        uint32_t v = (uint32_t)((float)w * (float)max_size / (float)h);
        a = v > 1 ? v : 1; b = max_size;
    }
    if (nw) *nw = a;
    if (nh) *nh = b;
}
KMG_ABI_CATCH_VOID

// ---------------------------------------------------------------------------------------------
// device-pointer API: conversions that need no kmg_lloyd
// ---------------------------------------------------------------------------------------------
extern "C" int kmg_dev_rgb_to_lab(kmg_processor *p, const uint8_t *d_rgba, uint64_t n, float *d_lab3, void *stream)
try {
    if (!p || !d_rgba || !d_lab3 || n == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad rgb_to_lab arguments");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_rgb_to_lab((const uint32_t *)d_rgba, n, p->d_lut, d_lab3, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_dev_resize(kmg_processor *p, const uint8_t *d_rgba, uint32_t w, uint32_t h, uint32_t nw,
                              uint32_t nh, uint8_t *d_out, void *stream)
try {
    if (!p || !d_rgba || !d_out || !w || !h || !nw || !nh) return fail(KMG_ERR_INVALID_ARGUMENT, "bad resize arguments");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_resize((const uint32_t *)d_rgba, w, h, nw, nh, (uint32_t *)d_out, S(stream)));
    return KMG_OK;
}
KMG_ABI_CATCH

