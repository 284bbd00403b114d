// kmg_api.hip -- the host-buffer calls of the C ABI (include/kmeans_hip.h): ImageProcessor::{palette, find, reduce}
// (core/src/lib.rs:67-164) -- argument checking, upload, the host sequencing of operations.rs on top of kmg_lloyd_* / kmg_dev_apply,
// download.  There is no CPU data path in this file: every per-pixel step is a kernel launch.

#include "kmg_state.h"

// ---------------------------------------------------------------------------------------------
// host-buffer API (ImageProcessor::{palette, find, reduce})
// ---------------------------------------------------------------------------------------------
namespace {

struct LloydGuard {
    kmg_lloyd *s = nullptr;
    ~LloydGuard() { kmg_lloyd_destroy(s); }
};

int check_image(const kmg_processor *p, const uint8_t *rgba, uint32_t w, uint32_t h)
{
    if (!p) return fail(KMG_ERR_INVALID_ARGUMENT, "processor is NULL");
    if (!rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "image pointer is NULL");
    if (w == 0 || h == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "image has zero width or height");
    if ((uint64_t)w * h > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image has more than 2^32-1 pixels");
    return KMG_OK;
}

// operations.rs:15-88 extract_palette_kmeans on a device-resident image -> host centroid table
int extract_palette_kmeans(kmg_processor *p, const uint8_t *d_rgba, uint32_t w, uint32_t h, uint32_t k,
                           hipStream_t st, float *c4)
{
    int rc;
    const uint8_t *src = d_rgba;
    uint32_t sw = w, sh = h;
    StreamBuf small;
    const uint32_t m = p->opt.shrink_max_dim;
    if (m && (w > m || h > m)) {                                       // structures.rs:67-74
        kmg_resized_dims(w, h, m, &sw, &sh);
        HIP_TRY(small.alloc(p, (size_t)sw * sh * 4, st));
        if ((rc = kmg_dev_resize(p, d_rgba, w, h, sw, sh, (uint8_t *)small.ptr, st)) != KMG_OK) return rc;
        src = (const uint8_t *)small.ptr;
    }
    LloydGuard g;
    if ((rc = lloyd_create_impl(p, k, &g.s, st)) != KMG_OK) return rc;
    if ((rc = kmg_lloyd_init_centroids(g.s, src, sw, sh, st)) != KMG_OK) return rc;   // operations.rs:73
    if (log_debug()) {
        std::vector<float> c(4 * k);
        if (kmg_lloyd_get_centroids(g.s, c.data(), st) == KMG_OK) {
            fprintf(stderr, "[kmeans_hip] == Initial centroids: ==\n");
            for (uint32_t i = 0; i < k; ++i)
                fprintf(stderr, "[kmeans_hip] Centroid %u = [%g, %g, %g, %g]\n", i, c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]);
        }
    }
    uint32_t it = 0;
    if ((rc = kmg_lloyd_run(g.s, src, (uint64_t)sw * sh, nullptr, &it, st)) != KMG_OK) return rc;  // operations.rs:85
    if ((rc = kmg_lloyd_get_centroids(g.s, c4, st)) != KMG_OK) return rc;
    if (log_debug()) {
        fprintf(stderr, "[kmeans_hip] == Final centroids at iteration %u: ==\n", it);
        for (uint32_t i = 0; i < k; ++i)
            fprintf(stderr, "[kmeans_hip] Centroid %u = [%g, %g, %g, %g]\n", i, c4[4 * i], c4[4 * i + 1], c4[4 * i + 2], c4[4 * i + 3]);
    }
    return KMG_OK;
}

// An image between a caller's (pageable) buffer and the device.  The calls that use this return when the work is done, so a large
// image goes through a synchronous copy once `st` has drained: the runtime pipelines it through pinned staging at PCIe
// rate (256 MiB: 4.7 ms each way on the MI355X box between touched buffers), where hipMemcpyAsync of pageable memory takes
// 16-20 ms (tools/host_copy_probe.py, tools/reduce_host_probe.py).  A download into a result buffer whose pages do not exist yet
// still takes 16-40 ms, the caller's page faults; asking for those pages ahead (MADV_POPULATE_WRITE on helper threads during
// the GPU work) was measured and is not in: -6 ms per call in a fresh process, +8 ms in a long-running one.
// Round 4: the copy is ordered on the call's own stream (hipMemcpyWithStream), never on the legacy null stream -- that one
// synchronises with every blocking stream of the host application and serialises the calls this API lets run concurrently on
// one processor (examples/parallel.rs).  An image of 32 MiB or more is cut into kCopyParts row ranges copied by as many host
// threads, each on a stream of its own from the processor's idle list: the staging copies and -- for a result buffer whose
// pages do not exist yet -- the page faults of the ranges then proceed side by side (8192^2 download into fresh pages:
// 26-40 ms as one copy).
constexpr size_t kCopyParts = 4;

}  // namespace

hipError_t kmg::copy_host_image(kmg_processor *p, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st)
{
    if (bytes < ((size_t)1 << 20)) return hipMemcpyAsync(dst, src, bytes, kind, st);
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if (bytes < ((size_t)32 << 20)) return hipMemcpyWithStream(dst, src, bytes, kind, st);
    StreamGuard extra[kCopyParts - 1];
    hipStream_t streams[kCopyParts] = {st};
    for (size_t i = 1; i < kCopyParts; ++i) {
        if ((e = extra[i - 1].acquire(p)) != hipSuccess) return e;
        streams[i] = extra[i - 1].st;
    }
    const size_t part = ((bytes / kCopyParts) + 4095u) & ~(size_t)4095u;
    hipError_t results[kCopyParts];
    std::thread workers[kCopyParts - 1];
    const int device = p->device;
    auto copy_part = [&](size_t i) {
        const size_t off = i * part;
        if (off >= bytes) { results[i] = hipSuccess; return; }
        const size_t nb = std::min(part, bytes - off);
        hipError_t r = i ? hipSetDevice(device) : hipSuccess;          // (a fresh thread has no current device)
        if (r == hipSuccess) r = hipMemcpyWithStream((uint8_t *)dst + off, (const uint8_t *)src + off, nb, kind, streams[i]);
        results[i] = r;
    };
    // (a std::thread that cannot start throws std::system_error: the part is then copied on this thread, and the workers
    // already running are joined whatever happens -- a joinable std::thread that goes out of scope ends the process)
    for (size_t i = 1; i < kCopyParts; ++i) {
        results[i] = hipErrorUnknown;
        try { workers[i - 1] = std::thread(copy_part, i); } catch (const std::exception &) { copy_part(i); }
    }
    copy_part(0);
    for (size_t i = 1; i < kCopyParts; ++i)
        if (workers[i - 1].joinable()) workers[i - 1].join();
    for (size_t i = 0; i < kCopyParts; ++i)
        if (results[i] != hipSuccess) return results[i];
    return hipSuccess;
}

namespace {

int upload_image(kmg_processor *p, const uint8_t *rgba, uint32_t w, uint32_t h, hipStream_t st, StreamBuf &buf)
{
    const size_t bytes = (size_t)w * h * 4;
    HIP_TRY(buf.alloc(p, bytes, st));
    HIP_TRY(copy_host_image(p, buf.ptr, rgba, bytes, hipMemcpyHostToDevice, st));   // structures.rs:31-65
    return KMG_OK;
}

// find_colors / dither_colors + OutputTexture::pull_image (structures.rs:441-470)
int apply_and_download(kmg_processor *p, const uint8_t *d_rgba, uint32_t w, uint32_t h, const float *c4,
                       uint32_t k, int mode, hipStream_t st, uint8_t *out_rgba)
{
    int rc;
    StreamBuf out;
    const size_t bytes = (size_t)w * h * 4;
    HIP_TRY(out.alloc(p, bytes, st));
    if ((rc = kmg_dev_apply(p, d_rgba, w, h, 0, c4, k, mode, (uint8_t *)out.ptr, st)) != KMG_OK) return rc;
    HIP_TRY(copy_host_image(p, out_rgba, out.ptr, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return KMG_OK;
}

// octree_palette (lib.rs:288-331) on a device-resident image: shrink to <= 128 on the device, pull
// the <= 128x128 image, run the reference's CPU octree on it, sort ascending by palette-crate Lab L
int octree_palette_of(kmg_processor *p, const uint8_t *d_rgba, uint32_t w, uint32_t h, uint32_t color_count,
                      hipStream_t st, std::vector<std::array<uint8_t, 4>> &colors)
{
    const uint32_t MAX_SIZE = 128;                                     // lib.rs:293
    int rc;
    uint32_t sw = w, sh = h;
    const uint8_t *src = d_rgba;
    StreamBuf small;
    if (w > MAX_SIZE || h > MAX_SIZE) {
        kmg_resized_dims(w, h, MAX_SIZE, &sw, &sh);
        HIP_TRY(small.alloc(p, (size_t)sw * sh * 4, st));
        if ((rc = kmg_dev_resize(p, d_rgba, w, h, sw, sh, (uint8_t *)small.ptr, st)) != KMG_OK) return rc;
        src = (const uint8_t *)small.ptr;
    }
    std::vector<uint8_t> host((size_t)sw * sh * 4);
    HIP_TRY(hipMemcpyAsync(host.data(), src, host.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    colors = octree_sorted_palette(host.data(), (uint64_t)sw * sh, color_count);
    return KMG_OK;
}

}  // namespace

std::vector<std::array<uint8_t, 4>> kmg::octree_sorted_palette(const uint8_t *host_rgba, uint64_t n_pixels, uint32_t color_count)
{
    std::vector<std::array<uint8_t, 4>> colors = octree_palette(host_rgba, n_pixels, color_count);   // operations.rs:90-97
    std::vector<float> L(colors.size());
    for (size_t i = 0; i < colors.size(); ++i) {
        float lab[3];
        crate_srgb8_to_lab(colors[i].data(), lab);
        L[i] = lab[0];
    }
    std::vector<size_t> order(colors.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return L[a] < L[b]; });   // lib.rs:320-328
    std::vector<std::array<uint8_t, 4>> sorted(colors.size());
    for (size_t i = 0; i < order.size(); ++i) sorted[i] = colors[order[i]];
    return sorted;
}

// lib.rs:255-286: pull_values (palette-crate Lab -> sRGB8) then sort ascending by Lab L
void kmg::sorted_palette_of(const float *c4, uint32_t color_count, uint8_t *out_rgba)
{
    struct Entry { float L; uint8_t px[4]; };
    std::vector<Entry> e(color_count);
    for (uint32_t i = 0; i < color_count; ++i) {
        uint8_t rgb[3];
        float lab[3];
        crate_lab_to_srgb8(&c4[4 * i], rgb);
        e[i].px[0] = rgb[0]; e[i].px[1] = rgb[1]; e[i].px[2] = rgb[2]; e[i].px[3] = 255;
        crate_srgb8_to_lab(rgb, lab);
        e[i].L = lab[0];
    }
    std::stable_sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) { return a.L < b.L; });
    for (uint32_t i = 0; i < color_count; ++i) memcpy(out_rgba + 4 * i, e[i].px, 4);
}

// ColorTree::{add_color, reduce} (core/src/octree.rs): host helper, needs no device
extern "C" int kmg_octree_palette(const uint8_t *rgba, uint64_t n_pixels, uint32_t color_count, uint8_t *out_rgba,
                                  uint32_t *out_count)
try {
    if (!rgba || !out_rgba || !out_count || n_pixels == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "bad octree arguments");
    const std::vector<std::array<uint8_t, 4>> c = octree_palette(rgba, n_pixels, color_count);
    for (size_t i = 0; i < c.size(); ++i) memcpy(out_rgba + 4 * i, c[i].data(), 4);
    *out_count = (uint32_t)c.size();
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_find(kmg_processor *p, const uint8_t *rgba, uint32_t w, uint32_t h, const uint8_t *palette_rgba,
                        uint32_t n_colors, int mode, uint8_t *out_rgba)
try {
    int rc;
    if ((rc = check_image(p, rgba, w, h)) != KMG_OK) return rc;
    if (!palette_rgba || n_colors == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "palette is empty");
    if (!out_rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "output pointer is NULL");
    HIP_TRY(hipSetDevice(p->device));
    StreamGuard sg;
    HIP_TRY(sg.acquire(p));
    std::vector<float> c4(4 * (size_t)n_colors);
    if ((rc = kmg_palette_to_centroids(palette_rgba, n_colors, c4.data())) != KMG_OK) return rc;  // lib.rs:86-87
    StreamBuf img;
    if ((rc = upload_image(p, rgba, w, h, sg.st, img)) != KMG_OK) return rc;
    return apply_and_download(p, (const uint8_t *)img.ptr, w, h, c4.data(), n_colors, mode, sg.st, out_rgba);
}
KMG_ABI_CATCH

extern "C" int kmg_reduce(kmg_processor *p, const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t color_count,
                          int algo, int mode, uint8_t *out_rgba)
try {
    int rc;
    if ((rc = check_image(p, rgba, w, h)) != KMG_OK) return rc;
    if (color_count == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "k must be an integer higher than 0");
    if (!out_rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "output pointer is NULL");
    if (algo != KMG_ALGO_KMEANS && algo != KMG_ALGO_OCTREE) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown algorithm %d", algo);
    if (mode < KMG_MODE_REPLACE || mode > KMG_MODE_MELD) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    if (algo == KMG_ALGO_KMEANS && color_count > KMG_MAX_K)
        return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", color_count, KMG_MAX_K);
    HIP_TRY(hipSetDevice(p->device));
    StreamGuard sg;
    HIP_TRY(sg.acquire(p));
    StreamBuf img;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = upload_image(p, rgba, w, h, sg.st, img)) != KMG_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    if (algo == KMG_ALGO_OCTREE) {                                     // lib.rs:133-136
        std::vector<std::array<uint8_t, 4>> colors;
        if ((rc = octree_palette_of(p, (const uint8_t *)img.ptr, w, h, color_count, sg.st, colors)) != KMG_OK) return rc;
        if (colors.empty()) return fail(KMG_ERR_INVALID_ARGUMENT, "the octree returned no colour");
        if (colors.size() > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "the octree returned %zu colours, more than KMG_MAX_K = %u", colors.size(), KMG_MAX_K);
        std::vector<float> oc4(4 * colors.size());
        if ((rc = kmg_palette_to_centroids(colors[0].data(), (uint32_t)colors.size(), oc4.data())) != KMG_OK) return rc;
        return apply_and_download(p, (const uint8_t *)img.ptr, w, h, oc4.data(), (uint32_t)colors.size(), mode, sg.st, out_rgba);
    }
    std::vector<float> c4(4 * (size_t)color_count);
    if ((rc = extract_palette_kmeans(p, (const uint8_t *)img.ptr, w, h, color_count, sg.st, c4.data())) != KMG_OK) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    rc = apply_and_download(p, (const uint8_t *)img.ptr, w, h, c4.data(), color_count, mode, sg.st, out_rgba);
    if (log_debug()) {
        const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        fprintf(stderr, "[kmeans_hip] reduce %ux%u k=%u: upload %.2f ms, palette %.2f ms, output pass + download %.2f ms\n", w, h,
                color_count, ms(t0, t1), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
    }
    return rc;
}
KMG_ABI_CATCH

extern "C" int kmg_palette(kmg_processor *p, const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t color_count,
                           int algo, uint8_t *out_rgba, uint32_t *out_count)
try {
    int rc;
    if ((rc = check_image(p, rgba, w, h)) != KMG_OK) return rc;
    if (color_count == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "k must be an integer higher than 0");
    if (!out_rgba || !out_count) return fail(KMG_ERR_INVALID_ARGUMENT, "output pointer is NULL");
    if (algo != KMG_ALGO_KMEANS && algo != KMG_ALGO_OCTREE) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown algorithm %d", algo);
    if (algo == KMG_ALGO_KMEANS && color_count > KMG_MAX_K)
        return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", color_count, KMG_MAX_K);
    HIP_TRY(hipSetDevice(p->device));
    StreamGuard sg;
    HIP_TRY(sg.acquire(p));
    StreamBuf img;
    if ((rc = upload_image(p, rgba, w, h, sg.st, img)) != KMG_OK) return rc;
    if (algo == KMG_ALGO_OCTREE) {                                     // lib.rs:288-331
        std::vector<std::array<uint8_t, 4>> colors;
        if ((rc = octree_palette_of(p, (const uint8_t *)img.ptr, w, h, color_count, sg.st, colors)) != KMG_OK) return rc;
        for (size_t i = 0; i < colors.size(); ++i) memcpy(out_rgba + 4 * i, colors[i].data(), 4);
        *out_count = (uint32_t)colors.size();
        return KMG_OK;
    }
    std::vector<float> c4(4 * (size_t)color_count);
    if ((rc = extract_palette_kmeans(p, (const uint8_t *)img.ptr, w, h, color_count, sg.st, c4.data())) != KMG_OK) return rc;
    sorted_palette_of(c4.data(), color_count, out_rgba);
    *out_count = color_count;
    return KMG_OK;
}
KMG_ABI_CATCH

