// kmeans_color_gpu.hpp -- C++ host-side mirror of the reference crate's public API
// (core/src/lib.rs:24-165, core/src/image.rs) on top of the C ABI of include/kmeans_hip.h.
//
// Same names, argument meaning and error behaviour as the Rust crate `kmeans_color_gpu`:
//   ImageProcessor::new()                      -> ImageProcessor::create()   (throws on failure)
//   processor.palette(color_count, &image, algo)           -> std::vector<RGBA8>
//   processor.find(&image, &colors, &reduce_mode)          -> Image
//   processor.reduce(color_count, &image, &algo, &mode)    -> Image
// and, with no counterpart in the single-device reference (lib.rs:38-65 picks one adapter):
//   ImageProcessor::create_on({0, 1, ..})      -> the same object over a device LIST (kmg_group_*: one processor + RCCL rank per
//                                                 device; palette / find / reduce tile the image in row bands, same bytes)
//   processor.reduce_batch(color_count, images, algo, mode) -> whole images per device, side by side
// `anyhow::Result` errors become kmeans_color_gpu::Error exceptions carrying the kmg_status and the
// library's message.  Header only; link with -lkmeans_hip.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kmeans_hip.h"

namespace kmeans_color_gpu {

struct RGBA8 {                      // rgb::RGBA8 re-export (lib.rs:3)
    uint8_t r, g, b, a;
    bool operator==(const RGBA8 &o) const { return r == o.r && g == o.g && b == o.b && a == o.a; }
};
static_assert(sizeof(RGBA8) == 4, "RGBA8 must be 4 tightly packed bytes");

enum class Algorithm { Kmeans = KMG_ALGO_KMEANS, Octree = KMG_ALGO_OCTREE };                  // lib.rs:215-219
enum class ReduceMode { Replace = KMG_MODE_REPLACE, Dither = KMG_MODE_DITHER, Meld = KMG_MODE_MELD };  // lib.rs:234-239

inline const char *to_string(Algorithm a) { return a == Algorithm::Kmeans ? "kmeans" : "octree"; }   // lib.rs:221-232
inline const char *to_string(ReduceMode m)                                                           // lib.rs:241-253
{
    return m == ReduceMode::Replace ? "replace" : m == ReduceMode::Dither ? "dither" : "meld";
}

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &msg) : std::runtime_error(msg), status(s) {}
};

// image.rs:20-48: tightly packed row-major RGBA8, dimensions = (width, height)
struct Image {
    std::pair<uint32_t, uint32_t> dims;
    std::vector<RGBA8> rgba;

    Image(std::pair<uint32_t, uint32_t> dimensions, std::vector<RGBA8> pixels)
        : dims(dimensions), rgba(std::move(pixels)) {}
    const RGBA8 &get_pixel(uint32_t x, uint32_t y) const { return rgba[(size_t)x + (size_t)y * dims.first]; }
    std::pair<uint32_t, uint32_t> dimensions() const { return dims; }
    std::vector<uint8_t> into_raw_pixels() const
    {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(rgba.data());
        return std::vector<uint8_t>(p, p + rgba.size() * 4);
    }
};

// image.rs:50-64 copied_pixel
inline Image copied_pixel(std::pair<uint32_t, uint32_t> dimensions, const uint8_t *rgba_bytes)
{
    const RGBA8 *p = reinterpret_cast<const RGBA8 *>(rgba_bytes);
    return Image(dimensions, std::vector<RGBA8>(p, p + (size_t)dimensions.first * dimensions.second));
}

class ImageProcessor {
public:
    // lib.rs:38-65
    static ImageProcessor create() { return ImageProcessor(nullptr); }
    static ImageProcessor create(const kmg_options &opt) { return ImageProcessor(&opt); }
    // the same constructor over several devices of the node (HIP ordinals; empty = every visible device)
    static ImageProcessor create_on(const std::vector<int> &devices, const kmg_options *opt = nullptr, uint32_t flags = 0)
    {
        kmg_group_options go;
        kmg_default_group_options(&go);
        if (devices.size() > KMG_MAX_DEVICES) throw Error(KMG_ERR_INVALID_ARGUMENT, "too many devices");
        go.n_devices = (uint32_t)devices.size();
        for (size_t i = 0; i < devices.size(); ++i) go.devices[i] = devices[i];
        go.flags = flags;
        if (opt) go.processor = *opt;
        ImageProcessor p;
        check(kmg_group_create(&go, &p.g_));
        return p;
    }

    ImageProcessor(ImageProcessor &&o) noexcept : p_(o.p_), g_(o.g_) { o.p_ = nullptr; o.g_ = nullptr; }
    ImageProcessor &operator=(ImageProcessor &&o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; g_ = o.g_; o.p_ = nullptr; o.g_ = nullptr; }
        return *this;
    }
    ImageProcessor(const ImageProcessor &) = delete;
    ImageProcessor &operator=(const ImageProcessor &) = delete;
    ~ImageProcessor() { release(); }

    // lib.rs:67-77
    std::vector<RGBA8> palette(uint32_t color_count, const Image &image, Algorithm algo) const
    {
        std::vector<RGBA8> out(color_count ? color_count : 1);
        uint32_t n = 0;
        uint8_t *dst = reinterpret_cast<uint8_t *>(out.data());
        check(g_ ? kmg_group_palette(g_, bytes(image), image.dims.first, image.dims.second, color_count, (int)algo, dst, &n)
                 : kmg_palette(p_, bytes(image), image.dims.first, image.dims.second, color_count, (int)algo, dst, &n));
        out.resize(n);
        return out;
    }

    // lib.rs:79-114
    Image find(const Image &image, const std::vector<RGBA8> &colors, ReduceMode reduce_mode) const
    {
        Image out(image.dims, std::vector<RGBA8>(image.rgba.size()));
        const uint8_t *pal = reinterpret_cast<const uint8_t *>(colors.data());
        uint8_t *dst = reinterpret_cast<uint8_t *>(out.rgba.data());
        check(g_ ? kmg_group_find(g_, bytes(image), image.dims.first, image.dims.second, pal, (uint32_t)colors.size(), (int)reduce_mode, dst)
                 : kmg_find(p_, bytes(image), image.dims.first, image.dims.second, pal, (uint32_t)colors.size(), (int)reduce_mode, dst));
        return out;
    }

    // lib.rs:116-164
    Image reduce(uint32_t color_count, const Image &image, Algorithm algo, ReduceMode reduce_mode) const
    {
        Image out(image.dims, std::vector<RGBA8>(image.rgba.size()));
        uint8_t *dst = reinterpret_cast<uint8_t *>(out.rgba.data());
        check(g_ ? kmg_group_reduce(g_, bytes(image), image.dims.first, image.dims.second, color_count, (int)algo, (int)reduce_mode, dst)
                 : kmg_reduce(p_, bytes(image), image.dims.first, image.dims.second, color_count, (int)algo, (int)reduce_mode, dst));
        return out;
    }

    // a batch: whole images per device (a single-device processor takes them one after the other)
    std::vector<Image> reduce_batch(uint32_t color_count, const std::vector<Image> &images, Algorithm algo, ReduceMode reduce_mode) const
    {
        std::vector<Image> out;
        for (const Image &im : images) out.emplace_back(im.dims, std::vector<RGBA8>(im.rgba.size()));
        if (!g_) {
            for (size_t i = 0; i < images.size(); ++i) out[i] = reduce(color_count, images[i], algo, reduce_mode);
            return out;
        }
        std::vector<const uint8_t *> src;
        std::vector<uint8_t *> dst;
        std::vector<uint32_t> ws, hs;
        for (size_t i = 0; i < images.size(); ++i) {
            src.push_back(bytes(images[i])); dst.push_back(reinterpret_cast<uint8_t *>(out[i].rgba.data()));
            ws.push_back(images[i].dims.first); hs.push_back(images[i].dims.second);
        }
        check(kmg_group_reduce_batch(g_, (uint32_t)images.size(), src.data(), ws.data(), hs.data(), color_count, (int)algo, (int)reduce_mode,
                                     dst.data()));
        return out;
    }

    kmg_processor *handle() const { return g_ ? kmg_group_processor(g_, 0) : p_; }
    kmg_group *group() const { return g_; }

private:
    ImageProcessor() : p_(nullptr), g_(nullptr) {}
    explicit ImageProcessor(const kmg_options *opt) : p_(nullptr), g_(nullptr)
    {
        check(opt ? kmg_processor_create_ex(opt, &p_) : kmg_processor_create(&p_));
    }
    void release()
    {
        if (g_) kmg_group_destroy(g_);
        kmg_processor_destroy(p_);
        g_ = nullptr; p_ = nullptr;
    }
    static const uint8_t *bytes(const Image &im) { return reinterpret_cast<const uint8_t *>(im.rgba.data()); }
    static void check(int rc)
    {
        if (rc != KMG_OK) throw Error(rc, kmg_last_error());
    }
    kmg_processor *p_;
    kmg_group *g_;
};

}  // namespace kmeans_color_gpu
