// kmeans_color_gpu.hpp -- C++ host-side mirror of the reference crate's public API
// (core/src/lib.rs:24-165, core/src/image.rs) on top of the C ABI of include/kmeans_hip.h.
//
// Same names, argument meaning and error behaviour as the Rust crate `kmeans_color_gpu`:
//   ImageProcessor::new()                      -> ImageProcessor::create()   (throws on failure)
//   processor.palette(color_count, &image, algo)           -> std::vector<RGBA8>
//   processor.find(&image, &colors, &reduce_mode)          -> Image
//   processor.reduce(color_count, &image, &algo, &mode)    -> Image
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

    ImageProcessor(ImageProcessor &&o) noexcept : p_(o.p_) { o.p_ = nullptr; }
    ImageProcessor &operator=(ImageProcessor &&o) noexcept
    {
        if (this != &o) { kmg_processor_destroy(p_); p_ = o.p_; o.p_ = nullptr; }
        return *this;
    }
    ImageProcessor(const ImageProcessor &) = delete;
    ImageProcessor &operator=(const ImageProcessor &) = delete;
    ~ImageProcessor() { kmg_processor_destroy(p_); }

    // lib.rs:67-77
    std::vector<RGBA8> palette(uint32_t color_count, const Image &image, Algorithm algo) const
    {
        std::vector<RGBA8> out(color_count ? color_count : 1);
        uint32_t n = 0;
        check(kmg_palette(p_, bytes(image), image.dims.first, image.dims.second, color_count, (int)algo,
                          reinterpret_cast<uint8_t *>(out.data()), &n));
        out.resize(n);
        return out;
    }

    // lib.rs:79-114
    Image find(const Image &image, const std::vector<RGBA8> &colors, ReduceMode reduce_mode) const
    {
        Image out(image.dims, std::vector<RGBA8>(image.rgba.size()));
        check(kmg_find(p_, bytes(image), image.dims.first, image.dims.second,
                       reinterpret_cast<const uint8_t *>(colors.data()), (uint32_t)colors.size(), (int)reduce_mode,
                       reinterpret_cast<uint8_t *>(out.rgba.data())));
        return out;
    }

    // lib.rs:116-164
    Image reduce(uint32_t color_count, const Image &image, Algorithm algo, ReduceMode reduce_mode) const
    {
        Image out(image.dims, std::vector<RGBA8>(image.rgba.size()));
        check(kmg_reduce(p_, bytes(image), image.dims.first, image.dims.second, color_count, (int)algo,
                         (int)reduce_mode, reinterpret_cast<uint8_t *>(out.rgba.data())));
        return out;
    }

    kmg_processor *handle() const { return p_; }

private:
    explicit ImageProcessor(const kmg_options *opt) : p_(nullptr)
    {
        check(opt ? kmg_processor_create_ex(opt, &p_) : kmg_processor_create(&p_));
    }
    static const uint8_t *bytes(const Image &im) { return reinterpret_cast<const uint8_t *>(im.rgba.data()); }
    static void check(int rc)
    {
        if (rc != KMG_OK) throw Error(rc, kmg_last_error());
    }
    kmg_processor *p_;
};

}  // namespace kmeans_color_gpu
