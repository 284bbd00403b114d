// Exercises kmeans-gpu_amd/host/kmeans_color_gpu.hpp (the C++ mirror of the reference crate's public
// API, core/src/lib.rs:24-165) the way cli/src/main.rs:46-125 uses the crate.
//   check_host_api nogpu                          expects ImageProcessor::create() to fail loudly
//   check_host_api run in.rgba w h out.bin        palette / find / reduce of a raw RGBA8 image; results to out.bin
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "kmeans_color_gpu.hpp"

using namespace kmeans_color_gpu;

static void put(std::ofstream &f, const void *p, size_t n) { f.write(reinterpret_cast<const char *>(p), (std::streamsize)n); }

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "nogpu")) {
        try {
            ImageProcessor p = ImageProcessor::create();
            std::cout << "created\n";                        // a device exists: not the case this mode checks
            return 0;
        } catch (const Error &e) {
            std::cout << "error " << e.status << " " << e.what() << "\n";
            return (e.status != KMG_OK && strlen(e.what()) > 0) ? 0 : 1;
        }
    }
    if (argc != 6 || strcmp(argv[1], "run")) { fprintf(stderr, "usage\n"); return 2; }
    const uint32_t w = (uint32_t)atoi(argv[3]), h = (uint32_t)atoi(argv[4]);
    std::ifstream in(argv[2], std::ios::binary);
    std::vector<uint8_t> raw((size_t)w * h * 4);
    in.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)raw.size());
    if (!in) { fprintf(stderr, "short input\n"); return 2; }
    const Image image = copied_pixel({w, h}, raw.data());
    if (!(image.get_pixel(w - 1, h - 1) == image.rgba.back()) || image.dimensions() != std::make_pair(w, h)) return 3;

    ImageProcessor processor = ImageProcessor::create();
    std::ofstream out(argv[5], std::ios::binary);
    // palette -c 8 (kmeans and octree), lib.rs:67-77
    for (Algorithm algo : {Algorithm::Kmeans, Algorithm::Octree}) {
        const std::vector<RGBA8> pal = processor.palette(8, image, algo);
        const uint32_t n = (uint32_t)pal.size();
        put(out, &n, 4);
        put(out, pal.data(), pal.size() * 4);
    }
    // find with a fixed palette in the three modes, lib.rs:79-114
    const std::vector<RGBA8> colors = {{0, 0, 0, 255}, {255, 255, 255, 255}, {200, 30, 30, 255}, {30, 60, 200, 255}};
    for (ReduceMode mode : {ReduceMode::Replace, ReduceMode::Dither, ReduceMode::Meld}) {
        const Image r = processor.find(image, colors, mode);
        put(out, r.rgba.data(), r.rgba.size() * 4);
    }
    // reduce -c 8, lib.rs:116-164
    {
        const Image r = processor.reduce(8, image, Algorithm::Kmeans, ReduceMode::Dither);
        const std::vector<uint8_t> bytes = r.into_raw_pixels();
        put(out, bytes.data(), bytes.size());
        const Image o = processor.reduce(8, image, Algorithm::Octree, ReduceMode::Replace);
        put(out, o.rgba.data(), o.rgba.size() * 4);
    }
    // the same object over a device list (kmg_group_*; two ranks sharing device 0 through the loopback exchange): same bytes
    {
        ImageProcessor many = ImageProcessor::create_on({0, 0}, nullptr, KMG_GROUP_LOOPBACK);
        const Image one = processor.reduce(8, image, Algorithm::Kmeans, ReduceMode::Dither);
        const Image two = many.reduce(8, image, Algorithm::Kmeans, ReduceMode::Dither);
        if (!(one.rgba == two.rgba) || !(processor.palette(8, image, Algorithm::Kmeans) == many.palette(8, image, Algorithm::Kmeans))) return 5;
        const std::vector<Image> batch = many.reduce_batch(5, {image, image}, Algorithm::Kmeans, ReduceMode::Replace);
        if (batch.size() != 2 || !(batch[0].rgba == processor.reduce(5, image, Algorithm::Kmeans, ReduceMode::Replace).rgba) ||
            !(batch[0].rgba == batch[1].rgba)) return 6;
        std::cout << "device list ok\n";
    }
    // error behaviour: anyhow::Err -> exception with the library's message
    int errors = 0;
    try { processor.reduce(0, image, Algorithm::Kmeans, ReduceMode::Replace); } catch (const Error &e) { errors += e.status == KMG_ERR_INVALID_ARGUMENT; }
    try { processor.find(image, {}, ReduceMode::Replace); } catch (const Error &e) { errors += e.status == KMG_ERR_INVALID_ARGUMENT; }
    try { processor.palette(8, Image({0, 0}, {}), Algorithm::Kmeans); } catch (const Error &e) { errors += e.status == KMG_ERR_INVALID_ARGUMENT; }
    std::cout << "errors " << errors << "\n";
    std::cout << to_string(Algorithm::Octree) << " " << to_string(ReduceMode::Meld) << "\n";
    return errors == 3 ? 0 : 4;
}
