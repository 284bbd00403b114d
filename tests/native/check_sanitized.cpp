// check_sanitized.cpp -- the host-only code of the project under -fsanitize=address,undefined ("CPU oracle / host
// code under sanitizers"; GPU sanitizers are not available on the pool).  Built by `make -C oracle asan` from this file +
// oracle/kmg_oracle.c + the product's host headers (csrc/kmg_octree.h, kmg_color.h, kmg_math.h), run by
// tests/test_sanitizers.py.  Every call the golden tests make is made here on small inputs (sizes chosen to hit the ragged ends:
// widths that are not multiples of 4 or 16, k = 1, k > pixels, images smaller than the shrink limit); any out-of-bounds access,
// use of uninitialised padding through memcmp, signed overflow or misaligned access ends the run (-fno-sanitize-recover).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kmg_color.h"
#include "kmg_math.h"
#include "kmg_octree.h"
#include "kmg_oracle.h"

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void octree_both(const std::vector<uint8_t> &px, uint32_t k)
{
    const uint64_t n = px.size() / 4;
    std::vector<uint8_t> want(4 * (size_t)(k > n ? n : k) + 4);
    const uint32_t nw = orc_octree_palette(px.data(), n, k, want.data());
    const auto got = kmg::octree_palette(px.data(), n, k);           // the product's (csrc/kmg_octree.h)
    CHECK(got.size() == nw);
    for (size_t i = 0; i < got.size() && i < nw; ++i) CHECK(memcmp(got[i].data(), &want[4 * i], 4) == 0);
}

int main()
{
    // ---- images: ragged sizes around the reference's limits (structures.rs:23: 256; lib.rs:293: 128) ----
    struct Dim { uint32_t w, h; };
    const Dim dims[] = {{1, 1}, {3, 5}, {17, 9}, {257, 3}, {30, 300}, {48, 40}};
    for (const Dim &d : dims) {
        const uint64_t n = (uint64_t)d.w * d.h;
        std::vector<uint8_t> img(4 * n), out(4 * n);
        orc_synth_uniform(0x5EED0000u + d.w, n, img.data());
        for (uint32_t k : {1u, 2u, 5u, 16u}) {
            std::vector<float> c4(4 * k);
            std::vector<uint8_t> pal(4 * k);
            orc_extract_palette_kmeans(img.data(), d.w, d.h, k, 256, c4.data());
            orc_palette(img.data(), d.w, d.h, k, pal.data());
            for (int mode = 0; mode < 3; ++mode) {
                // (the k-means inside orc_reduce is the same for every mode: once per k is enough under the sanitizers)
                if (mode == (int)(k % 3u)) orc_reduce(img.data(), d.w, d.h, k, mode, out.data());
                else orc_reduce_octree(img.data(), d.w, d.h, k, mode, out.data());
                for (uint64_t i = 0; i < n; ++i) CHECK(out[4 * i + 3] == 255);
                orc_find(img.data(), d.w, d.h, pal.data(), k, mode, out.data());
                for (uint64_t i = 0; i < n; ++i) CHECK(out[4 * i + 3] == 255);
            }
            std::vector<uint8_t> opal(4 * (size_t)k + 4);
            CHECK(orc_palette_octree(img.data(), d.w, d.h, k, opal.data()) <= k);
            octree_both(std::vector<uint8_t>(img.begin(), img.begin() + 4 * (n < 800 ? n : 800)), k);
            // the product's host colour maths against the oracle's (structures.rs:523-553, 581-617; mix_colors.wgsl:53-67)
            for (uint32_t i = 0; i < k; ++i) {
                float a[3], b[3];
                uint8_t ra[3], rb[3];
                kmg::crate_srgb8_to_lab(&pal[4 * i], a);
                orc_palette_srgb8_to_lab(&pal[4 * i], b);
                CHECK(memcmp(a, b, sizeof a) == 0);
                kmg::crate_lab_to_srgb8(&c4[4 * i], ra);
                orc_palette_lab_to_srgb8(&c4[4 * i], rb);
                CHECK(memcmp(ra, rb, 3) == 0);
            }
            if (k >= 2) CHECK(kmg::dither_threshold(c4.data(), k) == orc_dither_threshold(c4.data(), k));
        }
        // the pieces, one by one (S1-S12)
        std::vector<float> lab(3 * n);
        orc_rgb_to_lab(img.data(), n, lab.data());
        const uint32_t k = 7;
        std::vector<float> c4(4 * k);
        orc_init_centroids(lab.data(), d.w, d.h, k, c4.data());
        std::vector<uint32_t> labels(n), labels2(n);
        std::vector<int64_t> acc(4 * k), acc2(4 * k);
        for (int literal = 0; literal < 3; ++literal) orc_assign(lab.data(), n, c4.data(), k, literal, labels.data());
        orc_assign(lab.data(), n, c4.data(), k, 0, labels.data());
        orc_accumulate(lab.data(), labels.data(), n, k, acc.data());
        orc_assign_accumulate_rgba(img.data(), n, c4.data(), k, labels2.data(), acc2.data());
        CHECK(labels == labels2);
        CHECK(acc == acc2);
        std::vector<float> c4b = c4;
        orc_finalize(acc.data(), k, 1.0f, c4b.data());
        orc_lloyd(lab.data(), n, k, c4.data(), labels.data(), 128, 8, 1.0f);
        std::vector<uint32_t> idx(n);
        orc_dither(lab.data(), d.w, d.h, c4.data(), k, idx.data());
        for (uint64_t i = 0; i < n; ++i) CHECK(idx[i] <= k);
        std::vector<float> melded(3 * n);
        orc_meld(lab.data(), d.w, d.h, c4.data(), k, melded.data());
        orc_lab_to_rgba8(melded.data(), n, out.data());
        uint32_t nw = 0, nh = 0;
        orc_resized_dims(d.w, d.h, 8, &nw, &nh);
        std::vector<uint8_t> small(4 * (size_t)nw * nh);
        orc_resize(img.data(), d.w, d.h, nw, nh, small.data());
    }
    // ---- the product's octree on the shapes its own tests use: one pixel, one colour, more colours asked than there are ----
    {
        std::vector<uint8_t> one = {9, 10, 20, 255};
        octree_both(one, 1); octree_both(one, 5);
        std::vector<uint8_t> flat;
        for (int i = 0; i < 150; ++i) for (uint8_t v : {(uint8_t)(10 + i % 3), (uint8_t)20, (uint8_t)30, (uint8_t)255}) flat.push_back(v);
        for (uint32_t k : {1u, 2u, 3u, 10u, 4000u}) octree_both(flat, k);
        std::vector<uint8_t> noise(4 * 1200);                       // (the oracle's octree is linear scans: quadratic in the leaves)
        orc_synth_uniform(77, 1200, noise.data());
        for (uint32_t k : {1u, 16u, 256u, 1500u}) octree_both(noise, k);
        CHECK(kmg::octree_palette(noise.data(), 1200, 0).empty());                       // octree.rs:67-69
    }
    // ---- kmg_math.h on the edges of its domain ----
    {
        float lut[256], olut[256];
        kmg::build_srgb_lut100(lut);
        orc_srgb_lut(olut);
        CHECK(memcmp(lut, olut, sizeof lut) == 0);
        for (float x : {0.001f, 0.008856f, 0.008857f, 0.5f, 1.0f, 1.0000001f, 2.0f}) CHECK(kmg::cbrt_cr(x) == orc_cbrt(x));
        for (uint32_t c = 0; c < (1u << 24); c += 4099) {
            float L, a, b, want[3];
            const uint8_t px[4] = {(uint8_t)(c & 255), (uint8_t)((c >> 8) & 255), (uint8_t)((c >> 16) & 255), 255};
            kmg::linear100_to_lab(lut[px[0]], lut[px[1]], lut[px[2]], L, a, b);
            orc_rgb_to_lab(px, 1, want);
            CHECK(L == want[0] && a == want[1] && b == want[2]);
            uint8_t back[4], oback[4];
            const float lab3[3] = {L, a, b};
            kmg::shader_lab_to_rgba8(lab3, back);
            orc_lab_to_rgba8(lab3, 1, oback);
            CHECK(memcmp(back, oback, 4) == 0);
        }
    }
    if (failures) { fprintf(stderr, "%d checks failed\n", failures); return 1; }
    printf("sanitized ok\n");
    return 0;
}
