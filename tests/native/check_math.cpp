// check_math.cpp -- host-side check of kmeans-gpu_amd/csrc/kmg_math.h (the arithmetic the gfx950
// kernels execute), compiled with g++ by tests/test_host_math.py:
//   1. kmg::cbrt_cr is the correctly rounded cube root on every binary32 in [1e-3, 2]
//   2. the Lab of all 2^24 colours equals the oracle's (oracle/kmg_oracle.c), bit for bit
//   3. cie94 / cie94_key equal the oracle's on random pairs
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "kmg_math.h"
#include "kmg_color.h"
#include "kmg_oracle.h"

static uint64_t s64 = 88172645463325252ull;
static uint32_t rnd() { s64 ^= s64 << 13; s64 ^= s64 >> 7; s64 ^= s64 << 17; return (uint32_t)(s64 >> 16); }

int main()
{
    long bad = 0;
    uint32_t lo = kmg::float_to_bits(0.001f), hi = kmg::float_to_bits(2.0f);
    for (uint32_t u = lo; u <= hi; ++u) {
        float x = kmg::bits_to_float(u);
        if (kmg::cbrt_cr(x) != (float)cbrt((double)x)) ++bad;
    }
    printf("cbrt_mismatches %ld\n", bad);

    float lut[256], olut[256];
    kmg::build_srgb_lut100(lut);
    orc_srgb_lut(olut);
    long lutbad = 0;
    for (int i = 0; i < 256; ++i) lutbad += memcmp(&lut[i], &olut[i], 4) != 0;
    printf("lut_mismatches %ld\n", lutbad);

    long labbad = 0;
    const uint32_t CH = 1 << 16;
    std::vector<uint8_t> px(4 * CH);
    std::vector<float> want(3 * CH);
    for (uint32_t base = 0; base < (1u << 24); base += CH) {
        for (uint32_t i = 0; i < CH; ++i) {
            uint32_t c = base + i;
            px[4 * i] = c & 255; px[4 * i + 1] = (c >> 8) & 255; px[4 * i + 2] = (c >> 16) & 255; px[4 * i + 3] = 255;
        }
        orc_rgb_to_lab(px.data(), CH, want.data());
        for (uint32_t i = 0; i < CH; ++i) {
            float L, a, b;
            kmg::linear100_to_lab(lut[px[4 * i]], lut[px[4 * i + 1]], lut[px[4 * i + 2]], L, a, b);
            float got[3] = {L, a, b};
            labbad += memcmp(got, &want[3 * i], 12) != 0;
        }
    }
    printf("lab_mismatches %ld\n", labbad);

    long dbad = 0;
    for (int t = 0; t < 2000000; ++t) {
        float p[3], c[3];
        p[0] = (rnd() % 100001) * 1e-3f; p[1] = (rnd() % 256001) * 1e-3f - 128.0f; p[2] = (rnd() % 256001) * 1e-3f - 128.0f;
        c[0] = (rnd() % 100001) * 1e-3f; c[1] = (rnd() % 256001) * 1e-3f - 128.0f; c[2] = (rnd() % 256001) * 1e-3f - 128.0f;
        float a = kmg::cie94(p[0], p[1], p[2], c[0], c[1], c[2]), b = orc_cie94(p, c);
        kmg::PixelTerms pt = kmg::pixel_terms(p[0], p[1], p[2]);
        float ka = kmg::cie94_key(pt, c[0], c[1], c[2], kmg::chroma(c[1], c[2])), kb = orc_cie94_key(p, c);
        dbad += memcmp(&a, &b, 4) != 0;
        dbad += memcmp(&ka, &kb, 4) != 0;
    }
    printf("distance_mismatches %ld\n", dbad);
    return (bad || lutbad || labbad || dbad) ? 1 : 0;
}
