// colour_index (kmeans-gpu_amd/csrc/kmg_table.h: the cell-major order of the colour table, computed with two 24-bit multiplies)
// against its plain bit-by-bit form, and index_to_rgb as its inverse, over all 2^24 colours and a few alpha bytes.
#include <cstdio>
#include "kmg_table.h"

int main()
{
    unsigned long long bad = 0, bad_inverse = 0;
    for (uint32_t alpha : {0u, 1u, 0x7Fu, 0xFFu})
        for (uint32_t c = 0; c < (1u << 24); ++c) {
            const uint32_t px = c | (alpha << 24);
            const uint32_t idx = kmg::colour_index(px);
            bad += idx != kmg::colour_index_reference(px) || idx >= (1u << 24);
            uint32_t r, g, b;
            kmg::index_to_rgb(idx, r, g, b);
            bad_inverse += (r | (g << 8) | (b << 16)) != c;
        }
    printf("colour_index_mismatches %llu\ninverse_mismatches %llu\n", bad, bad_inverse);
    return bad || bad_inverse ? 1 : 0;
}
