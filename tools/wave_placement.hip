// wave_placement.hip -- where do the waves of a SMALL grid land?  Every wave records (XCC_ID, HW_ID) and then spins ~20 us so
// that all workgroups are resident together; the host prints how many waves share a SIMD / a CU / an XCD.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/wave_placement tools/wave_placement.hip && /tmp/wave_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_where(uint32_t *out, int spin, int use_barrier)
{
    extern __shared__ float s_dummy[];
    if (use_barrier) { s_dummy[threadIdx.x] = 1.0f; __syncthreads(); }
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}

int main()
{
    uint32_t *d; hipMalloc(&d, 1 << 20);
    std::vector<uint32_t> h(1 << 18);
    for (int lds : {0, 13 << 10})
        for (int barrier : {0, 1})
            for (int grid : {4, 16, 64, 171, 256}) {
                hipMemset(d, 0, 1 << 20);
                hipLaunchKernelGGL(k_where, dim3(grid), dim3(256), lds, 0, d, 2000 /* x 10 ns */, barrier);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), d, grid * 4 * 8, hipMemcpyDeviceToHost);
                std::map<uint64_t, int> simd, cu, xcd;
                for (int w = 0; w < grid * 4; ++w) {
                    const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xF;
                    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: 3 bits)
                    const uint32_t simd_id = (hw >> 4) & 3, cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    const uint64_t cu_key = ((uint64_t)xcc << 16) | (se << 8) | (sh << 4) | cu_id;
                    simd[(cu_key << 2) | simd_id]++; cu[cu_key]++; xcd[xcc]++;
                }
                int max_simd = 0, max_cu = 0;
                for (auto &e : simd) max_simd = std::max(max_simd, e.second);
                for (auto &e : cu) max_cu = std::max(max_cu, e.second);
                printf("lds %5d barrier %d grid %3d (%4d waves): %3zu XCDs, %3zu CUs, %4zu SIMDs in use; most waves on one SIMD %d, on one CU %d\n",
                       lds, barrier, grid, grid * 4, xcd.size(), cu.size(), simd.size(), max_simd, max_cu);
            }
    return 0;
}
