// lut_gather.hip -- microbenchmark for the label-materialisation pass of the colour-table path:
// labels[i] = LUT[cell_major(rgb_i)] for 8192x8192 uniformly random pixels (worst case for the
// gather) with a 16 MiB u8 table; also a coherent image (sorted colours) for the best case.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t cell_major(uint32_t px)
{
    uint32_t r = px & 255u, g = (px >> 8) & 255u, b = (px >> 16) & 255u;
    return ((r >> 3) << 19) | ((g >> 3) << 14) | ((b >> 3) << 9) | ((r & 7u) << 6) | ((g & 7u) << 3) | (b & 7u);
}

template <int PPT>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ rgba, uint64_t n,
                                                const uint8_t *__restrict__ lut, uint32_t *__restrict__ labels)
{
    constexpr uint64_t TILE = 256ull * PPT;
    const uint64_t tiles = n / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t idx[PPT];
#pragma unroll
        for (int g = 0; g < PPT / 4; ++g) {
            uint4 v = *reinterpret_cast<const uint4 *>(rgba + tile * TILE + g * 1024 + threadIdx.x * 4);
            idx[g * 4] = cell_major(v.x); idx[g * 4 + 1] = cell_major(v.y);
            idx[g * 4 + 2] = cell_major(v.z); idx[g * 4 + 3] = cell_major(v.w);
        }
        uint32_t lab[PPT];
#pragma unroll
        for (int p = 0; p < PPT; ++p) lab[p] = lut[idx[p]];
#pragma unroll
        for (int g = 0; g < PPT / 4; ++g)
            *reinterpret_cast<uint4 *>(labels + tile * TILE + g * 1024 + threadIdx.x * 4) =
                make_uint4(lab[g * 4], lab[g * 4 + 1], lab[g * 4 + 2], lab[g * 4 + 3]);
    }
}

__global__ void k_fill(uint32_t *rgba, uint64_t n, int coherent)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = coherent ? (i >> 2) : (i + 1) * 0x9E3779B97F4A7C15ull;
    if (!coherent) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; }
    rgba[i] = (uint32_t)(z & 0xFFFFFF) | 0xFF000000u;
}

int main()
{
    const uint64_t n = 8192ull * 8192ull;
    uint32_t *rgba, *labels; uint8_t *lut;
    hipMalloc(&rgba, n * 4); hipMalloc(&labels, n * 4); hipMalloc(&lut, 1 << 24);
    hipMemset(lut, 7, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int coherent = 0; coherent < 2; ++coherent) {
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, rgba, n, coherent);
        for (int grid : {1024, 2048, 4096, 8192, 16384}) {
            for (int ppt : {4, 8, 16}) {
                auto launch = [&]() {
                    if (ppt == 4) hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, rgba, n, lut, labels);
                    if (ppt == 8) hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, rgba, n, lut, labels);
                    if (ppt == 16) hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(256), 0, 0, rgba, n, lut, labels);
                };
                launch(); hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < 5; ++r) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                printf("coherent=%d grid=%5d ppt=%2d  %.3f ms  %.1f GB/s (8 B/px)\n", coherent, grid, ppt, ms, 8.0 * n / ms / 1e6);
            }
        }
    }
    return 0;
}
