// gather_combo.hip -- emulates the hierarchical label pass: LDS level (40% resolved), u32 sub table
// (1 MiB, 60% of pixels), then for 29% of pixels one of: per-colour u8 table (16 MiB) / 32-byte block
// (16 B + 8 B loads) / 16-byte block (one load) / 8-byte block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash(uint64_t i)
{
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (uint32_t)z;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_combo(const uint32_t *__restrict__ idx, uint64_t n, const uint32_t *__restrict__ sub,
                                                const uint8_t *__restrict__ fine, const uint4 *__restrict__ blocks, uint32_t nblocks,
                                                uint32_t *__restrict__ out)
{
    __shared__ uint16_t s_cell[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) s_cell[i] = (uint16_t)i;
    __syncthreads();
    const uint64_t tiles = n / 8192;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(idx + tile * 8192 + g * 4096 + threadIdx.x * 4));
            v[g * 4] = q.x; v[g * 4 + 1] = q.y; v[g * 4 + 2] = q.z; v[g * 4 + 3] = q.w;
        }
        uint32_t r[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t ci = v[p] & 0xFFFFFF, sel = (v[p] >> 24) % 100u;
            uint32_t e = s_cell[ci >> 9];
            if (sel >= 40) e = sub[ci >> 6];
            r[p] = e;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t ci = v[p] & 0xFFFFFF, sel = (v[p] >> 24) % 100u;
            if (sel >= 71) {
                if (MODE == 0) r[p] = fine[ci];
                if (MODE == 1) { const uint4 *b = blocks + 2ull * ((ci * 2654435761u) % nblocks); uint4 c = b[0]; uint2 l = *reinterpret_cast<const uint2 *>(b + 1); r[p] = c.x ^ c.w ^ l.x ^ l.y; }
                if (MODE == 2) { const uint4 *b = blocks + ((ci * 2654435761u) % nblocks); uint4 c = b[0]; r[p] = c.x ^ c.w; }
                if (MODE == 3) { const uint2 *b = reinterpret_cast<const uint2 *>(blocks) + ((ci * 2654435761u) % nblocks); uint2 c = b[0]; r[p] = c.x ^ c.y; }
                if (MODE == 4) r[p] = ci;
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 q = {r[g * 4], r[g * 4 + 1], r[g * 4 + 2], r[g * 4 + 3]};
            __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(out + tile * 8192 + g * 4096 + threadIdx.x * 4));
        }
    }
}

__global__ void k_fill(uint32_t *idx, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = hash(i);
}

int main()
{
    const uint64_t n = 8192ull * 8192ull;
    uint32_t *idx, *out, *sub; uint8_t *fine; uint4 *blocks;
    hipMalloc(&idx, n * 4); hipMalloc(&out, n * 4); hipMalloc(&sub, 4 << 18); hipMalloc(&fine, 1 << 24); hipMalloc(&blocks, 8 << 20);
    hipMemset(sub, 1, 4 << 18); hipMemset(fine, 1, 1 << 24); hipMemset(blocks, 1, 8 << 20);
    hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
    };
    const uint32_t nb = 76000;
    printf("fine u8 16MiB      : %.3f ms\n", time([&] { hipLaunchKernelGGL(k_combo<0>, dim3(512), dim3(1024), 0, 0, idx, n, sub, fine, blocks, nb, out); }));
    printf("32B block (16+8 B) : %.3f ms\n", time([&] { hipLaunchKernelGGL(k_combo<1>, dim3(512), dim3(1024), 0, 0, idx, n, sub, fine, blocks, nb, out); }));
    printf("16B block          : %.3f ms\n", time([&] { hipLaunchKernelGGL(k_combo<2>, dim3(512), dim3(1024), 0, 0, idx, n, sub, fine, blocks, nb, out); }));
    printf("8B block           : %.3f ms\n", time([&] { hipLaunchKernelGGL(k_combo<3>, dim3(512), dim3(1024), 0, 0, idx, n, sub, fine, blocks, nb, out); }));
    printf("no third level     : %.3f ms\n", time([&] { hipLaunchKernelGGL(k_combo<4>, dim3(512), dim3(1024), 0, 0, idx, n, sub, fine, blocks, nb, out); }));
    return 0;
}
