// gather_compact.hip -- does a divergent byte gather cost per INSTRUCTION or per ACTIVE LANE?  The label pass issues eight
// gathers per thread with ~16 % of the lanes active in each; here the same gathers are issued (A) that way and (B) compacted per
// lane: every lane walks its own active pixels, so a wave issues max-over-lanes instructions (~4) with most lanes active.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_compact tools/gather_compact.hip && /tmp/gather_compact
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
__global__ __launch_bounds__(1024) void k_gather(const uint32_t *__restrict__ idx, uint64_t n, const uint8_t *__restrict__ tab,
                                                 uint32_t mask, uint32_t active_pct, uint32_t *__restrict__ out)
{
    const uint64_t tiles = n / 8192;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(idx + tile * 8192 + g * 4096 + threadIdx.x * 4);
            u32x4 q = __builtin_nontemporal_load(p);
            v[g * 4] = q.x; v[g * 4 + 1] = q.y; v[g * 4 + 2] = q.z; v[g * 4 + 3] = q.w;
        }
        uint32_t r[8];
        uint32_t fm = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) { r[p] = v[p] & 255u; fm |= ((v[p] >> 24) % 100u < active_pct ? 1u : 0u) << p; }
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if ((fm >> p) & 1u) r[p] = tab[v[p] & mask];
        } else if (MODE == 2) {
            // the aligned dword that holds the byte, then a shift: is a sub-dword load dearer than a dword load?
            const uint32_t *tab32 = reinterpret_cast<const uint32_t *>(tab);
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if ((fm >> p) & 1u) { const uint32_t a = v[p] & mask; r[p] = (tab32[a >> 2] >> ((a & 3u) * 8u)) & 255u; }
        } else if (MODE == 1) {
            while (__ballot(fm != 0u)) {
                const uint32_t p = fm ? (uint32_t)__builtin_ctz(fm) : 8u;
                uint32_t c = v[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) c = p == (uint32_t)q ? v[q] : c;
                uint32_t val = 0;
                if (p < 8u) val = tab[c & mask];
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] = p == (uint32_t)q ? val : r[q];
                fm &= fm - 1u;
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 *p = reinterpret_cast<u32x4 *>(out + tile * 8192 + g * 4096 + threadIdx.x * 4);
            u32x4 q = {r[g * 4], r[g * 4 + 1], r[g * 4 + 2], r[g * 4 + 3]};
            __builtin_nontemporal_store(q, p);
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
    uint32_t *idx, *out; uint8_t *tab;
    hipMalloc(&idx, n * 4); hipMalloc(&out, n * 4); hipMalloc(&tab, 64 << 20);
    hipMemset(tab, 3, 64 << 20);
    hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
    };
    for (uint32_t bits : {16u, 24u})
        for (uint32_t pct : {0u, 8u, 16u, 30u, 100u}) {
            const uint32_t mask = (1u << bits) - 1;
            // (dynamic LDS of 128 KiB: one workgroup per CU, as the label pass)
            float a = time([&] { hipLaunchKernelGGL((k_gather<0>), dim3(256), dim3(1024), 128 << 10, 0, idx, n, tab, mask, pct, out); });
            float b = time([&] { hipLaunchKernelGGL((k_gather<1>), dim3(256), dim3(1024), 128 << 10, 0, idx, n, tab, mask, pct, out); });
            float c = time([&] { hipLaunchKernelGGL((k_gather<2>), dim3(256), dim3(1024), 128 << 10, 0, idx, n, tab, mask, pct, out); });
            printf("table 2^%-2u B  active %3u %%   masked x8 %.1f us   compacted %.1f us   dword loads %.1f us\n", bits, pct, a * 1e3, b * 1e3, c * 1e3);
        }
    return 0;
}
