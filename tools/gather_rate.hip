// gather_rate.hip -- random-gather throughput vs table size (which levels of a hierarchical label
// table stay L2-resident).  67M random indices; 1-byte and 2-byte entries; fraction of active lanes.
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

template <typename T, bool NT>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ idx, uint64_t n, const T *__restrict__ tab,
                                                uint32_t mask, uint32_t active_pct, uint32_t *__restrict__ out)
{
    const uint64_t tiles = n / 2048;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(idx + tile * 2048 + g * 1024 + threadIdx.x * 4);
            u32x4 q = NT ? __builtin_nontemporal_load(p) : *p;
            v[g * 4] = q.x; v[g * 4 + 1] = q.y; v[g * 4 + 2] = q.z; v[g * 4 + 3] = q.w;
        }
        uint32_t r[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            r[p] = 0;
            if ((v[p] >> 24) % 100u < active_pct) r[p] = tab[v[p] & mask];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 *p = reinterpret_cast<u32x4 *>(out + tile * 2048 + g * 1024 + threadIdx.x * 4);
            u32x4 q = {r[g * 4], r[g * 4 + 1], r[g * 4 + 2], r[g * 4 + 3]};
            if (NT) __builtin_nontemporal_store(q, p); else *p = q;
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
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
    };
    for (int nt = 0; nt < 2; ++nt)
        for (uint32_t bits : {0u, 16u, 18u, 19u, 20u, 21u, 22u, 24u}) {
            for (uint32_t pct : {100u, 30u, 10u}) {
                uint32_t mask = bits ? ((1u << bits) - 1) : 0;
                float ms1 = time([&] {
                    if (nt) hipLaunchKernelGGL((k_gather<uint8_t, true>), dim3(4096), dim3(256), 0, 0, idx, n, tab, mask, pct, out);
                    else hipLaunchKernelGGL((k_gather<uint8_t, false>), dim3(4096), dim3(256), 0, 0, idx, n, tab, mask, pct, out);
                });
                float ms2 = time([&] {
                    if (nt) hipLaunchKernelGGL((k_gather<uint16_t, true>), dim3(4096), dim3(256), 0, 0, idx, n, (uint16_t *)tab, mask, pct, out);
                    else hipLaunchKernelGGL((k_gather<uint16_t, false>), dim3(4096), dim3(256), 0, 0, idx, n, (uint16_t *)tab, mask, pct, out);
                });
                printf("nt=%d entries=2^%-2u active=%3u%%  u8 %.3f ms  u16 %.3f ms\n", nt, bits, pct, ms1, ms2);
            }
        }
    return 0;
}
