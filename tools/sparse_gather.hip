// sparse_gather.hip -- does a byte gather with few active lanes cost the texture addresser as much as a full one?
// N random 1-byte gathers from a 16 MiB table, issued (a) densely: every lane of a wave-instruction gathers,
// (b) sparsely: 8 wave-instructions with 1/8 of the lanes active each (what a label pass with ~15 % slab pixels
// does).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/sparse_gather tools/sparse_gather.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(1024) void k_dense(const uint32_t *idx, const uint8_t *tab, uint32_t *out, uint32_t n_per_thread)
{
    uint32_t acc = 0;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t i = 0; i < n_per_thread; ++i) acc += tab[idx[t + i * stride] & 0xFFFFFFu];
    out[t] = acc;
}

// the same gathers, but instruction i only has the lanes with (lane & 7) == (i & 7) active, and 8x as many instructions
__global__ __launch_bounds__(1024) void k_sparse(const uint32_t *idx, const uint8_t *tab, uint32_t *out, uint32_t n_per_thread)
{
    uint32_t acc = 0;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t i = 0; i < n_per_thread; ++i) {
        const uint32_t v = idx[t + i * stride] & 0xFFFFFFu;
#pragma unroll
        for (uint32_t p = 0; p < 8; ++p)
            if (((threadIdx.x + i) & 7u) == p) acc += tab[(v + p * 0x2001u) & 0xFFFFFFu];
    }
    out[t] = acc;
}

int main()
{
    const uint32_t blocks = 256, threads = 1024, npt = 64;
    const size_t n = (size_t)blocks * threads * npt;
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    uint32_t *d_idx, *d_out; uint8_t *d_tab;
    hipMalloc(&d_idx, n * 4); hipMalloc(&d_out, blocks * threads * 4); hipMalloc(&d_tab, 1 << 24);
    hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_tab, 1, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_dense, dim3(blocks), dim3(threads), 0, 0, d_idx, d_tab, d_out, npt);
            else hipLaunchKernelGGL(k_sparse, dim3(blocks), dim3(threads), 0, 0, d_idx, d_tab, d_out, npt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %zu gathers in %.3f ms = %.2f G gathers/s\n", which ? "sparse (1/8 lanes per instr)" : "dense ", n, ms, n / ms * 1e-6);
        }
    }
    return 0;
}
