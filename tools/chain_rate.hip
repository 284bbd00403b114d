// chain_rate.hip -- how fast does ONE wave per SIMD run a dependent chain, and does it depend on how much of the device is
// busy / how long the kernel is (clock ramp)?  ns per dependent v_fma_f32 and per independent-4 group, by grid and length.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/chain_rate tools/chain_rate.hip && /tmp/chain_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int ILP>
__global__ __launch_bounds__(256) void k_chain(float *out, int iters, float a, float b)
{
    float x[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) x[q] = (float)threadIdx.x * 1e-3f + (float)q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int q = 0; q < ILP; ++q) x[q] = __builtin_fmaf(x[q], a, b);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < ILP; ++q) s += x[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with integer work (v_xor_b32 + v_add_u32 per step) and with compare + select (v_cmp + v_cndmask per step)
template <int ILP, int KIND>
__global__ __launch_bounds__(256) void k_chain_int(uint32_t *out, int iters, uint32_t a, uint32_t b)
{
    uint32_t x[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) x[q] = threadIdx.x * 2654435761u + (uint32_t)q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int q = 0; q < ILP; ++q) {
                if (KIND == 0) x[q] = (x[q] ^ a) + b;
                else x[q] = x[q] > a ? x[q] - b : x[q] + a;           // v_cmp + v_sub + v_add + v_cndmask
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < ILP; ++q) s += x[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float *out; hipMalloc(&out, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, int reps) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
    };
    // does the dispatcher spread few workgroups over the CUs, with and without LDS?  (one wave per SIMD = 2.5 ns per dependent fma)
    for (int lds : {0, 13 << 10, 40 << 10, 100 << 10})
        for (int grid : {64, 171, 256}) {
            // (four independent chains: one wave per SIMD runs them at 1.25 ns per fma, waves that share a SIMD at 1.75 ns x waves)
            float t1 = time([&] { hipLaunchKernelGGL((k_chain<4>), dim3(grid), dim3(256), lds, 0, out, 1024, 0.999f, 0.001f); }, 50);
            printf("lds %6d B  grid %4d  4 chains of 16384 fma: %8.1f us = %5.2f ns per fma\n", lds, grid, t1 * 1e3, t1 * 1e6 / 65536.0);
        }
    // how much independence inside ONE wave the full rate needs: per-SIMD time per wave-instruction by chains per wave and waves per SIMD
    for (int grid : {256, 1024, 2048})
        for (int ilp : {1, 2, 3, 4, 8}) {
            const int iters = 4096;
            float t = 0;
            if (ilp == 1) t = time([&] { hipLaunchKernelGGL((k_chain<1>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, 20);
            if (ilp == 2) t = time([&] { hipLaunchKernelGGL((k_chain<2>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, 20);
            if (ilp == 3) t = time([&] { hipLaunchKernelGGL((k_chain<3>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, 20);
            if (ilp == 4) t = time([&] { hipLaunchKernelGGL((k_chain<4>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, 20);
            if (ilp == 8) t = time([&] { hipLaunchKernelGGL((k_chain<8>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, 20);
            const double waves_per_simd = grid / 256.0, instrs = (double)iters * 16 * ilp;
            printf("waves/SIMD %2.0f  chains/wave %d: %6.2f ns per wave-instruction per SIMD\n", waves_per_simd, ilp, t * 1e6 / (instrs * waves_per_simd));
        }
    for (int kind : {0, 1})
        for (int ilp : {1, 2, 4}) {
            const int iters = 4096, grid = 1024;
            uint32_t *o = (uint32_t *)out;
            float t = 0;
#define KMG_RUN(I, K) t = time([&] { hipLaunchKernelGGL((k_chain_int<I, K>), dim3(grid), dim3(256), 0, 0, o, iters, 0x9E3779B9u, 12345u); }, 20)
            if (kind == 0) { if (ilp == 1) KMG_RUN(1, 0); else if (ilp == 2) KMG_RUN(2, 0); else KMG_RUN(4, 0); }
            else           { if (ilp == 1) KMG_RUN(1, 1); else if (ilp == 2) KMG_RUN(2, 1); else KMG_RUN(4, 1); }
#undef KMG_RUN
            const double per_step = kind == 0 ? 2.0 : 4.0;
            printf("4 waves/SIMD  %s  chains/wave %d: %6.2f ns per step per SIMD (~%.0f instructions per step)\n",
                   kind == 0 ? "xor + add   " : "cmp + select", ilp, t * 1e6 / ((double)iters * 8 * ilp * 4.0), per_step);
        }
    for (int grid : {21, 171, 256, 1024, 2048})
        for (int iters : {64, 1024, 16384}) {
            const int reps = iters >= 16384 ? 5 : 50;
            float t1 = time([&] { hipLaunchKernelGGL((k_chain<1>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, reps);
            float t4 = time([&] { hipLaunchKernelGGL((k_chain<4>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 0.001f); }, reps);
            const double n1 = (double)iters * 16, n4 = (double)iters * 16 * 4;
            printf("grid %4d  chain of %7.0f fma: %8.1f us = %5.2f ns per fma | 4 independent chains: %8.1f us = %5.2f ns per fma\n",
                   grid, n1, t1 * 1e3, t1 * 1e6 / n1, t4 * 1e3, t4 * 1e6 / n4);
        }
    return 0;
}
