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
