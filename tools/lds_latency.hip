// lds_latency.hip -- round trip of a broadcast ds_read_b128 as ONE wave per SIMD sees it, alone and with arithmetic between the
// reads (the shape of k_assign's centroid loop on a small image: four reads, ~60 vector instructions, repeat).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_latency tools/lds_latency.hip && /tmp/lds_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int WORK, int READS>
__global__ __launch_bounds__(256) void k_lds(float *out, int iters, float a)
{
    __shared__ float4 s[256];
    s[threadIdx.x] = make_float4((float)threadIdx.x, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    float x0 = (float)threadIdx.x * 1e-3f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f;
    uint32_t j = 0;
    for (int i = 0; i < iters; ++i) {
        float4 c[READS > 0 ? READS : 1];
#pragma unroll
        for (int r = 0; r < READS; ++r) c[r] = s[(j + r) & 255u];          // uniform index: broadcast reads
        j = (j + READS) & 255u;
#pragma unroll
        for (int r = 0; r < READS; ++r) { x0 += c[r].x; x1 += c[r].y; x2 += c[r].z; x3 += c[r].w; }
#pragma unroll
        for (int w = 0; w < WORK; ++w) {
            x0 = __builtin_fmaf(x0, a, 0.001f); x1 = __builtin_fmaf(x1, a, 0.001f);
            x2 = __builtin_fmaf(x2, a, 0.001f); x3 = __builtin_fmaf(x3, a, 0.001f);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
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
    const int iters = 4096;
    for (int grid : {171, 1024}) {
        float t;
        t = time([&] { hipLaunchKernelGGL((k_lds<0, 1>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); }, 20);
        printf("grid %4d  1 read,  no work : %7.1f ns per iteration\n", grid, t * 1e6 / iters);
        t = time([&] { hipLaunchKernelGGL((k_lds<0, 4>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); }, 20);
        printf("grid %4d  4 reads, no work : %7.1f ns per iteration\n", grid, t * 1e6 / iters);
        t = time([&] { hipLaunchKernelGGL((k_lds<12, 4>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); }, 20);
        printf("grid %4d  4 reads, 48 fma  : %7.1f ns per iteration\n", grid, t * 1e6 / iters);
        t = time([&] { hipLaunchKernelGGL((k_lds<12, 0>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); }, 20);
        printf("grid %4d  0 reads, 48 fma  : %7.1f ns per iteration\n", grid, t * 1e6 / iters);
    }
    return 0;
}
