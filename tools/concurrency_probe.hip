// Do a label-pass-like kernel (1 workgroup of 1024 threads + 128 KiB LDS per CU, streaming) and a
// cube-pass-like kernel (256-thread workgroups, VALU bound, ~120 VGPRs) overlap when launched on two streams?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_stream(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n4)
{
    __shared__ uint32_t tab[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) tab[i] = i * 2654435761u;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 1024) {
        u32x4 v = __builtin_nontemporal_load(in + i);
        v.x = tab[v.x & 32767]; v.y = tab[v.y & 32767]; v.z = tab[v.z & 32767]; v.w = tab[v.w & 32767];
        __builtin_nontemporal_store(v, out + i);
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_valu(float *out, int iters)
{
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = threadIdx.x * 0.001f + r;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = fmaf(a[r], 1.0001f, 0.5f);
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += a[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const uint64_t n4 = (8192ull * 8192ull) / 4;
    u32x4 *in, *out; float *o2;
    CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&out, n4 * 16)); CK(hipMalloc(&o2, 4096 * 256 * 4));
    CK(hipMemset(in, 1, n4 * 16));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, bool L, bool C, int reps) {
        for (int w = 0; w < 2; ++w) {
            if (L) hipLaunchKernelGGL(k_stream, dim3(256), dim3(1024), 0, s1, in, out, n4);
            if (C) hipLaunchKernelGGL(k_valu<100>, dim3(1024), dim3(256), 0, s2, o2, 600);
        }
        hipDeviceSynchronize();
        hipEventRecord(e0, s1); hipStreamWaitEvent(s2, e0, 0);
        for (int r = 0; r < reps; ++r) {
            if (L) hipLaunchKernelGGL(k_stream, dim3(256), dim3(1024), 0, s1, in, out, n4);
            if (C) hipLaunchKernelGGL(k_valu<100>, dim3(1024), dim3(256), 0, s2, o2, 600);
        }
        hipEventRecord(e1, s2); hipStreamWaitEvent(s1, e1, 0);
        hipEventRecord(e1, s1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s %.3f ms per rep\n", name, ms / reps);
    };
    time("stream alone", true, false, 20);
    time("valu alone", false, true, 20);
    time("both, two streams", true, true, 20);
    return 0;
}
