// valu_rate.hip -- microbenchmark: sustained issue rate of v_fma_f32 vs v_pk_fma_f32 vs mixed
// VALU on gfx950 (decides whether packing the CIE94 key arithmetic into v_pk_* pays).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define N_ITERS 4096

__global__ __launch_bounds__(256) void k_fma(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_pk_fma(float *out, float a, float b)
{
    float2v x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < N_ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av), "v"(bv));
    }
    float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// the ops the key loop uses: sub, mul, max, cmp+cndmask
__global__ __launch_bounds__(256) void k_mix(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITERS; ++i) {
        asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_max_f32 %2, %2, %8\n v_sub_f32 %3, %3, %9\n"
                     "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_mul_f32 %5, %5, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// sgpr operand variant (constant bus)
__global__ __launch_bounds__(256) void k_fma_sgpr(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITERS; ++i) {
        asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %9, %3\n"
                     "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %9, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %9, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "s"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <typename F>
static void run(const char *name, F kernel, float *d, int blocks, double lane_ops_per_instr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double wave_instr = (double)blocks * 4 * N_ITERS * 8;
    double per_simd = wave_instr / 1024.0;
    printf("%-12s blocks=%5d  %.3f ms  wave-instr/s=%.3e  cycles/wave-instr/SIMD @2.4GHz=%.2f  lane-ops/s=%.3e\n", name, blocks, ms,
           wave_instr / (ms * 1e-3), (ms * 1e-3 * 2.4e9) / per_simd, wave_instr * 64 * lane_ops_per_instr / (ms * 1e-3));
}

int main()
{
    float *d; hipMalloc(&d, 8192 * 256 * sizeof(float));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run("v_fma_f32", k_fma, d, blocks, 1);
        run("v_pk_fma_f32", k_pk_fma, d, blocks, 2);
        run("mix", k_mix, d, blocks, 1);
        run("v_sub sgpr", k_fma_sgpr, d, blocks, 1);
    }
    return 0;
}
