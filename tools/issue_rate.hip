// issue_rate.hip -- microbenchmark: what one SIMD of gfx950 sustains for the instruction patterns of the
// colour-table kernels (VALU chains, SALU, VALU<->SALU hand-offs, v_readlane, the candidate loop itself) at
// 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate tools/issue_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define N_ITERS 2048

#define KERNEL(name, body, ninstr)                                                                         \
    __global__ __launch_bounds__(256) void name(float *out, float a, float b, uint32_t m)                  \
    {                                                                                                      \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        uint32_t s0 = m, s1 = m + 1, s2 = m + 2, s3 = m + 3;                                               \
        for (int i = 0; i < N_ITERS; ++i) { body }                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3); \
    }                                                                                                      \
    static const int name##_n = ninstr;

// 8 independent VALU
KERNEL(k_valu_indep, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));, 8)
// 8 dependent VALU (one chain)
KERNEL(k_valu_chain, asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                                  "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                                  : "+v"(x0) : "v"(a), "v"(b));, 8)
// 8 independent SALU
KERNEL(k_salu_indep, asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                                  "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                                  : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");, 8)
// 4 VALU + 4 SALU interleaved, independent
KERNEL(k_mixed, asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %4, %4, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %5, %5, 3\n"
                             "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %6, %6, 5\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %7, %7, 7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc");, 8)
// v_readlane with an SGPR lane select, independent, results unused by VALU
KERNEL(k_readlane, asm volatile("v_readlane_b32 %0, %4, %8\n v_readlane_b32 %1, %5, %8\n v_readlane_b32 %2, %6, %8\n v_readlane_b32 %3, %7, %8\n"
                                "v_readlane_b32 %0, %5, %8\n v_readlane_b32 %1, %6, %8\n v_readlane_b32 %2, %7, %8\n v_readlane_b32 %3, %4, %8\n"
                                : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(m & 63u));, 8)
// readlane -> VALU consuming the SGPR
KERNEL(k_readlane_use, asm volatile("v_readlane_b32 %4, %0, %6\n v_readlane_b32 %5, %1, %6\n s_nop 0\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %5\n"
                                    "v_fma_f32 %0, %2, %2, %0\n v_fma_f32 %1, %3, %3, %1\n"
                                    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&s"(s0), "=&s"(s1) : "s"(m & 63u));, 6)
// VALU cmp -> VCC -> 2 cndmask (per colour), two colours
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n"
                               "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "vcc");, 6)
// ballot-style: v_cmp -> SGPR pair -> s_ff1 -> v_readlane -> VALU   (the cross-lane hand-off chain)
KERNEL(k_handoff, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_ff1_i32_b64 %3, vcc\n s_nop 0\n v_readlane_b32 %4, %2, %3\n s_nop 0\n v_add_f32 %0, %0, %4\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "=&s"(s0), "=&s"(s1) :: "vcc");, 4)
// the candidate loop of k_cube_scan, two colours per lane (5 SALU + 5 readlane + 34 VALU)
KERNEL(k_candidate, asm volatile(
           "s_ff1_i32_b32 %8, %12\n s_add_i32 %9, %12, -1\n v_readlane_b32 %10, %0, %8\n s_and_b32 %12, %9, %12\n v_readlane_b32 %11, %1, %8\n"
           "v_readlane_b32 %9, %2, %8\n v_readlane_b32 %8, %3, %8\n s_nop 0\n"
           "v_sub_f32 %4, %0, %10\n v_sub_f32 %5, %1, %11\n v_sub_f32 %6, %2, %9\n v_sub_f32 %7, %3, %8\n"
           "v_mul_f32 %4, %4, %4\n v_mul_f32 %7, %7, %7\n v_fma_f32 %5, %5, %5, %4\n v_fma_f32 %5, %6, %6, %5\n v_sub_f32 %5, %5, %7\n v_max_f32 %5, 0, %5\n"
           "v_fma_f32 %4, %7, %2, %4\n v_fma_f32 %4, %5, %3, %4\n v_cmp_lt_f32 vcc, %4, %0\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n"
           "v_sub_f32 %4, %1, %10\n v_sub_f32 %5, %2, %11\n v_sub_f32 %6, %3, %9\n v_sub_f32 %7, %0, %8\n"
           "v_mul_f32 %4, %4, %4\n v_mul_f32 %7, %7, %7\n v_fma_f32 %5, %5, %5, %4\n v_fma_f32 %5, %6, %6, %5\n v_sub_f32 %5, %5, %7\n v_max_f32 %5, 0, %5\n"
           "v_fma_f32 %4, %7, %2, %4\n v_fma_f32 %4, %5, %3, %4\n v_cmp_lt_f32 vcc, %4, %2\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %6, vcc\n"
           "s_or_b32 %12, %12, 0x10000\n"
           : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "+s"(m) :: "vcc", "scc");, 44)
// IEEE divide (pixel terms): 1 / x
KERNEL(k_div, x0 = 1.0f / (x0 + a); x1 = 1.0f / (x1 + b);, 24)
KERNEL(k_rcp, x0 = __builtin_amdgcn_rcpf(x0 + a); x1 = __builtin_amdgcn_rcpf(x1 + b); x2 = __builtin_amdgcn_rcpf(x2 + a); x3 = __builtin_amdgcn_rcpf(x3 + b);, 8)
// LDS broadcast read + use
__global__ __launch_bounds__(256) void k_lds_bcast(float *out, float a, float b, uint32_t m)
{
    __shared__ float4 tab[256];
    tab[threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    float x0 = threadIdx.x, x1 = 0, x2 = 0, x3 = 0;
    uint32_t j = m;
    for (int i = 0; i < N_ITERS; ++i) {
        const float4 c = tab[j & 255u];
        x0 += c.x; x1 += c.y; x2 += c.z; x3 += c.w;
        j = j * 5u + 1u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
static const int k_lds_bcast_n = 8;

template <typename F>
static void run(const char *name, F kernel, int n_instr, float *d, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0xF0F0F0F0u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0xF0F0F0F0u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double wave_instr = (double)blocks * 4 * N_ITERS * n_instr;
    printf("%-16s waves/SIMD %d  %.3f ms  cycles per wave-instr per SIMD @2.4GHz = %6.2f\n", name, blocks / 256, ms,
           (ms * 1e-3 * 2.4e9) / (wave_instr / 1024.0));
}

#define RUN(k) run(#k, k, k##_n, d, blocks)
int main()
{
    float *d; hipMalloc(&d, 8192 * 256 * sizeof(float));
    for (int blocks : {256, 512, 1024, 2048}) {     // 1, 2, 4, 8 waves per SIMD
        RUN(k_valu_indep); RUN(k_valu_chain); RUN(k_salu_indep); RUN(k_mixed); RUN(k_readlane); RUN(k_readlane_use);
        RUN(k_cmp_cnd); RUN(k_handoff); RUN(k_candidate); RUN(k_div); RUN(k_rcp); RUN(k_lds_bcast);
        printf("\n");
    }
    return 0;
}
