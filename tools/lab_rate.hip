// lab_rate.hip -- microbenchmark + exhaustive equality check of per-pixel sRGB -> Lab variants on gfx950.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -I kmeans-gpu_amd/csrc -o tools/lab_rate tools/lab_rate.hip
// The device build of kmg_math.h (hardware log2 / exp2 seed for the cube root, multiply-and-correct for the
// divisions by the white point) must produce the SAME floats as the host definition for all 2^24 colours; cube
// root and quotients are additionally compared over every binary32 of their input ranges.  Exit code 1 on any difference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "kmg_math.h"
#include "kmg_color.h"

using namespace kmg;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- the definitions the device code must reproduce: bit-trick seed + three binary32 Newton steps (what the
// host build of kmg_math.h runs, verified against libm by tests/test_host_math.py) and IEEE divisions ----------
__device__ __forceinline__ float cbrt_bits(float x)
{
    uint32_t ir = 0x54A21D2Au - float_to_bits(x) / 3u;
    float r = bits_to_float(ir);
    float xt = x * 0.33333334f;
    for (int i = 0; i < 3; ++i) {
        float r3 = r * r * r;
        r = r * fmaf(-xt, r3, 1.3333334f);
    }
    float rr = r * r;
    double yd = (double)(x * rr);
    double gd = (double)(rr * 0.33333334f);
    double xd = (double)x;
    for (int i = 0; i < 2; ++i) {
        double res = fma(-yd * yd, yd, xd);
        yd = fma(res, gd, yd);
    }
    return (float)yd;
}

__device__ __forceinline__ float lab_f_ref(float t) { return t > 0.008856f ? cbrt_bits(t) : fmaf(7.787f, t, 16.0f / 116.0f); }

__device__ __forceinline__ void lab_ref(float r, float g, float b, float &L, float &A, float &B)
{
    const float X = fmaf(0.1804375f, b, fmaf(0.3575761f, g, 0.4124564f * r));
    const float Y = fmaf(0.0721750f, b, fmaf(0.7151522f, g, 0.2126729f * r));
    const float Z = fmaf(0.9503041f, b, fmaf(0.1191920f, g, 0.0193339f * r));
    const float fx = lab_f_ref(X / 95.0489f);
    const float fy = lab_f_ref(Y / 100.0f);
    const float fz = lab_f_ref(Z / 108.8840f);
    L = fmaf(116.0f, fy, -16.0f);
    A = 500.0f * (fx - fy);
    B = 200.0f * (fy - fz);
}

// ---- kernels ----------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void k_convert(const uint32_t *__restrict__ px, uint64_t n, const float *__restrict__ lut,
                                                 float4 *__restrict__ out, unsigned long long *__restrict__ sink)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t p = px[i];
        float L, a, b;
        const float r = s_lut[p & 255u], g = s_lut[(p >> 8) & 255u], bb = s_lut[(p >> 16) & 255u];
        if (VARIANT == 0) lab_ref(r, g, bb, L, a, b);
        else if (VARIANT == 1) linear100_to_lab(r, g, bb, L, a, b);
        else { L = r; a = g; b = bb; }                              // the loop alone
        if (out) out[i] = make_float4(L, a, b, 0.0f);
        acc += L + a + b;
    }
    if (acc == 123.456f) atomicAdd(sink, 1ull);
}

__global__ void k_compare(const float4 *a, const float4 *b, uint64_t n, unsigned long long *mism)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 x = a[i], y = b[i];
    if (float_to_bits(x.x) != float_to_bits(y.x) || float_to_bits(x.y) != float_to_bits(y.y) || float_to_bits(x.z) != float_to_bits(y.z))
        atomicAdd(mism, 1ull);
}

// every binary32 in [lo_bits, hi_bits): cbrt_cr (device: hardware seed) vs the bit-trick seed
__global__ void k_cbrt_all(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mism)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + lo_bits;
    if (i >= hi_bits) return;
    const float x = bits_to_float((uint32_t)i);
    if (float_to_bits(cbrt_cr(x)) != float_to_bits(cbrt_bits(x))) atomicAdd(mism, 1ull);
}

// every binary32 in [lo, hi): x / c (IEEE) vs div_white
template <int WHICH>
__global__ void k_div_all(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mism)
{
    constexpr float c = WHICH == 0 ? 95.0489f : (WHICH == 1 ? 100.0f : 108.8840f);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + lo_bits;
    if (i >= hi_bits) return;
    const float x = bits_to_float((uint32_t)i);
    if (float_to_bits(div_white<WHICH>(x)) != float_to_bits(x / c)) atomicAdd(mism, 1ull);
}

int main()
{
    const uint64_t n = 1ull << 24;
    std::vector<uint32_t> h(n);
    for (uint64_t i = 0; i < n; ++i) h[i] = (uint32_t)((i * 0x9E3779B1ull) & 0xFFFFFFull) | 0xFF000000u;   // a permutation of all colours
    float lut[256];
    build_srgb_lut100(lut);
    uint32_t *d_px; float *d_lut; float4 *d_a, *d_b; unsigned long long *d_cnt;
    CHECK(hipMalloc(&d_px, n * 4)); CHECK(hipMalloc(&d_lut, 1024)); CHECK(hipMalloc(&d_a, n * 16)); CHECK(hipMalloc(&d_b, n * 16));
    CHECK(hipMalloc(&d_cnt, 64));
    CHECK(hipMemcpy(d_px, h.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lut, lut, 1024, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_cnt, 0, 64));

    hipLaunchKernelGGL(k_convert<0>, dim3(4096), dim3(256), 0, 0, d_px, n, d_lut, d_a, d_cnt);
    hipLaunchKernelGGL(k_convert<1>, dim3(4096), dim3(256), 0, 0, d_px, n, d_lut, d_b, d_cnt);
    hipLaunchKernelGGL(k_compare, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_a, d_b, n, d_cnt + 1);
    // [1e-3, 2]
    const uint32_t lo = float_to_bits(1.0e-3f), hi = float_to_bits(2.0f) + 1u;
    hipLaunchKernelGGL(k_cbrt_all, dim3((hi - lo + 255) / 256), dim3(256), 0, 0, lo, hi, d_cnt + 2);
    // quotients: every binary32 in [2^-20, 128) (the XYZ values are in [0, 109]; smaller ones only meet the linear branch)
    const uint32_t dlo = float_to_bits(9.5367431640625e-7f), dhi = float_to_bits(128.0f);
    hipLaunchKernelGGL(k_div_all<0>, dim3((dhi - dlo + 255) / 256), dim3(256), 0, 0, dlo, dhi, d_cnt + 3);
    hipLaunchKernelGGL(k_div_all<1>, dim3((dhi - dlo + 255) / 256), dim3(256), 0, 0, dlo, dhi, d_cnt + 4);
    hipLaunchKernelGGL(k_div_all<2>, dim3((dhi - dlo + 255) / 256), dim3(256), 0, 0, dlo, dhi, d_cnt + 5);
    unsigned long long cnt[8];
    CHECK(hipMemcpy(cnt, d_cnt, 64, hipMemcpyDeviceToHost));
    printf("Lab of all 2^24 colours, linear100_to_lab (device) vs bit-trick seed + IEEE divisions: %llu mismatches\n", cnt[1]);
    printf("cbrt_cr (device) vs bit-trick seed over every binary32 in [1e-3, 2] (%u values): %llu mismatches\n", hi - lo, cnt[2]);
    printf("div_white vs IEEE division over every binary32 in [2^-20, 128): /95.0489 %llu, /100 %llu, /108.884 %llu mismatches\n", cnt[3], cnt[4], cnt[5]);

    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[3] = {"bit-trick seed + IEEE divisions", "linear100_to_lab (kmg_math.h)", "loop alone"};
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int it = 0; it < 4; ++it) {
                if (v == 0) hipLaunchKernelGGL(k_convert<0>, dim3(4096), dim3(256), 0, 0, d_px, n, d_lut, (float4 *)nullptr, d_cnt);
                if (v == 1) hipLaunchKernelGGL(k_convert<1>, dim3(4096), dim3(256), 0, 0, d_px, n, d_lut, (float4 *)nullptr, d_cnt);
                if (v == 2) hipLaunchKernelGGL(k_convert<2>, dim3(4096), dim3(256), 0, 0, d_px, n, d_lut, (float4 *)nullptr, d_cnt);
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-34s %8.3f ms per 2^26 pixels  (%.1f ps / pixel)\n", names[v], ms, ms * 1e9 / (4.0 * n));
        }
    }
    return (cnt[1] | cnt[2] | cnt[3] | cnt[4] | cnt[5]) ? 1 : 0;
}
