// tools/ubench/sqrt_probe.hip -- exhaustive check of candidate correctly-rounded sqrtf sequences on gfx950.
// The fused FFT+MFCC kernel needs sqrtf(x) bit-identical to the IEEE (correctly rounded) result for every magnitude.
// exact_sqrtf (spectral.hip) gets there from v_sqrt_f32 with a two-sided residual test (11 VALU).  A Markstein-style
// sequence from v_rsq_f32 (g = x*r, h = r/2, g + (x - g*g)*h) is 5 VALU, but whether its last rounding is always the
// correct one depends on the actual values the hardware's rsq returns -- so every float in [2^-96, FLT_MAX] is tried.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float ref_residual(float x) {  // exact_sqrtf's fast path
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = rd <= 0.0f ? sd : s;
    r = ru > 0.0f ? su : r;
    return r;
}
__device__ __forceinline__ float cand_m1(float x) {  // rsq, one correction
    const float r = __builtin_amdgcn_rsqf(x);
    const float g = x * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float cand_m2(float x) {  // v_sqrt, correction with h from rsq
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float cand_m3(float x) {  // rsq, refine g and h once (Newton-coupled), then the correction
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

__global__ void probe(unsigned long long *bad, uint32_t *first) {
    const uint32_t lo = 0x0F800000u, hi = 0x7F800000u;  // [2^-96, inf)
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long b[5] = {0, 0, 0, 0, 0};
    for (uint64_t v = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; v < hi; v += stride) {
        const float x = __uint_as_float((uint32_t)v);
        const float t = (float)sqrt((double)x);  // double sqrt is correctly rounded; no double-rounding case exists for sqrt
        const float c[4] = {ref_residual(x), cand_m1(x), cand_m2(x), cand_m3(x)};
        for (int i = 0; i < 4; i++)
            if (__float_as_uint(c[i]) != __float_as_uint(t)) {
                if (b[i] == 0 && atomicAdd(&bad[8 + i], 1ull) < 4) first[i * 4 + (atomicAdd(&bad[12 + i], 1ull) & 3)] = (uint32_t)v;
                b[i]++;
            }
        b[4]++;
    }
    for (int i = 0; i < 5; i++) atomicAdd(&bad[i], b[i]);
}

int main() {
    unsigned long long *d_bad; uint32_t *d_first;
    CHECK(hipMalloc(&d_bad, 16 * 8)); CHECK(hipMalloc(&d_first, 16 * 4));
    CHECK(hipMemset(d_bad, 0, 16 * 8)); CHECK(hipMemset(d_first, 0, 16 * 4));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d_bad, d_first);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad[16]; uint32_t first[16];
    CHECK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost));
    const char *name[4] = {"v_sqrt + two-sided residual test (exact_sqrtf)", "rsq: g=x*r, h=r/2, g+(x-g*g)*h", "v_sqrt g, h from rsq, g+(x-g*g)*h",
                           "rsq + one coupled Newton step + correction"};
    printf("%llu floats in [2^-96, FLT_MAX] against (float)sqrt((double)x):\n", bad[4]);
    for (int i = 0; i < 4; i++) {
        printf("  %-50s %llu wrong", name[i], bad[i]);
        if (bad[i]) { printf("  e.g."); for (int k = 0; k < 4 && k < (int)bad[i]; k++) printf(" 0x%08x", first[i * 4 + k]); }
        printf("\n");
    }
    return 0;
}
