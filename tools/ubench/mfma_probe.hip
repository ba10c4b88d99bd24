// tools/ubench/mfma_probe.hip -- operand layout and issue cost of v_mfma_f64_4x4x4_4b_f64 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void probe(double *out) {
    const int l = threadIdx.x;
    const double x = (double)(1 << (l % 16)) + 65536.0 * (l / 16);  // distinct power-of-two tags inside each block of 16
    double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);      // A = x, B = ones
    double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);     // A = ones, B = d1
    double e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x, 0.0, 0, 0, 0);      // A = ones, B = x
    double e2 = __builtin_amdgcn_mfma_f64_4x4x4f64(e1, 1.0, 0.0, 0, 0, 0);     // A = e1, B = ones
    out[l] = d1; out[64 + l] = d2; out[128 + l] = e1; out[192 + l] = e2;
}

template <int MODE>
__global__ __launch_bounds__(64) void cost(double *out, unsigned long long *cyc) {
    double a[8], acc[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; acc[i] = 0; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], 1.0, acc[i], 0, 0, 0);
            if (MODE == 1) { acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], 1.0, 0.0, 0, 0, 0); a[i] = a[i] * 1.0000001 + 1e-9; a[(i + 3) & 7] += 1e-12; }  // mfma + 3 VALU
            if (MODE == 2) { a[i] = a[i] * 1.0000001 + 1e-9; a[(i + 3) & 7] += 1e-12; }                                                                       // the 3 VALU alone
            if (MODE == 3) { double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], 1.0, 0.0, 0, 0, 0); acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d, 0.0, 0, 0, 0); }  // dependent pair
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 8; i++) s += acc[i] + a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 1024 * 64 * 8)); CHECK(hipMalloc(&cyc, 1024 * 8));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
    double h[256]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    const char *names[4] = {"d1 = mfma(A=x,B=1)", "d2 = mfma(A=1,B=d1)", "e1 = mfma(A=1,B=x)", "e2 = mfma(A=e1,B=1)"};
    for (int r = 0; r < 4; r++) {
        printf("%s, lanes 0..15 (block 0), tags are 2^lane:\n ", names[r]);
        for (int l = 0; l < 16; l++) printf(" %.0f", h[64 * r + l]);
        printf("\n lanes 16..19: %.0f %.0f %.0f %.0f\n", h[64 * r + 16], h[64 * r + 17], h[64 * r + 18], h[64 * r + 19]);
    }
    const char *cn[4] = {"mfma 4x4x4 f64, 8 independent accumulators", "mfma + 3 fp64 VALU", "3 fp64 VALU alone", "dependent mfma pair"};
    for (int m = 0; m < 4; m++) {
        if (m == 0) hipLaunchKernelGGL(cost<0>, dim3(1024), dim3(64), 0, 0, out, cyc);
        if (m == 1) hipLaunchKernelGGL(cost<1>, dim3(1024), dim3(64), 0, 0, out, cyc);
        if (m == 2) hipLaunchKernelGGL(cost<2>, dim3(1024), dim3(64), 0, 0, out, cyc);
        if (m == 3) hipLaunchKernelGGL(cost<3>, dim3(1024), dim3(64), 0, 0, out, cyc);
        CHECK(hipDeviceSynchronize());
        unsigned long long c[1024]; CHECK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < 1024; i++) avg += c[i]; avg /= 1024;
        printf("%-46s %.2f clk per loop body (one wave per SIMD)\n", cn[m], avg / (4096.0 * 8));
    }
    return 0;
}
