// tools/ubench/ubench.hip -- issue cost of single instructions on gfx950 (cycles per wave64 instruction on one SIMD).
// One wavefront per SIMD (1024 workgroups of 64 lanes); each kernel runs ITER iterations of 16 independent copies of the
// instruction under test (independent => throughput, not latency) plus a DEP variant (a dependent chain => latency).
// cycles/instr = elapsed * clock / (ITER * 16), clock from s_memtime deltas measured in the same kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;

template <int OP, bool DEP>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc) {
    double a[16];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; u[i] = threadIdx.x * 7u + i; }
    const double c = 1.0000001;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int j = DEP ? 0 : i;
            if constexpr (OP == 0) a[j] = a[j] + c;                                        // v_add_f64
            if constexpr (OP == 1) a[j] = a[j] * c;                                        // v_mul_f64
            if constexpr (OP == 2) a[j] = __builtin_fma(a[j], c, c);                       // v_fma_f64
            if constexpr (OP == 3) u[j] = (unsigned)__builtin_amdgcn_update_dpp((int)u[j], (int)u[(j + 1) & 15], 0x140, 0xf, 0xc, false);  // dpp row_mirror, bank-masked
            if constexpr (OP == 4) u[j] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[j], 0xB1, 0xf, 0xf, true);   // dpp quad_perm
            if constexpr (OP == 5) { auto r = __builtin_amdgcn_permlane32_swap(u[j], u[(j + 1) & 15], false, false); u[j] = r[0]; u[(j + 1) & 15] = r[1]; }
            if constexpr (OP == 6) { auto r = __builtin_amdgcn_permlane16_swap(u[j], u[(j + 1) & 15], false, false); u[j] = r[0]; u[(j + 1) & 15] = r[1]; }
            if constexpr (OP == 7) u[j] = u[j] * 3u + 1u;                                   // v_mad_u32_u24-ish integer
            if constexpr (OP == 8) { float f = __uint_as_float(u[j]); f = f * 1.0001f; u[j] = __float_as_uint(f); }  // v_mul_f32
            if constexpr (OP == 9) u[j] = (unsigned)__builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 32) << 2), (int)u[j]);  // ds_bpermute
            if constexpr (OP == 10) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {__uint_as_float(u[j]), __uint_as_float(u[(j + 8) & 15])}; v = v * (f2){1.0001f, 0.9999f}; u[j] = __float_as_uint(v.x); u[(j + 8) & 15] = __float_as_uint(v.y); }  // v_pk_mul_f32
            if constexpr (OP == 11) a[j] = (double)(int)a[j];                              // cvt_i32_f64 + cvt_f64_i32
            if constexpr (OP == 12) a[j] = floor(a[j]);                                     // v_floor_f64
            if constexpr (OP == 13) { float f = __uint_as_float(u[j]); f = __builtin_amdgcn_sqrtf(f); u[j] = __float_as_uint(f); }  // v_sqrt_f32
            if constexpr (OP == 14) a[j] = (double)__uint_as_float(u[j]) + a[j];            // cvt_f64_f32 + add
            if constexpr (OP == 15) u[j] = (unsigned)__shfl_xor((int)u[j], 16);             // what __shfl_xor compiles to
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { s += a[i]; x ^= u[i]; }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_block = 64;  // 64: one wave per SIMD, 256: four
template <int OP, bool DEP>
void run(const char *name, double *out, unsigned long long *cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP, DEP>), dim3(1024), dim3(g_block), 0, 0, out, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, DEP>), dim3(1024), dim3(g_block), 0, 0, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[1024]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 1024; i++) avg += (double)h[i]; avg /= 1024;
    // per-SIMD throughput: (waves per SIMD) instructions retire per measured per-wave interval
    printf("%-34s %s  %d wave/SIMD  %7.2f clk per instr per wave  %6.2f clk per instr per SIMD   (%.3f ms)\n", name, DEP ? "dep  " : "indep",
           g_block / 64, avg / (ITER * 16.0), avg / (ITER * 16.0) / (g_block / 64), ms);
}

int main(int argc, char **argv) {
    if (argc > 1) g_block = atoi(argv[1]);
    double *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 1024 * 256 * 8)); CHECK(hipMalloc(&cyc, 1024 * 8));
#define R(OP, NAME) run<OP, false>(NAME, out, cyc); run<OP, true>(NAME, out, cyc);
    R(0, "v_add_f64") R(1, "v_mul_f64") R(2, "v_fma_f64") R(3, "v_mov_b32 dpp row_mirror bank_mask") R(4, "v_mov_b32 dpp quad_perm")
    R(5, "v_permlane32_swap_b32") R(6, "v_permlane16_swap_b32") R(7, "v_mul/add u32") R(8, "v_mul_f32") R(9, "ds_bpermute_b32")
    R(10, "v_pk_mul_f32") R(11, "cvt f64<->i32 pair") R(12, "v_floor_f64") R(13, "v_sqrt_f32") R(14, "cvt_f64_f32 + add_f64") R(15, "__shfl_xor 16")
    return 0;
}
