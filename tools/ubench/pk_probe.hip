// tools/ubench/pk_probe.hip -- what a packed fp32 instruction costs on gfx950 by FORM: plain, with operand swizzles (op_sel / op_sel_hi), with
// negations, the fused kernel's butterfly block itself, against the unpacked equivalents -- at one and two wavefronts per SIMD.
// (round 5: the fused FFT+MFCC kernel issues 216 packed instructions per frame, most of them swizzled; if a swizzle costs passes, the
// reference's butterflies are cheaper as plain v_mul / v_add.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define EIGHT(INS)                                                                                                   \
    asm volatile(INS(%0) "\n\t" INS(%1) "\n\t" INS(%2) "\n\t" INS(%3) "\n\t" INS(%4) "\n\t" INS(%5) "\n\t" INS(%6) "\n\t" INS(%7) \
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]) \
                 : "v"(a), "v"(b), "v"(c))
#define I_PLAIN(D) "v_pk_mul_f32 " #D ", %8, %9"
#define I_SELHI(D) "v_pk_mul_f32 " #D ", %8, %9 op_sel_hi:[0,1]"
#define I_SWZ(D) "v_pk_mul_f32 " #D ", %8, %9 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]"
#define I_ADDNEG(D) "v_pk_add_f32 " #D ", %8, %9 neg_lo:[0,1] neg_hi:[0,1]"
#define I_ADD(D) "v_pk_add_f32 " #D ", %8, %9"
#define I_FMA(D) "v_pk_fma_f32 " #D ", %8, %9, %10"
#define I_FMASWZ(D) "v_pk_fma_f32 " #D ", %8, %9, %10 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"

template <int OP>
__global__ void probe(float *out, unsigned long long *cyc, int iters) {
    v2f a = {1.0001f + threadIdx.x * 1e-6f, 0.9999f}, b = {0.99991f, 1.00002f}, c = {1e-9f, -1e-9f};
    v2f d[8];
    float s[16];
    for (int i = 0; i < 16; i++) s[i] = 1.0f + i * 1e-3f + threadIdx.x * 1e-6f;
    for (int i = 0; i < 8; i++) d[i] = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if constexpr (OP == 0) EIGHT(I_PLAIN);
        if constexpr (OP == 1) EIGHT(I_SELHI);
        if constexpr (OP == 2) EIGHT(I_SWZ);
        if constexpr (OP == 3) EIGHT(I_ADDNEG);
        if constexpr (OP == 4) EIGHT(I_ADD);
        if constexpr (OP == 5) EIGHT(I_FMA);
        if constexpr (OP == 6) EIGHT(I_FMASWZ);
        if constexpr (OP == 7) {  // 16 unpacked multiplies = the work of 8 packed ones
            asm volatile("v_mul_f32 %0, %16, %0\n\tv_mul_f32 %1, %16, %1\n\tv_mul_f32 %2, %16, %2\n\tv_mul_f32 %3, %16, %3\n\t"
                         "v_mul_f32 %4, %16, %4\n\tv_mul_f32 %5, %16, %5\n\tv_mul_f32 %6, %16, %6\n\tv_mul_f32 %7, %16, %7\n\t"
                         "v_mul_f32 %8, %16, %8\n\tv_mul_f32 %9, %16, %9\n\tv_mul_f32 %10, %16, %10\n\tv_mul_f32 %11, %16, %11\n\t"
                         "v_mul_f32 %12, %16, %12\n\tv_mul_f32 %13, %16, %13\n\tv_mul_f32 %14, %16, %14\n\tv_mul_f32 %15, %16, %15"
                         : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]),
                           "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15])
                         : "v"(b.x));
        }
        if constexpr (OP == 8) {  // the butterfly pair of the fused kernel (mxg_spectral.h bfly2): 10 packed instructions
            v2f p1, q1, p2, q2;
            asm volatile("s_nop 0\n\t"
                "v_pk_mul_f32 %[p1], %[w1], %[k1] op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 %[q1], %[w1], %[k1] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
                "v_pk_mul_f32 %[p2], %[w2], %[k2] op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 %[q2], %[w2], %[k2] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
                "v_pk_add_f32 %[p1], %[p1], %[q1]\n\t"
                "v_pk_add_f32 %[p2], %[p2], %[q2]\n\t"
                "v_pk_add_f32 %[k1], %[j1], %[p1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 %[j1], %[j1], %[p1]\n\t"
                "v_pk_add_f32 %[k2], %[j2], %[p2] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 %[j2], %[j2], %[p2]\n\t"
                "s_nop 0"
                : [j1] "+v"(d[0]), [k1] "+v"(d[1]), [j2] "+v"(d[2]), [k2] "+v"(d[3]), [p1] "=&v"(p1), [q1] "=&v"(q1), [p2] "=&v"(p2), [q2] "=&v"(q2)
                : [w1] "v"(a), [w2] "v"(b));
        }
        if constexpr (OP == 9) {  // the same two butterflies as 20 unpacked instructions (L/fft.cpp:184-192 op for op)
            float jr = d[0].x, ji = d[0].y, kr = d[1].x, ki = d[1].y, jr2 = d[2].x, ji2 = d[2].y, kr2 = d[3].x, ki2 = d[3].y, t0_, t1_, t2_, t3_;
            asm volatile("v_mul_f32 %8, %12, %2\n\tv_mul_f32 %9, %13, %3\n\tv_mul_f32 %10, %12, %3\n\tv_mul_f32 %11, %13, %2\n\t"
                         "v_sub_f32 %8, %8, %9\n\tv_add_f32 %10, %10, %11\n\t"
                         "v_sub_f32 %2, %0, %8\n\tv_sub_f32 %3, %1, %10\n\tv_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %10\n\t"
                         "v_mul_f32 %8, %14, %6\n\tv_mul_f32 %9, %15, %7\n\tv_mul_f32 %10, %14, %7\n\tv_mul_f32 %11, %15, %6\n\t"
                         "v_sub_f32 %8, %8, %9\n\tv_add_f32 %10, %10, %11\n\t"
                         "v_sub_f32 %6, %4, %8\n\tv_sub_f32 %7, %5, %10\n\tv_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %10"
                         : "+v"(jr), "+v"(ji), "+v"(kr), "+v"(ki), "+v"(jr2), "+v"(ji2), "+v"(kr2), "+v"(ki2), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)
                         : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
            d[0] = v2f{jr, ji}; d[1] = v2f{kr, ki}; d[2] = v2f{jr2, ji2}; d[3] = v2f{kr2, ki2};
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += d[i].x + d[i].y;
    for (int i = 0; i < 16; i++) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_iter, float *out, unsigned long long *cyc) {
    const int iters = 4096;
    for (int wps = 1; wps <= 2; wps++) {
        const int threads = 256 * wps, blocks = 256;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        static unsigned long long h[256 * 8];
        CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * threads / 64, hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < blocks * threads / 64; i++) avg += h[i]; avg /= blocks * threads / 64;
        printf("%-52s %d wave/SIMD  %7.2f counter ticks per instr per wave, %6.2f per SIMD   kernel %.3f ms => %.2f ns per instr per SIMD\n", name, wps,
               avg / (iters * (double)per_iter), avg / (iters * (double)per_iter) / wps, ms, ms * 1e6 / (iters * (double)per_iter * wps));
    }
}

int main() {
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    run<0>("v_pk_mul_f32 plain", 8, out, cyc);
    run<1>("v_pk_mul_f32 op_sel_hi:[0,1]", 8, out, cyc);
    run<2>("v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0] neg_lo", 8, out, cyc);
    run<3>("v_pk_add_f32 neg_lo neg_hi", 8, out, cyc);
    run<4>("v_pk_add_f32 plain", 8, out, cyc);
    run<5>("v_pk_fma_f32 plain", 8, out, cyc);
    run<6>("v_pk_fma_f32 swizzled", 8, out, cyc);
    run<7>("v_mul_f32 x16 (dependent pairs of 8)", 16, out, cyc);
    run<8>("bfly2 block: 10 packed (per packed instr)", 10, out, cyc);
    run<9>("the same two butterflies unpacked: 20 (per instr)", 20, out, cyc);
    return 0;
}
