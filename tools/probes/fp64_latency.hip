// Dependent-issue latency and throughput of the fp64 vector operations K2f's filter recurrence is made of (one wavefront on a SIMD, s_memtime).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o /tmp/fp64_latency tools/probes/fp64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP, int CHAINS>
__global__ void probe(double *io, unsigned long long *cyc, int iters) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = io[threadIdx.x + 64 * c];
    const double k = io[1000], m = io[1001];
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(k));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(m), "v"(k));
                if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&x[c]) : "v"((float)k));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int *)&x[c]) : "v"((int)i) : );
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int c = 0; c < CHAINS; c++) io[threadIdx.x + 64 * c] = x[c];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int CHAINS>
void run(const char *name, double *d, unsigned long long *c, int waves) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<OP, CHAINS>), dim3(1), dim3(64 * waves), 0, 0, d, c, iters);
    unsigned long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-12s %d chain(s), %d wavefront(s) per workgroup: %.2f cycles per instruction of a chain (s_memtime ticks, 100 MHz? see below)\n", name, CHAINS, waves,
           (double)h / (iters * 16.0 * CHAINS));
}
int main() {
    double *d; unsigned long long *c;
    hipMalloc(&d, 8192 * 8); hipMalloc(&c, 64);
    std::vector<double> h(8192, 1.0); h[1000] = 1e-9; h[1001] = 1.0000001;
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int waves : {1, 4, 8}) {
        run<0, 1>("v_add_f64", d, c, waves); run<0, 2>("v_add_f64", d, c, waves); run<0, 4>("v_add_f64", d, c, waves);
        run<1, 1>("v_mul_f64", d, c, waves); run<1, 4>("v_mul_f64", d, c, waves);
        run<2, 1>("v_fma_f64", d, c, waves); run<2, 2>("v_fma_f64", d, c, waves); run<2, 4>("v_fma_f64", d, c, waves);
        run<3, 1>("v_add_f32", d, c, waves); run<3, 4>("v_add_f32", d, c, waves);
        run<4, 1>("v_cndmask", d, c, waves); run<4, 4>("v_cndmask", d, c, waves);
    }
    // wall-clock calibration of the counter: a long run timed with events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<2, 1>), dim3(1), dim3(64), 0, 0, d, c, 200000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("calibration: %llu ticks in %.3f ms = %.1f MHz; %d dependent v_fma_f64 => %.2f ns each\n", hc, ms, hc / ms / 1000.0, 200000 * 16, ms * 1e6 / (200000.0 * 16));
    return 0;
}
