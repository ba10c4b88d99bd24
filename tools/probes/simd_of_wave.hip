// Which SIMD does wavefront w of a 512-lane (and 256-lane) workgroup run on?  HW_REG_HW_ID (gfx9: register 4): wave_id [3:0], simd_id [5:4],
// cu_id [11:8], se_id [15:13].  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_of_wave tools/probes/simd_of_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
    const unsigned id = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
    // keep the workgroup resident for a moment so that several are in flight
    for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(8);
}
int main() {
    for (int block : {256, 512}) {
        const int nb = 512, wpb = block / 64;
        unsigned *d;
        hipMalloc(&d, nb * wpb * 4);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(block), 0, 0, d);
        std::vector<unsigned> h(nb * wpb);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("workgroup of %d lanes: simd id of wavefront 0..%d (first 12 workgroups; cu id in brackets)\n", block, wpb - 1);
        for (int b = 0; b < 12; b++) {
            printf("  wg %2d:", b);
            for (int w = 0; w < wpb; w++) printf(" %u", (h[b * wpb + w] >> 4) & 3);
            printf("   [cu %u se %u]\n", (h[b * wpb] >> 8) & 15, (h[b * wpb] >> 13) & 7);
        }
        // histogram of the pattern (simd of wave w == w % 4 ?)
        int same = 0, pair = 0;
        for (int b = 0; b < nb; b++) {
            bool m = true, p = true;
            for (int w = 0; w < wpb; w++) {
                m = m && (((h[b * wpb + w] >> 4) & 3) == ((h[b * wpb] >> 4) + w) % 4);
                if (w >= 4) p = p && (((h[b * wpb + w] >> 4) & 3) == ((h[b * wpb + w - 4] >> 4) & 3));
            }
            same += m; pair += p;
        }
        printf("  round-robin from wavefront 0's SIMD in %d of %d workgroups; w and w + 4 on one SIMD in %d\n", same, nb, pair);
        hipFree(d);
    }
    return 0;
}
