// mix.hip -- maxiMix::stereo over a voice bank + the mixdown over voices (gfx950).
//
// Reference: maxiMix::stereo, src/maximilian.cpp:503-509:
//     two[0] = input*sqrt(1.0-x);  two[1] = input*sqrt(x);       (x clamped to [0,1])
// and the user-side `mix += ...` over voices (e.g. 15.polysynth/main.cpp:67).  The per-voice
// products are IEEE-exact (sqrt is correctly rounded on gfx950); the SUM over voices is ours:
// a fixed-shape reduction (1024 strided partial sums per row, then a binary tree), so it is
// deterministic run to run but not the reference's left-to-right order => the mix carries the
// tolerance stated in DESIGN.md, the per-voice signals stay bit-exact.
//
// K3 (HBM-read bound): one 1024-thread workgroup per sample row; every wavefront load is
// 512 contiguous bytes of the row; the gains are two [V] arrays (L2/L3 resident, 1 MB).
#include "mxg_common.h"

namespace mxg {
namespace {

constexpr int kMixThreads = 1024;

// gains[0][v] = sqrt(1-x), gains[1][v] = sqrt(x)   (C:504-507)
__global__ void pan_gains_kernel(size_t V, const double *__restrict__ pan,
                                 double *__restrict__ gains) {
    size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double x = pan[v];
    if (x > 1) x = 1;
    if (x < 0) x = 0;
    gains[v] = sqrt(1.0 - x);
    gains[V + v] = sqrt(x);
}

__global__ __launch_bounds__(kMixThreads) void mix_stereo_kernel(
    size_t V, const double *__restrict__ in, const double *__restrict__ gains,
    double *__restrict__ mix) {
    __shared__ double s_red[2 * kMixThreads];
    const size_t n = blockIdx.x;
    const double *row = in + n * V;
    double l = 0.0, r = 0.0;
    // 4 independent loads in flight per thread; the accumulation order per thread stays
    // v = t, t+1024, t+2048, ... so the result does not depend on the unrolling.
    size_t v = threadIdx.x;
    for (; v + 3 * kMixThreads < V; v += 4 * kMixThreads) {
        const double x0 = row[v], x1 = row[v + kMixThreads], x2 = row[v + 2 * kMixThreads],
                     x3 = row[v + 3 * kMixThreads];
        const double a0 = gains[v], a1 = gains[v + kMixThreads], a2 = gains[v + 2 * kMixThreads],
                     a3 = gains[v + 3 * kMixThreads];
        const double b0 = gains[V + v], b1 = gains[V + v + kMixThreads], b2 = gains[V + v + 2 * kMixThreads],
                     b3 = gains[V + v + 3 * kMixThreads];
        l += x0 * a0; r += x0 * b0;
        l += x1 * a1; r += x1 * b1;
        l += x2 * a2; r += x2 * b2;
        l += x3 * a3; r += x3 * b3;
    }
    for (; v < V; v += kMixThreads) {
        double x = row[v];
        l += x * gains[v];
        r += x * gains[V + v];
    }
    s_red[threadIdx.x] = l;
    s_red[kMixThreads + threadIdx.x] = r;
    __syncthreads();
    for (int s = kMixThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_red[threadIdx.x] += s_red[threadIdx.x + s];
            s_red[kMixThreads + threadIdx.x] += s_red[kMixThreads + threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mix[2 * n] = s_red[0];
        mix[2 * n + 1] = s_red[kMixThreads];
    }
}

double *g_gains = nullptr;
size_t g_gains_cap = 0;

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" int mxg_mix_stereo(size_t V, size_t N, const double *d_in, const double *d_pan,
                              double *d_mix, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_in && d_pan && d_mix, "null device pointer");
    if (N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    if (g_gains_cap < 2 * V) {  // grow-only scratch for the per-voice gains
        if (g_gains) MXG_HIP(hipFree(g_gains));
        g_gains = nullptr;
        g_gains_cap = 0;
        MXG_HIP(hipMalloc(&g_gains, sizeof(double) * 2 * (V ? V : 1)));
        g_gains_cap = 2 * V;
    }
    if (V)
        hipLaunchKernelGGL(pan_gains_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V,
                           d_pan, g_gains);
    hipLaunchKernelGGL(mix_stereo_kernel, dim3((unsigned)N), dim3(kMixThreads), 0, st, V, d_in,
                       g_gains, d_mix);
    return check_hip(hipGetLastError(), "mix_stereo launch");
}
