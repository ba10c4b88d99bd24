// mix.hip -- maxiMix::stereo / quad / ambisonic over a voice bank + the mixdown over voices.
//
// Reference, src/maximilian.cpp: stereo C:503-509, quad C:512-522, ambisonic C:525-541.  Every
// bus output of the reference is `input * gain_c(x[,y[,z]])` with the gain built from sqrt and
// + - * only, so gains and per-voice products are IEEE-exact here (sqrt is correctly rounded on
// gfx950).  The user-side `mix += ...` over voices (e.g. 15.polysynth/main.cpp:67) is ours: a
// fixed-shape reduction (1024 strided partial sums per row and channel, a butterfly inside each
// wavefront, then the 16 wavefront sums left to right), deterministic run to run but not the
// reference's left-to-right order => the mix carries the tolerance stated in DESIGN.md, the
// per-voice bus signals (optional d_bus output) stay bit-exact.
//
// ambisonic quirks kept (C:530-531): `if (z>1) y=1; if (z<0) y=0;` -- z itself is never
// clamped and y is overwritten; eight[0..3] = input*(sqrt(..)*1.0 - z) (precedence as written).
//
// K3 (HBM-read bound): one 1024-thread workgroup per sample row; every wavefront load is
// 512 contiguous bytes of the row; the gains are C [V] arrays (L2 resident).
#include "mxg_common.h"

namespace mxg {
namespace {

constexpr int kMixThreads = 1024;

template <int C>
__global__ void bus_gains_kernel(size_t V, const double *__restrict__ px,
                                 const double *__restrict__ py, const double *__restrict__ pz,
                                 double *__restrict__ gains) {
    size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double x = px[v];
    if (x > 1) x = 1;
    if (x < 0) x = 0;
    if constexpr (C == 2) {  // C:504-507
        gains[v] = sqrt(1.0 - x);
        gains[V + v] = sqrt(x);
    } else {
        double y = py[v];
        if (y > 1) y = 1;
        if (y < 0) y = 0;
        if constexpr (C == 4) {  // C:517-520
            gains[v] = sqrt((1.0 - x) * y);
            gains[V + v] = sqrt((1.0 - x) * (1.0 - y));
            gains[2 * V + v] = sqrt(x * y);
            gains[3 * V + v] = sqrt(x * (1.0 - y));
        } else {  // C:530-539
            const double z = pz[v];
            if (z > 1) y = 1;
            if (z < 0) y = 0;
            gains[v] = (sqrt((1.0 - x) * y) * 1.0 - z);
            gains[V + v] = (sqrt((1.0 - x) * (1.0 - y)) * 1.0 - z);
            gains[2 * V + v] = (sqrt(x * y) * 1.0 - z);
            gains[3 * V + v] = (sqrt(x * (1.0 - y)) * 1.0 - z);
            gains[4 * V + v] = (sqrt((1.0 - x) * y) * z);
            gains[5 * V + v] = (sqrt((1.0 - x) * (1.0 - y)) * z);
            gains[6 * V + v] = sqrt((x * y) * z);
            gains[7 * V + v] = sqrt((x * (1.0 - y)) * z);
        }
    }
}

// BUS: also write the per-voice bus signals bus[n][c][v] (what the reference leaves in
// two/four/eight for voice v at sample n).  R sample rows per workgroup share one read of the gains
// (an L2 stream as large as the HBM stream for stereo); every row keeps its own accumulators and the
// per-thread order v = t, t+1024, ... so the bits do not depend on R.
template <int C, bool BUS, int R>
__global__ __launch_bounds__(kMixThreads) void mix_bus_kernel(
    size_t V, size_t N, const double *__restrict__ in, const double *__restrict__ gains,
    double *__restrict__ bus, double *__restrict__ mix) {
    __shared__ double s_red[R][C][kMixThreads / 64];
    const size_t n0 = (size_t)blockIdx.x * R;
    const double *row[R];
    double *brow[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const size_t n = (n0 + r < N) ? n0 + r : N - 1;  // a surplus row re-reads the last one, never stored
        row[r] = in + n * V;
        brow[r] = BUS ? bus + n * C * V : nullptr;
    }
    double acc[R][C];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < C; c++) acc[r][c] = 0.0;
    constexpr int U = (C == 2) ? (R == 1 ? 8 : 4) : (C == 4 ? 4 / R + (R > 4) : 2 / R + (R > 1));
    static_assert(U >= 1, "unroll");
    size_t v = threadIdx.x;
    for (; v + (U - 1) * kMixThreads < V; v += U * kMixThreads) {
        double x[R][U], g[U][C];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int r = 0; r < R; r++) x[r][u] = row[r][v + u * kMixThreads];
#pragma unroll
            for (int c = 0; c < C; c++) g[u][c] = gains[(size_t)c * V + v + u * kMixThreads];
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const double p = x[r][u] * g[u][c];
                    if constexpr (BUS) {
                        if (n0 + r < N) brow[r][(size_t)c * V + v + u * kMixThreads] = p;
                    }
                    acc[r][c] += p;
                }
            }
    }
    for (; v < V; v += kMixThreads) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double x = row[r][v];
#pragma unroll
            for (int c = 0; c < C; c++) {
                const double p = x * gains[(size_t)c * V + v];
                if constexpr (BUS) {
                    if (n0 + r < N) brow[r][(size_t)c * V + v] = p;
                }
                acc[r][c] += p;
            }
        }
    }
    // butterfly inside the wavefront (every lane ends with the same sum), then 16 wave sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < C; c++) {
            double s = acc[r][c];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
            if (lane == 0) s_red[r][c][wave] = s;
        }
    __syncthreads();
    if (threadIdx.x < R * C) {
        const int r = threadIdx.x / C, c = threadIdx.x % C;
        if (n0 + r < N) {
            double s = s_red[r][c][0];
            for (int w = 1; w < kMixThreads / 64; w++) s += s_red[r][c][w];
            mix[C * (n0 + r) + c] = s;
        }
    }
}

template <int C>
int launch_bus(size_t V, size_t N, const double *d_in, const double *d_x, const double *d_y,
               const double *d_z, double *d_bus, double *d_mix, hipStream_t st) {
    double *g_gains = nullptr;  // per-stream scratch for the per-voice gains
    if (int s = scratch_get(SCR_MIX_GAINS, st, sizeof(double) * C * (V ? V : 1), (void **)&g_gains)) return s;
    if (V)
        hipLaunchKernelGGL((bus_gains_kernel<C>), dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st,
                           V, d_x, d_y, d_z, g_gains);
    const int rows = tune_get("mix_rows");
    KernelTimer kt("mix_bus_kernel", st);
    if (d_bus)
        hipLaunchKernelGGL((mix_bus_kernel<C, true, 1>), dim3((unsigned)N), dim3(kMixThreads), 0, st, V, N,
                           d_in, g_gains, d_bus, d_mix);
    else if (rows == 2 && C == 2)
        hipLaunchKernelGGL((mix_bus_kernel<C, false, 2>), dim3((unsigned)((N + 1) / 2)), dim3(kMixThreads), 0, st, V,
                           N, d_in, g_gains, d_bus, d_mix);
    else
        hipLaunchKernelGGL((mix_bus_kernel<C, false, 1>), dim3((unsigned)N), dim3(kMixThreads), 0, st, V, N,
                           d_in, g_gains, d_bus, d_mix);
    return check_hip(hipGetLastError(), "mix_bus launch");
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

int mxg_mix_bus(int channels, size_t V, size_t N, const double *d_in, const double *d_x,
                const double *d_y, const double *d_z, double *d_bus, double *d_mix, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(channels == 2 || channels == 4 || channels == 8,
                "channels must be 2 (stereo), 4 (quad) or 8 (ambisonic)");
    MXG_REQUIRE(d_in && d_x && d_mix, "null device pointer");
    MXG_REQUIRE(channels < 4 || d_y, "quad/ambisonic need d_y");
    MXG_REQUIRE(channels < 8 || d_z, "ambisonic needs d_z");
    if (N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    switch (channels) {
        case 2: return launch_bus<2>(V, N, d_in, d_x, d_y, d_z, d_bus, d_mix, st);
        case 4: return launch_bus<4>(V, N, d_in, d_x, d_y, d_z, d_bus, d_mix, st);
        default: return launch_bus<8>(V, N, d_in, d_x, d_y, d_z, d_bus, d_mix, st);
    }
}

int mxg_mix_stereo(size_t V, size_t N, const double *d_in, const double *d_pan, double *d_mix,
                   void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_in && d_pan && d_mix, "null device pointer");
    return mxg_mix_bus(2, V, N, d_in, d_pan, nullptr, nullptr, nullptr, d_mix, stream);
}

}  // extern "C"
