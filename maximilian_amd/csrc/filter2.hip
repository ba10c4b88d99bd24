// filter2.hip -- the reference's newer per-sample filters as voice banks:
//   maxiDCBlocker::play  H:1261-1266      ym1 = input - xm1 + R*ym1
//   maxiSVF::play        H:1303-1317      Cytomic trapezoidal SVF, mixed lp/bp/hp/notch output
//   maxiBiquad::play     H:1360-1367      direct form II
// Coefficients (maxiSVF::setParams H:1320-1332: tan; maxiBiquad::set H:1376-1478: tan, pow, sqrt) are
// evaluated on the HOST with the host libm by the reference's expressions (mxg_svf_coeffs_host,
// mxg_biquad_coeffs_host) and uploaded, exactly like lores/bandpass in voice.hip: the recurrences are
// then + - * only and bit-exact.  One lane = one filter, state in VGPRs for the block, inputs
// software-pipelined a chunk ahead of the output stores (see voice.hip for why).
// HBM: 8 B in + 8 B out per sample (K2 class, read+write bound).
#include <math.h>

#include "mxg_common.h"

namespace mxg {
namespace {

template <int KIND>
__global__ void __launch_bounds__(256) filter2_kernel(size_t V, size_t N, const double *__restrict__ in,
                                                      const double *__restrict__ coef, double *__restrict__ st,
                                                      double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double s0 = st[v], s1 = st[V + v], s2 = st[2 * V + v];
    // coefficient rows: DC {R}; SVF {g1,g2,g3,g4,k,lpmix,bpmix,hpmix,notchmix}; biquad {a0,a1,a2,b1,b2}
    constexpr int NC = KIND == 0 ? 1 : (KIND == 1 ? 9 : 5);
    double c[NC];
#pragma unroll
    for (int r = 0; r < NC; r++) c[r] = coef[(size_t)r * V + v];
    const double *ip = in + v;
    double *op = out + v;
    constexpr int U = 8;
    double xn[U];
#pragma unroll
    for (int i = 0; i < U; i++) xn[i] = ip[((size_t)i < N ? (size_t)i : N - 1) * V];
    for (size_t n0 = 0; n0 < N; n0 += U) {
        double xc[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            xc[i] = xn[i];
            const size_t m = (n0 + U + i < N) ? n0 + U + i : N - 1;  // clamped: no branch, surplus unused
            xn[i] = ip[m * V];
        }
#pragma unroll
        for (int i = 0; i < U; i++) {
            if (n0 + i >= N) break;
            const double x = xc[i];
            double o;
            if constexpr (KIND == 0) {  // s0 = xm1, s1 = ym1
                s1 = x - s0 + c[0] * s1;
                s0 = x;
                o = s1;
            } else if constexpr (KIND == 1) {  // s0 = v0z, s1 = v1, s2 = v2
                const double v1z = s1;
                const double v2z = s2;
                const double v3 = x + s0 - 2.0 * v2z;
                s1 += c[0] * v3 - c[1] * v1z;
                s2 += c[2] * v3 + c[3] * v1z;
                s0 = x;
                const double low = s2, band = s1;
                const double high = x - c[4] * s1 - s2;
                const double notch = x - c[4] * s1;
                o = (low * c[5]) + (band * c[6]) + (high * c[7]) + (notch * c[8]);
            } else {  // s0 = v[0], s1 = v[1], s2 = v[2]
                s0 = x - (c[3] * s1) - (c[4] * s2);
                o = (c[0] * s0) + (c[1] * s1) + (c[2] * s2);
                s2 = s1;
                s1 = s0;
            }
            *op = o;
            op += V;
        }
    }
    st[v] = s0;
    st[V + v] = s1;
    st[2 * V + v] = s2;
}

// The same recurrences with 16-BYTE read and write streams (round 3): a lane still owns one voice, but the input of two samples
// arrives as one 16-byte load per lane (two voices of one row) and leaves as one 16-byte store, the lanes of a pair swapping one
// value each way (pair_rows_swap / store_pair_rows, mxg_common.h).  V even, N even, both blocks 16-byte aligned; same bits.
template <int KIND, int ST>
__global__ void __launch_bounds__(256) filter2_pairs_kernel(size_t V, size_t N, const double *__restrict__ in,
                                                            const double *__restrict__ coef, double *__restrict__ st,
                                                            double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double s0 = st[v], s1 = st[V + v], s2 = st[2 * V + v];
    constexpr int NC = KIND == 0 ? 1 : (KIND == 1 ? 9 : 5);
    double c[NC];
#pragma unroll
    for (int r = 0; r < NC; r++) c[r] = coef[(size_t)r * V + v];
    const size_t odd = threadIdx.x & 1, vp = v & ~(size_t)1;
    const double *ip = in + vp;
    double *op = out + odd * V + vp;  // this lane's 16 bytes of row n + (lane & 1)
    constexpr int U = 8;
    double2v xn[U / 2];
    auto row_of = [&](size_t n) { const size_t r = n + odd; return r < N ? r : N - 1; };  // clamped: no branch, surplus unused
#pragma unroll
    for (int j = 0; j < U / 2; j++) xn[j] = *reinterpret_cast<const double2v *>(ip + row_of(2 * j) * V);
    for (size_t n0 = 0; n0 < N; n0 += U) {
        double xc[U];
#pragma unroll
        for (int j = 0; j < U / 2; j++) {
            pair_rows_swap(xn[j], xc[2 * j], xc[2 * j + 1]);
            xn[j] = *reinterpret_cast<const double2v *>(ip + row_of(n0 + U + 2 * j) * V);
        }
        double o[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            const double x = xc[i];
            if constexpr (KIND == 0) {
                s1 = x - s0 + c[0] * s1;
                s0 = x;
                o[i] = s1;
            } else if constexpr (KIND == 1) {
                const double v1z = s1;
                const double v2z = s2;
                const double v3 = x + s0 - 2.0 * v2z;
                s1 += c[0] * v3 - c[1] * v1z;
                s2 += c[2] * v3 + c[3] * v1z;
                s0 = x;
                const double low = s2, band = s1;
                const double high = x - c[4] * s1 - s2;
                const double notch = x - c[4] * s1;
                o[i] = (low * c[5]) + (band * c[6]) + (high * c[7]) + (notch * c[8]);
            } else {
                s0 = x - (c[3] * s1) - (c[4] * s2);
                o[i] = (c[0] * s0) + (c[1] * s1) + (c[2] * s2);
                s2 = s1;
                s1 = s0;
            }
            if (n0 + i + 1 == N) {  // the state after the LAST sample of the block (a ragged last chunk computes past it)
                st[v] = s0;
                st[V + v] = s1;
                st[2 * V + v] = s2;
            }
        }
#pragma unroll
        for (int j = 0; j < U / 2; j++) {
            if (n0 + 2 * j < N) store_pair_rows<ST>(op, o[2 * j], o[2 * j + 1]);  // (N even: a pair is inside or outside as a whole; wave-uniform)
            op += 2 * V;
        }
    }
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

int mxg_filter2_render(int kind, size_t V, size_t N, const double *d_in, const double *d_coef, double *d_st,
                       double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(kind >= 0 && kind <= 2, "kind must be 0 (maxiDCBlocker), 1 (maxiSVF) or 2 (maxiBiquad)");
    MXG_REQUIRE(d_in && d_coef && d_st && d_out, "null device pointer");
    if (V == 0 || N == 0) return MXG_OK;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;
    const dim3 grid((unsigned)((V + block - 1) / block));
    hipStream_t st = resolve_stream(stream);
    if (scan_applies(V, N)) return scan_filter_launch(kind, V, N, d_in, d_coef, d_st, d_out, st);  // tolerance mode (scan.hip)
    // 16-byte pair-row streams (knob rw_store: 0 automatic, 1 off, 2 / 3 / 4 plain / write-through / non-temporal stores)
    int rw = tune_get("rw_store");
    const bool pairs_ok = !(V & 1) && !(N & 1) && !(((uintptr_t)d_in) & 15) && !(((uintptr_t)d_out) & 15);
    if (rw == 0) rw = (V * N * sizeof(double) >= ((size_t)64 << 20)) ? 3 : 1;
    if (rw >= 2 && pairs_ok) {
        KernelTimer kt("filter2_kernel", st);
#define MXG_F2P(K)                                                                                                   \
    if (rw == 2) hipLaunchKernelGGL((filter2_pairs_kernel<K, 0>), grid, dim3(block), 0, st, V, N, d_in, d_coef, d_st, d_out); \
    else if (rw == 3) hipLaunchKernelGGL((filter2_pairs_kernel<K, 2>), grid, dim3(block), 0, st, V, N, d_in, d_coef, d_st, d_out); \
    else hipLaunchKernelGGL((filter2_pairs_kernel<K, 1>), grid, dim3(block), 0, st, V, N, d_in, d_coef, d_st, d_out);
        if (kind == 0) { MXG_F2P(0) } else if (kind == 1) { MXG_F2P(1) } else { MXG_F2P(2) }
#undef MXG_F2P
        return check_hip(hipGetLastError(), "filter2_pairs_kernel launch");
    }
    KernelTimer kt("filter2_kernel", st);
    switch (kind) {
        case 0: hipLaunchKernelGGL((filter2_kernel<0>), grid, dim3(block), 0, st, V, N, d_in, d_coef, d_st, d_out); break;
        case 1: hipLaunchKernelGGL((filter2_kernel<1>), grid, dim3(block), 0, st, V, N, d_in, d_coef, d_st, d_out); break;
        default: hipLaunchKernelGGL((filter2_kernel<2>), grid, dim3(block), 0, st, V, N, d_in, d_coef, d_st, d_out); break;
    }
    return check_hip(hipGetLastError(), "filter2_kernel launch");
}

// maxiSVF::setParams (H:1320-1332) on the host libm: rows g1, g2, g3, g4, k of h_coef [5][V]
int mxg_svf_coeffs_host(size_t V, const double *h_cutoff, const double *h_res, double *h_coef) {
    MXG_REQUIRE(h_cutoff && h_res && h_coef, "null pointer");
    const size_t sr = settings().sampleRate;
    for (size_t v = 0; v < V; v++) {
        const double freq = h_cutoff[v], res = h_res[v];
        const double g = tan(MXG_PI * freq / sr);
        const double damping = res == 0 ? 0 : 1.0 / res;
        const double k = damping;
        const double ginv = g / (1.0 + g * (g + k));
        h_coef[v] = ginv;
        h_coef[V + v] = 2.0 * (g + k) * ginv;
        h_coef[2 * V + v] = g * ginv;
        h_coef[3 * V + v] = 2.0 * ginv;
        h_coef[4 * V + v] = k;
    }
    return MXG_OK;
}

// maxiBiquad::set (H:1376-1478) on the host libm: rows a0, a1, a2, b1, b2 of h_coef [5][V].
// h_type[v] in 0..6 = LOWPASS, HIGHPASS, BANDPASS, NOTCH, PEAK, LOWSHELF, HIGHSHELF (H:1349-1358).
int mxg_biquad_coeffs_host(size_t V, const int32_t *h_type, const double *h_cutoff, const double *h_Q,
                           const double *h_peakGain, double *h_coef) {
    MXG_REQUIRE(h_type && h_cutoff && h_Q && h_peakGain && h_coef, "null pointer");
    const size_t sr = settings().sampleRate;
    const double SQRT2 = sqrt(2.0);
    for (size_t v = 0; v < V; v++) {
        const double cutoff = h_cutoff[v], Q = h_Q[v], peakGain = h_peakGain[v];
        double a0 = 0, a1 = 0, a2 = 0, b1 = 0, b2 = 0, norm = 0;
        const double G = pow(10.0, fabs(peakGain) / 20.0);  // the reference's `V`
        const double K = tan(MXG_PI * cutoff / sr);
        switch (h_type[v]) {
            case 0:
                norm = 1.0 / (1.0 + K / Q + K * K);
                a0 = K * K * norm; a1 = 2.0 * a0; a2 = a0;
                b1 = 2.0 * (K * K - 1.0) * norm; b2 = (1.0 - K / Q + K * K) * norm;
                break;
            case 1:
                norm = 1. / (1. + K / Q + K * K);
                a0 = 1 * norm; a1 = -2 * a0; a2 = a0;
                b1 = 2 * (K * K - 1) * norm; b2 = (1 - K / Q + K * K) * norm;
                break;
            case 2:
                norm = 1. / (1. + K / Q + K * K);
                a0 = K / Q * norm; a1 = 0.; a2 = -a0;
                b1 = 2. * (K * K - 1.) * norm; b2 = (1. - K / Q + K * K) * norm;
                break;
            case 3:
                norm = 1. / (1. + K / Q + K * K);
                a0 = (1. + K * K) * norm; a1 = 2. * (K * K - 1.) * norm; a2 = a0;
                b1 = a1; b2 = (1. - K / Q + K * K) * norm;
                break;
            case 4:
                if (peakGain >= 0.0) {
                    norm = 1. / (1. + 1. / Q * K + K * K);
                    a0 = (1. + G / Q * K + K * K) * norm; a1 = 2. * (K * K - 1.) * norm;
                    a2 = (1. - G / Q * K + K * K) * norm; b1 = a1; b2 = (1. - 1. / Q * K + K * K) * norm;
                } else {
                    norm = 1. / (1. + G / Q * K + K * K);
                    a0 = (1. + 1 / Q * K + K * K) * norm; a1 = 2. * (K * K - 1) * norm;
                    a2 = (1. - 1. / Q * K + K * K) * norm; b1 = a1; b2 = (1. - G / Q * K + K * K) * norm;
                }
                break;
            case 5:
                if (peakGain >= 0.) {
                    norm = 1. / (1. + SQRT2 * K + K * K);
                    a0 = (1. + sqrt(2. * G) * K + G * K * K) * norm; a1 = 2. * (G * K * K - 1.) * norm;
                    a2 = (1. - sqrt(2. * G) * K + G * K * K) * norm; b1 = 2. * (K * K - 1.) * norm;
                    b2 = (1. - SQRT2 * K + K * K) * norm;
                } else {
                    norm = 1. / (1. + sqrt(2. * G) * K + G * K * K);
                    a0 = (1. + SQRT2 * K + K * K) * norm; a1 = 2. * (K * K - 1.) * norm;
                    a2 = (1. - SQRT2 * K + K * K) * norm; b1 = 2. * (G * K * K - 1.) * norm;
                    b2 = (1. - sqrt(2. * G) * K + G * K * K) * norm;
                }
                break;
            case 6:
                if (peakGain >= 0.) {
                    norm = 1. / (1. + SQRT2 * K + K * K);
                    a0 = (G + sqrt(2. * G) * K + K * K) * norm; a1 = 2. * (K * K - G) * norm;
                    a2 = (G - sqrt(2. * G) * K + K * K) * norm; b1 = 2. * (K * K - 1) * norm;
                    b2 = (1. - SQRT2 * K + K * K) * norm;
                } else {
                    norm = 1. / (G + sqrt(2. * G) * K + K * K);
                    a0 = (1. + SQRT2 * K + K * K) * norm; a1 = 2. * (K * K - 1.) * norm;
                    a2 = (1. - SQRT2 * K + K * K) * norm; b1 = 2. * (K * K - G) * norm;
                    b2 = (G - sqrt(2. * G) * K + K * K) * norm;
                }
                break;
            default: return fail(MXG_ERR_INVALID, "mxg_biquad_coeffs_host: unknown filter type %d", h_type[v]);
        }
        h_coef[v] = a0; h_coef[V + v] = a1; h_coef[2 * V + v] = a2; h_coef[3 * V + v] = b1; h_coef[4 * V + v] = b2;
    }
    return MXG_OK;
}

}  // extern "C"
