// scan.hip -- SMALL banks of LINEAR recursive filters, parallel in TIME (tolerance mode; knob "time_parallel").
//
// Path: maxiBiquad::play H:1360-1367, maxiSVF::play H:1303-1317, maxiDCBlocker::play H:1261-1266 and maxiFilter::lores / hires
// C:455-484 with block-constant coefficients (hoisted on the host, as in filter2.hip / voice.hip).  The bank kernels give a
// voice to a LANE and walk the block sample by sample: a 6-voice patch (15.polysynth) occupies 6 lanes of one wavefront for
// 512 dependent steps -- 23-28 us of pure latency, however small the bank.  But these filters are linear, time-invariant maps
//      s[n+1] = A s[n] + B x[n],      y[n] = C s[n] + D x[n]            (s: 2-3 doubles per filter)
// so a block can be cut ALONG TIME: a wavefront takes ONE voice, lane k owns the L = N/64 samples [kL, (k+1)L):
//   1. every lane runs its segment from the ZERO state: e_k = the state its inputs alone leave (lane 0 starts from the
//      carried-in state instead, so e_0 is the true state after segment 0);
//   2. M = A^L, the L-step homogeneous transition, is obtained without naming A: lanes 0..2 run the recurrence on the unit
//      states with zero input (the step function is used as a black box), the columns are handed round with readlane;
//   3. the true segment end states obey end_k = M end_{k-1} + e_k: a Kogge-Stone scan over the 64 lanes with wavefront shuffles,
//      q_k += M^(2^j) * shfl_up(q, 2^j), j = 0..5, M^(2^j) by repeated squaring (wave-uniform);
//   4. lane k restarts from end_{k-1} and renders its L samples with the reference's own step, now with the right state.
// 2 L + L recurrence steps and a 6-step scan instead of N dependent steps: a 6-voice x 512-sample block takes a few
// microseconds.  The arithmetic is REORDERED (a state reaches a lane through matrix products instead of through the samples
// before it), so this is a TOLERANCE mode: |error| <= 1e-10 x the block's peak (measured <= 5e-12), stated and tested in tests/test_gpu_scan.py;
// the default (knob 0) stays the bit-exact lane-per-voice kernels.  Used for V <= 4096, N a multiple of 64 up to 2048.
#include "mxg_common.h"

namespace mxg {
namespace {

struct S3 {
    double a, b, c;
};

// the one-sample steps, verbatim from filter2.hip / voice.hip (state in s, returns the output)
template <int KIND>
struct Step {
    double c[9];
    __device__ __forceinline__ double operator()(S3 &s, double x) const {
        if constexpr (KIND == 0) {  // maxiDCBlocker: a = xm1, b = ym1
            s.b = x - s.a + c[0] * s.b;
            s.a = x;
            return s.b;
        } else if constexpr (KIND == 1) {  // maxiSVF: a = v0z, b = v1, c = v2
            const double v1z = s.b, v2z = s.c;
            const double v3 = x + s.a - 2.0 * v2z;
            s.b += c[0] * v3 - c[1] * v1z;
            s.c += c[2] * v3 + c[3] * v1z;
            s.a = x;
            const double low = s.c, band = s.b;
            const double high = x - c[4] * s.b - s.c;
            const double notch = x - c[4] * s.b;
            return (low * c[5]) + (band * c[6]) + (high * c[7]) + (notch * c[8]);
        } else if constexpr (KIND == 2) {  // maxiBiquad, direct form II: a = v[0], b = v[1], c = v[2]
            s.a = x - (c[3] * s.b) - (c[4] * s.c);
            const double o = (c[0] * s.a) + (c[1] * s.b) + (c[2] * s.c);
            s.c = s.b;
            s.b = s.a;
            return o;
        } else {  // maxiFilter::lores (3) / hires (4): a = x, b = y; c[0] = c, c[1] = r
            s.a = s.a + (x - s.b) * c[0];
            s.b = s.b + s.a;
            s.a = s.a * c[1];
            return KIND == 3 ? s.b : x - s.b;
        }
    }
};

struct M3 {
    double m[3][3];  // m[i][j]: coefficient of state j in new state i
};
__device__ __forceinline__ S3 apply(const M3 &M, const S3 &v) {
    S3 r;
    r.a = M.m[0][0] * v.a + M.m[0][1] * v.b + M.m[0][2] * v.c;
    r.b = M.m[1][0] * v.a + M.m[1][1] * v.b + M.m[1][2] * v.c;
    r.c = M.m[2][0] * v.a + M.m[2][1] * v.b + M.m[2][2] * v.c;
    return r;
}
__device__ __forceinline__ M3 square(const M3 &A) {
    M3 R;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R.m[i][j] = A.m[i][0] * A.m[0][j] + A.m[i][1] * A.m[1][j] + A.m[i][2] * A.m[2][j];
    return R;
}
__device__ __forceinline__ double bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

constexpr int kScanMaxL = 32;

// one wavefront per voice; NC coefficient rows [NC][V]; state rows st[r * V + v] for r in rowmap
template <int KIND, int L>
__global__ __launch_bounds__(64) void scan_filter_kernel(size_t V, size_t N, const double *__restrict__ in,
                                                          const double *__restrict__ coef, double *__restrict__ st,
                                                          double *__restrict__ out) {
    const size_t v = blockIdx.x;
    const int lane = threadIdx.x;
    constexpr int NC = KIND == 0 ? 1 : (KIND == 1 ? 9 : (KIND == 2 ? 5 : 2));
    constexpr int NS = KIND >= 3 ? 2 : 3;  // state rows used
    Step<KIND> step;
#pragma unroll
    for (int r = 0; r < NC; r++) step.c[r] = coef[(size_t)r * V + v];
    const S3 s_in = {st[v], st[V + v], NS == 3 ? st[2 * V + v] : 0.0};
    // this lane's inputs
    double x[L];
    const double *ip = in + ((size_t)lane * L) * V + v;
#pragma unroll
    for (int i = 0; i < L; i++) x[i] = ip[(size_t)i * V];
    // 1. zero-state response of the segment (lane 0: from the carried-in state)
    S3 q = lane == 0 ? s_in : S3{0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < L; i++) (void)step(q, x[i]);
    // 2. M = the L-step transition with zero input: lane j (< 3) carries unit state j
    S3 u = {lane == 0 ? 1.0 : 0.0, lane == 1 ? 1.0 : 0.0, lane == 2 ? 1.0 : 0.0};
#pragma unroll
    for (int i = 0; i < L; i++) (void)step(u, 0.0);
    M3 M;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        M.m[0][j] = bcast(u.a, j);
        M.m[1][j] = bcast(u.b, j);
        M.m[2][j] = bcast(u.c, j);
    }
    // 3. end_k = M end_{k-1} + e_k: Kogge-Stone over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        S3 p;
        p.a = __shfl_up(q.a, d);
        p.b = __shfl_up(q.b, d);
        p.c = __shfl_up(q.c, d);
        if (lane >= d) {
            const S3 t = apply(M, p);
            q.a += t.a;
            q.b += t.b;
            q.c += t.c;
        }
        if (d < 32) M = square(M);
    }
    // 4. the true start state of this segment = the end state of the one before
    S3 s;
    s.a = __shfl_up(q.a, 1);
    s.b = __shfl_up(q.b, 1);
    s.c = __shfl_up(q.c, 1);
    if (lane == 0) s = s_in;
    double *op = out + ((size_t)lane * L) * V + v;
#pragma unroll
    for (int i = 0; i < L; i++) op[(size_t)i * V] = step(s, x[i]);
    if (lane == 63) {  // (rendered from the scanned start state: the state a following block continues from)
        st[v] = s.a;
        st[V + v] = s.b;
        if (NS == 3) st[2 * V + v] = s.c;
    }
    (void)N;
}

template <int KIND>
int launch_scan(size_t V, size_t N, const double *in, const double *coef, double *st, double *out, hipStream_t s) {
    const dim3 grid((unsigned)V), blk(64);
    switch (N / 64) {
        case 1: hipLaunchKernelGGL((scan_filter_kernel<KIND, 1>), grid, blk, 0, s, V, N, in, coef, st, out); break;
        case 2: hipLaunchKernelGGL((scan_filter_kernel<KIND, 2>), grid, blk, 0, s, V, N, in, coef, st, out); break;
        case 4: hipLaunchKernelGGL((scan_filter_kernel<KIND, 4>), grid, blk, 0, s, V, N, in, coef, st, out); break;
        case 8: hipLaunchKernelGGL((scan_filter_kernel<KIND, 8>), grid, blk, 0, s, V, N, in, coef, st, out); break;
        case 16: hipLaunchKernelGGL((scan_filter_kernel<KIND, 16>), grid, blk, 0, s, V, N, in, coef, st, out); break;
        case 32: hipLaunchKernelGGL((scan_filter_kernel<KIND, 32>), grid, blk, 0, s, V, N, in, coef, st, out); break;
        default: return 1;
    }
    return 0;
}

}  // namespace

// Whether a (V, N) launch takes the time-parallel kernel: the knob, a small bank, a block of 64 * {1, 2, 4, 8, 16, 32} samples.
bool scan_applies(size_t V, size_t N) {
    if (!tune_get("time_parallel")) return false;
    if (V == 0 || V > 4096 || N % 64) return false;
    const size_t L = N / 64;
    return L >= 1 && L <= (size_t)kScanMaxL && (L & (L - 1)) == 0;
}

// kind: 0 maxiDCBlocker, 1 maxiSVF, 2 maxiBiquad (coefficient / state rows as mxg_filter2_render), 3 lores, 4 hires (coef rows c, r;
// state rows x, y of mxg_filter_render's [5][V])
int scan_filter_launch(int kind, size_t V, size_t N, const double *in, const double *coef, double *st, double *out, hipStream_t s) {
    KernelTimer kt("scan_filter_kernel", s);
    int rc = 1;
    switch (kind) {
        case 0: rc = launch_scan<0>(V, N, in, coef, st, out, s); break;
        case 1: rc = launch_scan<1>(V, N, in, coef, st, out, s); break;
        case 2: rc = launch_scan<2>(V, N, in, coef, st, out, s); break;
        case 3: rc = launch_scan<3>(V, N, in, coef, st, out, s); break;
        case 4: rc = launch_scan<4>(V, N, in, coef, st, out, s); break;
    }
    if (rc) return fail(MXG_ERR_INVALID, "scan_filter_launch: unsupported shape");
    return check_hip(hipGetLastError(), "scan_filter_kernel launch");
}

}  // namespace mxg
