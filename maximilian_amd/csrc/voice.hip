// voice.hip -- maxiFilter / maxiEnv voice banks and the fused subtractive voice on gfx950.
//
// Path (reference src/maximilian.cpp, cited as C:line):
//   maxiFilter::lores C:455-468, hires C:471-484, bandpass C:487-500, lopass C:442-446,
//   hipass C:449-453;  maxiEnv::adsr C:1415-1466 (== 7-arg overload C:1362-1413),
//   ar C:1319-1358;  maxiOsc::saw C:333-340.
// One lane owns one voice; filter memories, envelope amplitude and the five phase flags
// live in VGPRs for the whole block; signals are [N][V] sample-major so a wavefront's
// loads/stores are 512 contiguous bytes of one row.
//
// Numerics.  The recurrences use only + - * and compares => bit-exact.  The lores/hires
// coefficient pair (c, r) needs cos/pow/sqrt (C:459-461).  When cutoff/resonance are
// block-constant they are evaluated ON THE HOST with the host libm (mxg_filter_coeffs_host,
// the same expressions) and uploaded, which keeps the filter bit-exact.  When they are
// modulated per sample (14.monosynth/main.cpp:53) they are evaluated on the device
// (lores_coeffs_dev) under the tolerance stated in DESIGN.md -- a recursive filter amplifies
// a 1-ULP coefficient difference, so no ULP claim is made for that mode.
#include <math.h>

#include <type_traits>

#include "maxi_tables.h"
#include "mxg_common.h"
#include "mxg_env.h"
#include "mxg_lanefold.h"
#include "mxg_pace.h"
#include "mxg_sincos.h"

namespace mxg {

namespace {

// ---- maxiFilter -------------------------------------------------------------------------
struct Flt {
    double x, y, o0, o1, o2;
};

// C:456-461 on the device (modulated mode only).
__device__ __forceinline__ void lores_coeffs_dev(double cutoff, double resonance, double sr,
                                                 double &c, double &r) {
    if (cutoff < 10) cutoff = 10;
    if (cutoff > sr) cutoff = sr;
    if (resonance < 1.) resonance = 1.;
    double z = cos_small(MXG_TWOPI * cutoff / sr);  // argument in [0, 2*pi]: the short-range kernel of mxg_sincos.h
    c = 2 - 2 * z;
    double zm1 = z - 1.0;
    // pow(z-1, 3.0): libm's pow is within 1 ULP of the exact cube; (zm1*zm1)*zm1 is within
    // 1 ULP of it as well.  Both feed the tolerance mode only.
    double cube = (zm1 * zm1) * zm1;
    r = (sqrt(2.0) * sqrt(-cube) + resonance * (z - 1)) / (resonance * (z - 1));
}

// The same pair through ONE sine (round 6, the fused voice's modulated mode: 65 -> ~35 vector instructions per sample).  With
// s = sin(theta / 2), theta = TWOPI*cutoff/sr in [0, 2 pi]:  z = cos(theta) = 1 - 2 s^2, so  c = 2 - 2 z = 4 s^2  and, the square roots of
// C:461 taken symbolically,  sqrt(2) * sqrt(-(z - 1)^3) = 4 s^3  and  r = (4 s^3 - 2 res s^2) / (-2 res s^2) = 1 - (2 / res) s:
// no cos, no cube, two square roots and a division fewer -- and no cancellation in z - 1, so these are the TRUE coefficients where the
// reference's own carry its rounding of z (relative 1e-16 / (1 - z): up to 1e-10 at a 10 Hz cutoff).  Tolerance mode either way (DESIGN.md:
// 1e-11 of the voice's peak; the worst case of the difference, cutoff = amplitude * 10000 at the bottom of a release, is 1e-13).
// kr = 2 / max(res, 1) (C:458) and pisr = pi / sr are hoisted by the caller.  Where z rounds to 1 (cutoff clamped to sr: theta = 2 pi)
// the reference has c = 0 and r = 0 / 0: reproduced.
__device__ __forceinline__ void lores_coeffs_sin(double cutoff, const double kr, const double pisr, const double sr, double &c, double &r) {
    using namespace sincos_detail;
    if (cutoff < 10) cutoff = 10;
    if (cutoff > sr) cutoff = sr;
    const double phi = cutoff * pisr;  // theta / 2 in [0, pi]
    constexpr double kPiHi = 2.0 * kPio2Hi, kPiLo = 2.0 * kPio2Lo;
    const double y = phi > 0.5 * (kPiHi + kPiLo) ? (kPiHi - phi) + kPiLo : phi;  // sin(pi - phi) = sin(phi): y in [0, pi/2]
    // sin on [0, pi/2] as ONE odd polynomial (the Taylor coefficients through y^21: truncation 1.3e-18 at pi/2; ten fused multiply-adds
    // with the coefficient as the instruction's scalar operand) instead of fdlibm's two kernels on a folded argument and a select
    const double z = y * y;
    double pq = fma_kk(z, 1.9572941063391263e-20, -8.2206352466243295e-18);
    pq = fma_k(z, pq, 2.8114572543455206e-15);
    pq = fma_k(z, pq, -7.6471637318198164e-13);
    pq = fma_k(z, pq, 1.6059043836821613e-10);
    pq = fma_k(z, pq, -2.5052108385441720e-08);
    pq = fma_k(z, pq, 2.7557319223985893e-06);
    pq = fma_k(z, pq, -1.9841269841269841e-04);
    pq = fma_k(z, pq, 8.3333333333333332e-03);
    pq = fma_k(z, pq, -1.6666666666666666e-01);
    const double sv = y + (y * z) * pq;
    const double m = 2.0 * (sv * sv);  // 1 - z
    c = 2.0 * m;
    r = 1.0 - kr * sv;
    if (!(m > 0x1p-54)) {  // z = fl(1 - m) = 1: C:459-461 give c = 2 - 2 = 0 and r = 0 / 0 (a NaN cutoff lands here as well)
        c = m != m ? m : 0.0;
        r = __builtin_nan("");
    }
}

// The same pair where theta / 2 = cutoff * pi / sr <= pi / 4 for certain (cutoff <= sr / 4, tested by the caller for a whole chunk and wavefront:
// 14.monosynth's envelope x 10 000 Hz at 44.1 kHz never leaves that range): no fold, no upper clamp, the Taylor polynomial through y^15
// (truncation y^16 / 17! <= 5.9e-17 of the value at pi / 4) and no z = 1 case (the caller checks 10 * pi / sr > 2^-26, so m > 2^-51) --
// 14 vector instructions instead of ~37.  K2f on a lone wavefront is bound by its instruction COUNT (profiles/r06_k2f_split.md), so this
// is mode B's lever.  nkr = -2 / max(res, 1).  Tolerance mode: within an ulp of lores_coeffs_sin's s, c = 4 s^2 the same expression.
__device__ __forceinline__ void lores_coeffs_sin_small(double cutoff, const double nkr, const double pisr, double &c, double &r) {
    using namespace sincos_detail;
    asm("v_max_f64 %0, %1, %2" : "=v"(cutoff) : "v"(cutoff), "s"(10.0));  // C:456 (not a NaN here: the caller's test; fmax() would canonicalize first)
    const double y = cutoff * pisr;
    const double z = y * y;
    double pq = fma_kk(z, -7.6471637318198164e-13, 1.6059043836821613e-10);
    pq = fma_k(z, pq, -2.5052108385441720e-08);
    pq = fma_k(z, pq, 2.7557319223985893e-06);
    pq = fma_k(z, pq, -1.9841269841269841e-04);
    pq = fma_k(z, pq, 8.3333333333333332e-03);
    pq = fma_k(z, pq, -1.6666666666666666e-01);
    const double sv = fma(y * z, pq, y);
    c = 4.0 * (sv * sv);
    r = fma(nkr, sv, 1.0);
}

__device__ __forceinline__ double flt_lores(Flt &f, double input, double c, double r) {  // C:463-467
    f.x = f.x + (input - f.y) * c;
    f.y = f.y + f.x;
    f.x = f.x * r;
    return f.y;
}

template <int KIND, bool MOD>
__global__ void __launch_bounds__(256) filter_kernel(size_t V, size_t N, const double *__restrict__ in,
                              const double *__restrict__ cutoff, int cps,
                              const double *__restrict__ res, int rps,
                              const double *__restrict__ coef, double *__restrict__ st,
                              double *__restrict__ out, double sr) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    Flt f = {st[v], st[V + v], st[2 * V + v], st[3 * V + v], st[4 * V + v]};
    double c = 0, r = 0, cut0 = 0, res0 = 0;
    if constexpr (KIND <= MXG_FLT_HIRES && !MOD) {
        c = coef[v];
        r = coef[V + v];
    } else {
        cut0 = cutoff[v];  // row 0 of [N][V] or the [V] constant
        res0 = res ? res[v] : 0.0;
    }
    // bandpass: constant parameters -> host-libm coefficients; modulated -> device (C:489-495).
    double b0 = 0, b1 = 0, b2 = 0;
    auto bp_coeffs = [&](double cu, double rs) {
        if (cu > (sr * 0.5)) cu = (sr * 0.5);
        if (rs >= 1.) rs = 0.999999;
        double z = cos_small(MXG_TWOPI * cu / sr);
        b0 = (1 - rs) * (sqrt(rs * (rs - 4.0 * (z * z) + 2.0) + 1));
        b1 = 2 * z * rs;
        b2 = (rs * -1) * (rs * -1);
    };
    if constexpr (KIND == MXG_FLT_BANDPASS && !MOD) {
        b0 = coef[v];
        b1 = coef[V + v];
        b2 = coef[2 * V + v];
    }

    const double *ip = in + v;
    double *op = out + v;
    const double *cp = cutoff + v;
    const double *rp = res ? res + v : nullptr;
    // Software pipeline: the inputs of chunk k+1 are requested BEFORE chunk k's outputs are stored,
    // so the wait for them is a counted vmcnt(U) and never drains the store stream (loads and
    // stores retire in order on one counter; a load issued after a store would wait for it).
    constexpr int U = (KIND >= MXG_FLT_LOPASS) ? 16 : 8;  // measured: 1-pole kinds want more loads in flight
    double xn[U], cn[U], rn[U];
#pragma unroll
    for (int i = 0; i < U; i++) {
        // clamped index instead of a guarded load: no branch, the surplus values are never used
        const size_t m = (size_t)i < N ? (size_t)i : N - 1;
        xn[i] = ip[m * V];
        cn[i] = (MOD && cps) ? cp[m * V] : cut0;
        rn[i] = (MOD && rps) ? rp[m * V] : res0;
    }
    for (size_t n0 = 0; n0 < N; n0 += U) {
      double xc[U], cc[U], rc[U];
#pragma unroll
      for (int i = 0; i < U; i++) {
        xc[i] = xn[i]; cc[i] = cn[i]; rc[i] = rn[i];
        const size_t m = (n0 + U + i < N) ? n0 + U + i : N - 1;
        xn[i] = ip[m * V];
        if constexpr (MOD) {
            cn[i] = cps ? cp[m * V] : cut0;
            rn[i] = rps ? rp[m * V] : res0;
        }
      }
#pragma unroll
      for (int i = 0; i < U; i++) {
        if (n0 + i >= N) break;
        double x = xc[i];
        double cu = cc[i], rs = rc[i];
        double o;
        if constexpr (KIND == MXG_FLT_LORES || KIND == MXG_FLT_HIRES) {
            if constexpr (MOD) lores_coeffs_dev(cu, rs, sr, c, r);
            double y = flt_lores(f, x, c, r);
            o = (KIND == MXG_FLT_LORES) ? y : x - y;  // C:466 / C:482
        } else if constexpr (KIND == MXG_FLT_BANDPASS) {  // C:487-500
            if constexpr (MOD) bp_coeffs(cu, rs);
            o = b0 * x + b1 * f.o1 + b2 * f.o2;
            f.o2 = f.o1;
            f.o1 = o;
        } else if constexpr (KIND == MXG_FLT_LOPASS) {  // C:442-446
            o = f.o0 + cu * (x - f.o0);
            f.o0 = o;
        } else {  // hipass C:449-453
            o = x - (f.o0 + cu * (x - f.o0));
            f.o0 = o;
        }
        *op = o;
        op += V;
      }
    }
    st[v] = f.x;
    st[V + v] = f.y;
    st[2 * V + v] = f.o0;
    st[3 * V + v] = f.o1;
    st[4 * V + v] = f.o2;
}

// The block-constant forms with 16-BYTE read and write streams (round 4; the scheme of filter2_pairs_kernel, round 3): a lane still
// owns one voice, but the input of two samples arrives as one 16-byte load per lane (two voices of one row) and leaves as one 16-byte
// store, the lanes of a pair swapping one value each way (pair_rows_swap / store_pair_rows, mxg_common.h).  V even, N even, both
// blocks 16-byte aligned; the same recurrences in the same order: the same bits.
#ifndef MXG_FLT_SPREAD
#define MXG_FLT_SPREAD 0  // A/B (tools/build_ab.sh): 1 = every pair of samples is stored as soon as it is computed
#endif
template <int KIND, int ST, int U>
__global__ void __launch_bounds__(256) filter_pairs_kernel(size_t V, size_t N, const double *__restrict__ in,
                                                           const double *__restrict__ cutoff, const double *__restrict__ coef,
                                                           double *__restrict__ st, double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    Flt f = {st[v], st[V + v], st[2 * V + v], st[3 * V + v], st[4 * V + v]};
    double c = 0, r = 0, b0 = 0, b1 = 0, b2 = 0, cu = 0;
    if constexpr (KIND <= MXG_FLT_HIRES) {
        c = coef[v];
        r = coef[V + v];
    } else if constexpr (KIND == MXG_FLT_BANDPASS) {
        b0 = coef[v];
        b1 = coef[V + v];
        b2 = coef[2 * V + v];
    } else {
        cu = cutoff[v];
    }
    const size_t odd = threadIdx.x & 1, vp = v & ~(size_t)1;
    const double *ip = in + vp;
    const unsigned op_off = (unsigned)((odd * V + vp) * sizeof(double));  // this lane's 16 bytes of row n + (lane & 1), from row n's start
    // (U samples per chunk = U / 2 16-byte loads in flight per lane: 8 -> 4 MB over the whole machine at 65 536 voices, where HBM's
    // latency x bandwidth is ~12 MB; knob rw_chunk)
    double2v xn[U / 2];
    auto row_of = [&](size_t n) { const size_t rr = n + odd; return rr < N ? rr : N - 1; };  // clamped: no branch, surplus unused
#pragma unroll
    for (int j = 0; j < U / 2; j++) xn[j] = *reinterpret_cast<const double2v *>(ip + row_of(2 * j) * V);
    // consume the prologue's loads HERE: a single s_waitcnt at the loop header has to serve the entry path too, where these are the only
    // operations in flight -- vmcnt(0) -- and would then drain the stores of every iteration (see sample.hip, delay_kernel)
#pragma unroll
    for (int j = 0; j < U / 2; j++) asm volatile("" : "+v"(xn[j]));
    asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(f.o0), "+v"(f.o1), "+v"(f.o2));  // (the state and the coefficients likewise)
    asm volatile("" : "+v"(c), "+v"(r), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(cu));
    // One chunk.  LAST = false: a whole chunk that is not the block's last -- no test of the sample number anywhere (twelve wave-uniform
    // tests = scalar branches per chunk of eight samples in the first form).  LAST = true: the chunk that holds sample N - 1 (ragged or
    // not): the state is stored after exactly that sample.  Measured (round 4, same box, A/B builds): with the branches and with asm
    // stores that drained the store queue once per chunk 91.7-93.2 us; without either 93.5-94.0 us; without the branches but still
    // draining 104-111 us; a fixed pause per chunk changed nothing -- the kernel sits on the memory system's rate either way.
    auto chunk = [&](const size_t n0, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        double xc[U];
#pragma unroll
        for (int j = 0; j < U / 2; j++) {
            pair_rows_swap(xn[j], xc[2 * j], xc[2 * j + 1]);
            if constexpr (!LAST) xn[j] = *reinterpret_cast<const double2v *>(ip + row_of(n0 + U + 2 * j) * V);
        }
        double o[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            const double x = xc[i];
            if constexpr (KIND == MXG_FLT_LORES || KIND == MXG_FLT_HIRES) {
                const double y = flt_lores(f, x, c, r);
                o[i] = (KIND == MXG_FLT_LORES) ? y : x - y;  // C:466 / C:482
            } else if constexpr (KIND == MXG_FLT_BANDPASS) {  // C:487-500
                o[i] = b0 * x + b1 * f.o1 + b2 * f.o2;
                f.o2 = f.o1;
                f.o1 = o[i];
            } else if constexpr (KIND == MXG_FLT_LOPASS) {  // C:442-446
                o[i] = f.o0 + cu * (x - f.o0);
                f.o0 = o[i];
            } else {  // hipass C:449-453
                o[i] = x - (f.o0 + cu * (x - f.o0));
                f.o0 = o[i];
            }
            if constexpr (LAST) {
                if (n0 + i + 1 == N) {  // the state after the LAST sample of the block (a ragged last chunk computes past it)
                    st[v] = f.x;
                    st[V + v] = f.y;
                    st[2 * V + v] = f.o0;
                    st[3 * V + v] = f.o1;
                    st[4 * V + v] = f.o2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U / 2; j++)  // (N even: a pair is inside or outside as a whole)
            if (!LAST || n0 + 2 * j < N) store_pair_rows_at<ST>(out + (n0 + 2 * j) * V, op_off, o[2 * j], o[2 * j + 1]);
    };
    size_t n0 = 0;
    for (; n0 + U < N; n0 += U) chunk(n0, std::false_type{});
    chunk(n0, std::true_type{});
}

// lores / hires / bandpass with the coefficients given PER SAMPLE (coef [N][3][V]: c, r, - or inputs[0..2]), computed by the caller
// with the host libm (mxg_filter_coeffs_host per sample): the bit-exact form of a modulated cutoff -- what the per-sample engine of
// include/maximilian.h renders when a cutoff follows another object's output (14.monosynth's `ADSRout*10000`).  Small banks, short
// blocks: a plain loop.
template <int KIND>
__global__ void __launch_bounds__(256) filter_coefps_kernel(size_t V, size_t N, const double *__restrict__ in,
                                                             const double *__restrict__ coef, double *__restrict__ st,
                                                             double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    Flt f = {st[v], st[V + v], st[2 * V + v], st[3 * V + v], st[4 * V + v]};
    for (size_t n = 0; n < N; n++) {
        const double x = in[n * V + v];
        const double *cp = coef + (n * 3) * V + v;
        double o;
        if constexpr (KIND == MXG_FLT_BANDPASS) {  // C:487-500
            o = cp[0] * x + cp[V] * f.o1 + cp[2 * V] * f.o2;
            f.o2 = f.o1;
            f.o1 = o;
        } else {
            const double y = flt_lores(f, x, cp[0], cp[V]);
            o = (KIND == MXG_FLT_LORES) ? y : x - y;  // C:466 / C:482
        }
        out[n * V + v] = o;
    }
    st[v] = f.x;
    st[V + v] = f.y;
    st[2 * V + v] = f.o0;
    st[3 * V + v] = f.o1;
    st[4 * V + v] = f.o2;
}

// PX: 0 = 8-byte input / output streams; 1 / 2 / 3 = 16-byte pair rows both ways (pair_rows_swap / store_pair_rows, mxg_common.h) with
// plain / write-through / non-temporal stores -- V even, N even, both blocks 16-byte aligned (round 4; the same ticks in the same order).
template <int MODE, bool HASIN, bool TPV, int PX>
__global__ void __launch_bounds__(256) env_kernel(size_t V, size_t N, const double *__restrict__ in,
                           const int32_t *__restrict__ trig, int tpv,
                           const double *__restrict__ par, const int64_t *__restrict__ holdtime,
                           double *__restrict__ dst, int64_t *__restrict__ ist,
                           double *__restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((gid & ~(size_t)63) >= V) return;  // the whole wavefront is past the bank
    // (pair rows: the surplus lanes of the last wavefront shadow the last PAIR of voices, parity kept: voice_kernel)
    const size_t v = PX ? (gid < V ? gid : V - 2 + (gid & 1)) : live_voice(gid, V);
    constexpr int PST = PX == 2 ? 2 : (PX == 3 ? 1 : 0);
    Env e;
    env_load(e, V, v, par, holdtime, dst, ist);
    const size_t odd = threadIdx.x & 1, vp = v & ~(size_t)1;
    const double *ip = in ? in + (PX ? vp : v) : nullptr;
    double *op = PX ? out + odd * V + vp : out + v;  // pair rows: this lane's 16 bytes of row n + (lane & 1)
    // one chunk of U outputs, the first `cnt` of them inside the block (cnt even with pair rows: N is)
    auto emit = [&](const double (&o)[8], size_t cnt) {
        if constexpr (PX != 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if ((size_t)(2 * j) < cnt) store_pair_rows<PST>(op, o[2 * j], o[2 * j + 1]);
                op += 2 * V;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if ((size_t)i < cnt) *op = o[i];
                op += V;
            }
        }
    };
    // software pipeline (see filter_kernel): next chunk's inputs/triggers are requested before this
    // chunk's outputs are stored
    constexpr int U = 8;  // measured with the steady-state paths: 8 beats 4 on both paths
    double xn[U];
    double2v xr[U / 2];  // pair rows: the raw 16-byte loads of the next chunk
    int tn[U];
    GateGroup<U, int32_t> gcur;  // shared gate: see gate_group_load
    const auto gate_on = [](int32_t t) { return t == 1; };  // the test of C:1363 / C:1425
    auto row_of = [&](size_t n) { const size_t rr = n + odd; return rr < N ? rr : N - 1; };  // clamped: no branch, surplus unused
#pragma unroll
    for (int i = 0; i < U; i++) {
        const size_t m = (size_t)i < N ? (size_t)i : N - 1;
        if constexpr (!HASIN) xn[i] = 1.0;
        else if constexpr (PX == 0) xn[i] = ip[m * V];
        if constexpr (TPV) tn[i] = trig[m * V + v];
    }
    if constexpr (HASIN && PX != 0) {
#pragma unroll
        for (int j = 0; j < U / 2; j++) xr[j] = *reinterpret_cast<const double2v *>(ip + row_of(2 * j) * V);
    }
    if constexpr (!TPV) {
        gate_group_load(gcur, trig, N, 0, gate_on);
        asm volatile("" : "+v"(gcur.cls));
    }
    for (size_t n0 = 0; n0 < N; n0 += U) {
        double xc[U];
        int tc[U];
        if constexpr (HASIN && PX != 0) {
#pragma unroll
            for (int j = 0; j < U / 2; j++) {
                pair_rows_swap(xr[j], xc[2 * j], xc[2 * j + 1]);
                xr[j] = *reinterpret_cast<const double2v *>(ip + row_of(n0 + U + 2 * j) * V);
            }
        }
#pragma unroll
        for (int i = 0; i < U; i++) {
            if constexpr (!(HASIN && PX != 0)) xc[i] = xn[i];
            if constexpr (TPV) tc[i] = tn[i];
            const size_t m = (n0 + U + i < N) ? n0 + U + i : N - 1;
            if constexpr (HASIN && PX == 0) xn[i] = ip[m * V];
            if constexpr (TPV) tn[i] = trig[m * V + v];
        }
        int fast = 0;
        if constexpr (!TPV) {
            const int cc = (int)((n0 / U) & 63);
            if (cc == 0) {  // launches longer than 64 chunks: one drain of the store stream per 64*U samples
                if (n0) {
                    gate_group_load(gcur, trig, N, n0 / (64 * U), gate_on);
                }
            }
            const int g = lane_value(gcur.cls, cc);
            if constexpr (MODE == 0) {
                if (g > 0) fast = __all(env_in_sustain(e)) ? 1 : 0;
                else if (g < 0) fast = __all(env_in_release(e)) ? 2 : 0;
            }
        }
        if constexpr (MODE == 0) {
            // neither all-sustain nor all-release: the general steady chunk (mxg_env.h) -- every lane inside one stage of
            // its own for the whole chunk, gate constant per lane; computed on a copy, committed if every lane accepts
            if (fast == 0 && n0 + U <= N) {
                bool lane_gate, constant;
                if constexpr (TPV) {
                    lane_gate = tc[0] == 1;
                    constant = true;
#pragma unroll
                    for (int i = 1; i < U; i++) constant = constant && ((tc[i] == 1) == lane_gate);
                } else {
                    const int g = lane_value(gcur.cls, (int)((n0 / U) & 63));
                    lane_gate = g > 0;
                    constant = g != 0;
                }
                if (__all(constant)) {
                    Env s = e;
                    double o[U];
                    const bool ok = env_steady_chunk<U>(s, xc, lane_gate, o);
                    if (__all(ok)) {
                        e = s;
                        emit(o, U);
                        continue;
                    }
                }
            }
        }
        // (fast paths: the gate class of the chunk is constant, which a ragged last chunk's class covers as well -- but the state
        // must stop at sample N: they take whole chunks only)
        double o[U];
        const size_t cnt = N - n0 < (size_t)U ? N - n0 : (size_t)U;
        if (fast == 1 && cnt == (size_t)U) {
#pragma unroll
            for (int i = 0; i < U; i++) o[i] = env_sustain_tick(e, xc[i]);
        } else if (fast == 2 && cnt == (size_t)U) {
#pragma unroll
            for (int i = 0; i < U; i++) o[i] = env_release_tick(e, xc[i]);
        } else {
#pragma unroll
            for (int i = 0; i < U; i++) {
                if (n0 + i >= N) break;
                int t;
                if constexpr (TPV) t = tc[i];
                else t = lane_value(gcur.g[i], (int)((n0 / U) & 63));
                o[i] = (MODE == 0) ? env_adsr(e, xc[i], t) : env_ar(e, xc[i], t);
            }
        }
        emit(o, cnt);
    }
    env_store(e, V, v, dst, ist);
}

// maxiOsc::saw (C:333-340) for the fused voice: output = phase; if (phase >= 1.0) phase -= 2.0; phase += 1/(sr/f) * 2.
// The conditional subtraction is ONE addition of -2.0 or -0.0 (x + -0.0 = x for every x, both zeros and NaNs included; the two
// constants differ in their high word only): compare, select one dword, add -- the same bits in four instructions where compare,
// subtract and a two-dword select take five.  K2f on a lone wavefront is bound by its instruction count (profiles/r06_k2f_split.md).
template <bool ONE_ADD>
__device__ __forceinline__ double saw_tick(double &phase, double &hold, const double inc) {
    const double s = phase;
    hold = s;
    if constexpr (ONE_ADD) {
        const int hi = phase >= 1.0 ? (int)0xC0000000 : (int)0x80000000;
        phase = phase + __hiloint2double(hi, 0);
    } else {
        if (phase >= 1.0) phase -= 2.0;
    }
    phase += inc;
    return s;
}

#ifndef MXG_VOICE_RUNS
#define MXG_VOICE_RUNS 1  // A/B: 0 = the steady paths chunk by chunk
#endif
#ifndef MXG_MODB_SMALL
#define MXG_MODB_SMALL 1  // A/B (tools/build_ab.sh): 0 = every chunk of mode B through the full-range coefficients
#endif
// ---- fused subtractive voice (K2) ----------------------------------------------------------
// MODE 0: out = adsr(lores(saw(f), c, r), trig)             (coefficients hoisted, bit-exact)
// MODE 1: e = adsr(1., trig); out = lores(saw(f), e*cutoff, res) * e   (device coefficients)
// The trigger is read through a scalar load when it is shared by the bank (TPV = false): a
// per-sample VECTOR load would make every sample wait on vmcnt(0), i.e. on all earlier output
// stores as well (loads and stores retire in order on one counter) -- measured 3.7x slower.
// Per-voice triggers (TPV = true) are prefetched one 8-sample chunk ahead for the same reason.
// ST / PX: the store stream as in K1 (osc.hip): ST = 0 plain, 1 nt, 2 sc1; PX = two samples of a lane pair leave as ONE 16-byte
// store per lane (store_pair_rows, mxg_common.h; V even, out 16-byte aligned).  xcd: XCD-contiguous workgroup numbering.
//
// MIX (round 6): the fused maxiMix::stereo mixdown (C:503-509 applied voice after voice, summed in user code: 15.polysynth/main.cpp:54-70),
// K1m's producer / consumer form (osc.hip, osc_mixpc_kernel).  A workgroup is 512 lanes: wavefronts 0-3 are the PRODUCERS -- this
// kernel's instruction stream for 64 voices each, plus one ds_write_b64 per sample into a [16 samples][64 voices] tile -- and wavefronts
// 4-7 the CONSUMERS (mixpc_consume, mxg_lanefold.h: the transposed tile read, 16 + 16 products per lane, a fixed tree), consumer w + 4
// serving producer w on the same SIMD at lower priority.  Output besides the block: the per-workgroup rows partial[workgroup][N][2],
// the layout of mxg_osc_render_mix_rows, so that the grouped mix queue (comm.hip) folds them the same way.  ST = 3: no per-voice
// block at all (out may be null).  The ticks, their order and the per-voice stores are the ones of MIX = false: the same bits.
constexpr int kVoiceMixWin = 512;
// DIET (round 6): the fast paths' instruction diet -- saw_tick's one-add wrap, release chunks taken speculatively, the sustain / release
// test carried from chunk to chunk and whole RUNS of steady chunks in one tight loop: 16.7 + 3.2 -> ~12.5 + 0.5 vector + scalar
// instructions per sample.  A lone wavefront pays ~5 cycles per instruction it issues (profiles/r06_k2f_split.md): banks up to 32 768
// voices 36.6 -> 27 us.  At 65 536 voices -- every SIMD holding exactly one wavefront, the store stream the bound -- a FASTER stream by
// itself is SLOWER (47.7 -> 51 us: it runs into the memory system's back-pressure harder); there the kernel runs on the paced
// schedule (PACE below, mxg_pace.h) and the short stream is what lets the period be short: 42 us.  Same bits (knob voice_diet = 1:
// the round-5 stream, for comparison).
template <int MODE, int ST, bool PX, bool TPV, bool MIX, bool DIET>
__global__ void __launch_bounds__(MIX ? 512 : 256) voice_kernel(size_t V, size_t N, const double *__restrict__ freq,
                             const double *__restrict__ cutoff, const double *__restrict__ res,
                             const double *__restrict__ coef, const int32_t *__restrict__ trig,
                             int tpv, const double *__restrict__ par,
                             const int64_t *__restrict__ holdtime, double *__restrict__ ost,
                             double *__restrict__ fst, double *__restrict__ dst,
                             int64_t *__restrict__ ist, double *__restrict__ out, double sr, int xcd,
                             const double *__restrict__ pan, double *__restrict__ partial, unsigned *__restrict__ pace_ctl,
                             unsigned pace_arg) {
    constexpr int WIN = kVoiceMixWin;
    const size_t wg = (size_t)xcd_block(blockIdx.x, gridDim.x, xcd);
    const size_t gid = MIX ? wg * 256 + (threadIdx.x & 255) : wg * blockDim.x + threadIdx.x;  // (MIX: 256 voices per workgroup of 512 lanes)
    // mixdown form: the pair's ring of tiles, the four consumer rows of a window, the scratch row, the counters
    double *ring = nullptr, *s_part = nullptr, *my_part = nullptr, *s_dump = nullptr;
    int *f_prod = nullptr, *f_cons = nullptr;
    const int lane = threadIdx.x & 63, ts = lane & 15, tq = lane >> 4;
    if constexpr (MIX) {
        __shared__ __attribute__((aligned(16))) double s_all[4 * kPcRing * kTileWave + 4 * WIN * 2 + 256 + 8];
        const int pw = (threadIdx.x >> 6) & 3;  // the pair
        ring = s_all + pw * (kPcRing * kTileWave);
        s_part = s_all + 4 * kPcRing * kTileWave;  // [4 pairs][WIN][2]
        my_part = s_part + pw * (WIN * 2);
        s_dump = s_part + 4 * WIN * 2;  // [256]
        int *flags = reinterpret_cast<int *>(s_dump + 256);  // [4 pairs][2]: prod, cons
        f_prod = flags + 2 * pw;
        f_cons = flags + 2 * pw + 1;
        const bool producer = threadIdx.x < 256;
        if (threadIdx.x < 8) flags[threadIdx.x] = 0;
        if (producer) {
            const bool live = gid < V;
            double x = pan[live ? gid : V - 1];
            if (x > 1) x = 1;  // C:504
            if (x < 0) x = 0;  // C:505
            ring[lane] = live ? sqrt(1.0 - x) : 0.0;  // two[0] = input*sqrt(1.0-x)   C:506
            ring[64 + lane] = live ? sqrt(x) : 0.0;   // two[1] = input*sqrt(x)       C:507
        }
        __syncthreads();  // (the gains, the counters)
        if (!producer) {
            double gl[16], gr[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                gl[j] = ring[16 * tq + j];
                gr[j] = ring[64 + 16 * tq + j];
            }
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" : "+v"(gl[j]), "+v"(gr[j]));
            __syncthreads();  // every consumer has its gains: the ring is free
            mixpc_consume<WIN, 4>(N, ring, f_prod, f_cons, gl, gr, s_part, my_part, s_dump, partial + wg * N * 2);
            return;
        }
        __syncthreads();
        // (no early exit below: a producer wavefront past the bank shadows the bank's last voices -- the same loads, arithmetic and
        // stores of the same values to the same addresses -- with zero gains, so that every barrier is met)
    } else {
        if ((gid & ~(size_t)63) >= V) return;  // the whole wavefront is past the bank
    }
    // (pair rows: the surplus lanes of the last wavefront shadow the last PAIR of voices, parity kept, so that they exchange
    // among themselves and store the values their owners store, to the same addresses)
    const size_t v = PX ? (gid < V ? gid : V - 2 + (gid & 1)) : live_voice(gid, V);
    double phase = ost[v], hold = ost[V + v];
    Flt f = {fst[v], fst[V + v], fst[2 * V + v], fst[3 * V + v], fst[4 * V + v]};
    Env e;
    env_load(e, V, v, par, holdtime, dst, ist);
    const double inc = (1. / (sr / (freq[v]))) * 2.0;  // C:337
    double c = 0, r = 0, cut = 0, rs = 0, kr = 0, nkr = 0;
    const double pisr = MXG_PI / sr;
    // (mode B's small-angle chunks, lores_coeffs_sin_small: every cutoff of the chunk <= sr / 4; a NaN limit switches them off)
    const double small_lim = (MXG_MODB_SMALL && 10.0 * pisr > 0x1p-26) ? 0.25 * sr : __builtin_nan("");
    if constexpr (MODE == 0) {
        c = coef[v];
        r = coef[V + v];
    } else {
        cut = cutoff[v];
        rs = res[v];
        kr = 2.0 / (rs < 1. ? 1. : rs);  // C:458; a NaN resonance stays NaN (rs < 1 is false, 2 / NaN)
        nkr = -kr;
    }
    double *op = out + v;
    double *pp = out + (size_t)(threadIdx.x & 1) * V + (v & ~(size_t)1);  // pair rows: this lane's 16 bytes of row n + (lane & 1)
    // one chunk of U samples to the output: pairs of rows as 16-byte stores, or sample by sample
    // mixdown form: tile k of the pair's ring holds samples 16 k .. 16 k + 15 of the block (windows are whole tiles); ns = rows written
    // so far (wave-uniform).  The consumer must be done with the tile that a new one overwrites -- it normally was a tile ago (`seen`
    // is requested at the start of a tile and looked at one tile later: no wait on the way).
    int tk = 0, ns = 0, seen = 0, seen_next = 0;
    auto tile_rows = [&](const double *o, int cnt) {  // cnt rows from o[0..cnt)
        if constexpr (MIX) {
            if (ns == 0) {
                if (seen < tk - kPcRing + 1)
                    while ((seen = lds_flag_load(f_cons)) < tk - kPcRing + 1) __builtin_amdgcn_s_sleep(1);
                seen_next = lds_flag_load(f_cons);
                asm volatile("" ::: "memory");
            }
            double *tw = ring + (tk % kPcRing) * kTileWave + tq * kTileQuarter + ts + ns * kTileRow;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < cnt) tw[i * kTileRow] = o[i];
            ns += cnt;
            if (ns == kMixChunk) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                lds_flag_store(f_prod, tk + 1);  // (behind the tile writes in the LDS queue)
                seen = seen_next;
                ns = 0;
                tk++;
            }
        }
    };
    auto tile_flush = [&]() {  // a ragged last tile: zero rows behind the block's last sample, then publish
        if constexpr (MIX) {
            if (ns) {
                double *tw = ring + (tk % kPcRing) * kTileWave + tq * kTileQuarter + ts;
                for (int i = ns; i < kMixChunk; i++) tw[i * kTileRow] = 0.0;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                lds_flag_store(f_prod, tk + 1);
                ns = 0;
                tk++;
            }
        }
    };
    auto put1 = [&](double o) {  // one sample, sample by sample (ragged last chunk; 8-byte store streams)
        if constexpr (ST != 3) {
            store1<ST>(op, o);
            op += V;
        }
        tile_rows(&o, 1);
    };
    auto emit = [&](const double (&o)[8]) {
        if constexpr (ST == 3) {
        } else if constexpr (PX) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                store_pair_rows<ST>(pp, o[i], o[i + 1]);
                pp += 2 * V;
            }
            op += 8 * V;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                store1<ST>(op, o[i]);
                op += V;
            }
        }
        tile_rows(o, 8);
    };
    // consume the prologue loads here so no vmcnt(0) is needed inside the loop (see osc.hip K1m)
    asm volatile("" : "+v"(phase), "+v"(hold), "+v"(c), "+v"(r), "+v"(cut), "+v"(rs), "+v"(kr), "+v"(nkr));
    asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(e.amplitude), "+v"(e.output), "+v"(e.attack), "+v"(e.decay));
    asm volatile("" : "+v"(e.sustain), "+v"(e.release), "+v"(e.holdtime), "+v"(e.holdcount));
    constexpr int U = 8;
    // mode B, one whole chunk from its eight envelope values: oscillator, coefficients, filter, product.  The coefficients take the
    // small-angle form when every cutoff of the chunk, on every lane of the wavefront, is <= sr / 4 (a NaN compares false).
    auto modb_chunk = [&](const double (&av)[U]) {
        double cf[U], o[U];
        bool small = true;
#pragma unroll
        for (int i = 0; i < U; i++) {
            cf[i] = av[i] * cut;
            small = small && cf[i] <= small_lim;
        }
        if (__all(small)) {
#pragma unroll
            for (int i = 0; i < U; i++) {
                const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
                lores_coeffs_sin_small(cf[i], nkr, pisr, c, r);
                o[i] = flt_lores(f, s, c, r) * av[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < U; i++) {
                const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
                lores_coeffs_sin(cf[i], kr, pisr, sr, c, r);
                o[i] = flt_lores(f, s, c, r) * av[i];
            }
        }
        emit(o);
    };
    int tn[U];
    GateGroup<U, int32_t> gcur;  // shared gate: see gate_group_load
    const auto gate_on = [](int32_t t) { return t == 1; };  // the test of C:1363 / C:1425
    if constexpr (TPV) {
#pragma unroll
        for (int i = 0; i < U; i++) {
            const size_t m = (size_t)i < N ? (size_t)i : N - 1;
            tn[i] = trig[m * V + v];
        }
    } else {
        gate_group_load(gcur, trig, N, 0, gate_on);
        asm volatile("" : "+v"(gcur.cls));
    }
    // PACE (mxg_pace.h): chunk k not before t0 + k P ticks of the 100 MHz counter, P from the launch's controller
    Pace pc;
    pc.start(pace_ctl, pace_arg);
    auto pace_wait = [&](bool is_cheap) { pc.wait(is_cheap); };
    if constexpr (MIX) __builtin_amdgcn_s_setprio(2);  // the store-bound stream is the critical one: the consumer takes the issue slots it leaves
    int fast_prev = 0;  // the previous chunk's steady state (1 sustain, 2 release, 0 neither or unknown)
    for (size_t n0 = 0; n0 < N; n0 += U) {
      if constexpr (MIX)
        if (n0 && (n0 & (size_t)(WIN - 1)) == 0) mixpc_window_close<WIN>(s_part, partial + wg * N * 2 + (n0 - WIN) * 2, WIN);
      int tc[U];
      int fast = 0;
      if constexpr (TPV) {
#pragma unroll
        for (int i = 0; i < U; i++) tc[i] = tn[i];
#pragma unroll
        for (int i = 0; i < U; i++) {
          const size_t m = (n0 + U + i < N) ? n0 + U + i : N - 1;
          tn[i] = trig[m * V + v];
        }
      } else {
        const int cc = (int)((n0 / U) & 63);
        if (cc == 0) {  // launches longer than 64 chunks: one drain of the store stream per 64*U samples
          if (n0) {
            gate_group_load(gcur, trig, N, n0 / (64 * U), gate_on);
          }
        }
        // (a sustain or release chunk moves no flag and no count of the envelope -- amplitude and output only -- so a wavefront found in
        // one of the two stays there while the gate's class does: the five-compare test runs where the class changes, not every chunk)
        const int g = lane_value(gcur.cls, cc);
        if (g > 0) fast = (DIET && fast_prev == 1) ? 1 : (__all(env_in_sustain(e)) ? 1 : 0);
        else if (g < 0) fast = (DIET && fast_prev == 2) ? 2 : (__all(env_in_release(e)) ? 2 : 0);
        fast_prev = fast;
      }
      if (fast) {  // steady envelope: see env_in_sustain
        auto steady = [&](auto sustain) {
            constexpr bool SUS = decltype(sustain)::value;
            pace_wait(MODE == 0 || SUS);  // (mode B's release chunks evaluate the coefficient pair per sample)
            if constexpr (MODE == 1 && SUS) {
                // the envelope value, hence (cutoff, c, r), is the same for every sample of the chunk
                lores_coeffs_sin((1.0 * e.amplitude) * cut, kr, pisr, sr, c, r);
            }
            // A release chunk taken speculatively (C:1460-1463: while amplitude > 0, amplitude *= release; output = input*amplitude): the
            // eight amplitudes are one multiply each; they are the reference's if its test held BEFORE each of them -- the incoming
            // amplitude and the first seven products > 0, which for a release factor > 0 is a test of the incoming one and the seventh
            // (the products of a factor <= 1 fall, those of a factor > 1 rise) -- on every lane.  One multiply more per sample for the
            // output, where the guarded tick (env_release_tick) takes a compare and four selects besides.
            double am[U];
            bool spec = false;
            if constexpr (!SUS && DIET) {
                double a = e.amplitude;
#pragma unroll
                for (int i = 0; i < U; i++) {
                    a = a * e.release;
                    am[i] = a;
                }
                spec = __all(e.amplitude > 0. && e.release > 0. && am[U - 2] > 0.);
            }
            if constexpr (MODE == 1 && !SUS) {  // release: the eight envelope values first, then the chunk
                double av[U];
                if (spec) {
#pragma unroll
                    for (int i = 0; i < U; i++) av[i] = am[i];  // output = 1.0 * amplitude
                    e.amplitude = am[U - 1];
                    e.output = am[U - 1];
                } else {
#pragma unroll
                    for (int i = 0; i < U; i++) av[i] = env_release_tick(e, 1.0);
                }
                modb_chunk(av);
                return;
            }
            if constexpr (MODE == 0 && !SUS) {
                if (spec) {
                    double os[U];
#pragma unroll
                    for (int i = 0; i < U; i++) {
                        const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
                        os[i] = flt_lores(f, s, c, r) * am[i];
                        if constexpr (!(PX || MIX)) {  // (8-byte streams: each store where its value is ready, not eight at the chunk's end)
                            store1<ST>(op, os[i]);
                            op += V;
                        }
                    }
                    e.amplitude = am[U - 1];
                    e.output = os[U - 1];
                    if constexpr (PX || MIX) emit(os);
                    return;
                }
            }
            double ov[U];
#pragma unroll
            for (int i = 0; i < U; i++) {
                double o;
                if constexpr (MODE == 0) {
                    const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
                    double y = flt_lores(f, s, c, r);
                    o = SUS ? env_sustain_tick(e, y) : env_release_tick(e, y);
                } else {
                    double a = SUS ? env_sustain_tick(e, 1.0) : env_release_tick(e, 1.0);
                    const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
                    if constexpr (!SUS) lores_coeffs_sin(a * cut, kr, pisr, sr, c, r);
                    double y = flt_lores(f, s, c, r);
                    o = y * a;
                }
                if constexpr (PX || MIX) {
                    ov[i] = o;
                } else {
                    store1<ST>(op, o);
                    op += V;
                }
            }
            if constexpr (PX || MIX) emit(ov);
        };
        // RUNS (round 6, DIET): the gate's class of the group's 64 chunks sits in the lanes of gcur.cls, so the number of chunks from
        // this one on that keep its class is a count of trailing ones -- and a sustain or release chunk leaves the wavefront in its
        // state (see above): the whole run goes through ONE tight loop, a back-edge per chunk where the chunk-by-chunk form takes
        // half a dozen branches, a v_readlane and the scalar tests around them (a lone wavefront refetches after every taken branch).
        // A run ends with its group (512 samples = the mixdown form's window) at the latest; ragged chunks have class 0.
        int run = 1;
        if constexpr (DIET && !TPV && MXG_VOICE_RUNS) {
            const int cc = (int)((n0 / U) & 63);
            const unsigned long long same = (fast == 1 ? __ballot(gcur.cls > 0) : __ballot(gcur.cls < 0)) >> cc;  // (bit 0: this chunk)
            const unsigned long long stop = ~same;
            run = stop ? __builtin_ctzll(stop) : 64;
            if (run > 64 - cc) run = 64 - cc;
        }
        if (fast == 1) {
            for (int q = 0; q < run; q++) steady(std::true_type{});
        } else {
            for (int q = 0; q < run; q++) steady(std::false_type{});
        }
        n0 += (size_t)(run - 1) * U;
        continue;
      }
      pace_wait(false);
      if constexpr (MODE == 0) {
        // The general steady chunk (mxg_env.h): the oscillator and the filter do not depend on the envelope, so their 8
        // samples are formed first; the envelope then takes them in one speculative chunk if every lane of the wavefront stays
        // inside a stage of its own (attack, decay, hold ... each lane a different one), else sample by sample.
        if (n0 + U <= N) {
            bool lane_gate, constant;
            if constexpr (TPV) {
                lane_gate = tc[0] == 1;
                constant = true;
#pragma unroll
                for (int i = 1; i < U; i++) constant = constant && ((tc[i] == 1) == lane_gate);
            } else {
                const int g = lane_value(gcur.cls, (int)((n0 / U) & 63));
                lane_gate = g > 0;
                constant = g != 0;
            }
            if (__all(constant)) {
                double y[U], o[U];
#pragma unroll
                for (int i = 0; i < U; i++) {
                    const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
                    y[i] = flt_lores(f, s, c, r);
                }
                Env s2 = e;
                const bool ok = env_steady_chunk<U>(s2, y, lane_gate, o);
                if (__all(ok)) {
                    e = s2;
                } else {
#pragma unroll
                    for (int i = 0; i < U; i++) {
                        int t;
                        if constexpr (TPV) t = tc[i];
                        else t = lane_value(gcur.g[i], (int)((n0 / U) & 63));
                        o[i] = env_adsr(e, y[i], t);
                    }
                }
                emit(o);
                continue;
            }
        }
      }
      if constexpr (MODE == 1) {
        // The same speculative chunk for the modulated voice (round 6): here the ENVELOPE comes first (14.monosynth/main.cpp:50-55: the
        // envelope's output scales the cutoff), so its eight values are formed in one steady chunk (input 1.0: output = amplitude) if
        // every lane stays inside a stage of its own, and the oscillator, the coefficients and the filter follow sample by sample.
        if (n0 + U <= N) {
            bool lane_gate, constant;
            if constexpr (TPV) {
                lane_gate = tc[0] == 1;
                constant = true;
#pragma unroll
                for (int i = 1; i < U; i++) constant = constant && ((tc[i] == 1) == lane_gate);
            } else {
                const int g = lane_value(gcur.cls, (int)((n0 / U) & 63));
                lane_gate = g > 0;
                constant = g != 0;
            }
            if (__all(constant)) {
                const double ones[U] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
                double av[U];
                Env s2 = e;
                const bool ok = env_steady_chunk<U>(s2, ones, lane_gate, av);
                if (__all(ok)) {
                    e = s2;
                    modb_chunk(av);
                    continue;
                }
            }
        }
      }
      double og[U];
      const bool whole = n0 + U <= N;  // (wave-uniform)
#pragma unroll
      for (int i = 0; i < U; i++) {
        const size_t n = n0 + i;
        if (n >= N) break;
        int t;
        if constexpr (TPV) t = tc[i];
        else t = lane_value(gcur.g[i], (int)((n0 / U) & 63));
        double o;
        if constexpr (MODE == 0) {
            const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
            double y = flt_lores(f, s, c, r);
            o = env_adsr(e, y, t);
        } else {
            double a = env_adsr(e, 1.0, t);
            const double s = saw_tick<DIET>(phase, hold, inc);  // C:333-340
            lores_coeffs_sin(a * cut, kr, pisr, sr, c, r);
            double y = flt_lores(f, s, c, r);
            o = y * a;
        }
        if ((PX || MIX) && whole) {
            og[i] = o;
        } else {  // (a ragged last chunk goes out sample by sample, whatever the store stream)
            put1(o);
        }
      }
      if constexpr (PX || MIX)
        if (whole) emit(og);
    }
    if constexpr (MIX) {
        tile_flush();
        const size_t w0 = (N - 1) / WIN * WIN;
        mixpc_window_close<WIN>(s_part, partial + wg * N * 2 + w0 * 2, (int)(N - w0));
        __builtin_amdgcn_s_setprio(0);
    }
    ost[v] = phase;
    ost[V + v] = hold;
    fst[v] = f.x;
    fst[V + v] = f.y;
    fst[2 * V + v] = f.o0;
    fst[3 * V + v] = f.o1;
    fst[4 * V + v] = f.o2;
    env_store(e, V, v, dst, ist);
    if ((threadIdx.x & (MIX ? 255 : ~0u)) == 0 && (!MIX || threadIdx.x < 256)) pc.finish(pace_ctl, pace_arg, (unsigned)wg, gridDim.x);
}

inline dim3 grid_for(size_t V, int block) { return dim3((unsigned)((V + block - 1) / block)); }

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

// (diagnostics, not in maxigpu.h: the four pace controllers of a stream -- mode A / B x plain / mixdown, kPaceWords words each -- copied
// to the host; tools/probes/pace_ctl.py)
int mxg_debug_voice_pace(void *stream, unsigned *host) {
    hipStream_t st = resolve_stream(stream);
    unsigned *base = nullptr;
    bool fresh = false;
    if (int s = scratch_get(SCR_VOICE_PACE, st, 4 * kPaceWords * sizeof(unsigned), (void **)&base, &fresh)) return s;
    if (fresh) MXG_HIP(hipMemsetAsync(base, 0, 4 * kPaceWords * sizeof(unsigned), st));
    MXG_HIP(hipStreamSynchronize(st));
    return check_hip(hipMemcpy(host, base, 4 * kPaceWords * sizeof(unsigned), hipMemcpyDeviceToHost), "pace words");
}

int mxg_filter_render(int kind, size_t V, size_t N, const double *d_in, const double *d_cutoff,
                      int cps, const double *d_res, int rps, const double *d_coef, double *d_st,
                      double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(kind >= 0 && kind <= 4, "unknown filter kind");
    MXG_REQUIRE(d_in && d_cutoff && d_st && d_out, "null device pointer");
    MXG_REQUIRE(kind > MXG_FLT_BANDPASS || d_res, "lores/hires/bandpass need d_res");
    const bool mod = (cps || rps);
    MXG_REQUIRE(kind > MXG_FLT_BANDPASS || mod || d_coef,
                "constant-parameter lores/hires/bandpass need d_coef from mxg_filter_coeffs_host");
    if (V == 0 || N == 0) return MXG_OK;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;  // the bank kernels are compiled for <= 256 lanes per workgroup (512 VGPRs/lane budget)
    hipStream_t st = resolve_stream(stream);
    double sr = (double)settings().sampleRate;
    if (!mod && (kind == MXG_FLT_LORES || kind == MXG_FLT_HIRES) && scan_applies(V, N))  // tolerance mode (scan.hip)
        return scan_filter_launch(kind == MXG_FLT_LORES ? 3 : 4, V, N, d_in, d_coef, d_st, d_out, st);
    // 16-byte pair-row streams for the block-constant forms (knob rw_store, as mxg_filter2_render: 0 automatic = write-through stores
    // for blocks from 64 MB, 1 off, 2 / 3 / 4 plain / write-through / non-temporal stores)
    {
        int rw = tune_get("rw_store");
        // (filter_pairs_kernel addresses its rows with 32-bit byte offsets from a wave-uniform base: rows of 2 GiB and more -- 2^28
        // voices -- keep the 8-byte kernel, whose pointers are 64-bit)
        const bool pairs_ok = !mod && !(V & 1) && !(N & 1) && !(((uintptr_t)d_in) & 15) && !(((uintptr_t)d_out) & 15) &&
                              2 * (unsigned long long)V * sizeof(double) < (1ull << 32);
        if (rw == 0) rw = (V * N * sizeof(double) >= ((size_t)64 << 20)) ? 3 : 1;
        if (rw >= 2 && pairs_ok) {
            KernelTimer kt("filter_kernel", st);
#define MXG_FLP2(K, S)                                                                                                              \
    if (chunk == 4) hipLaunchKernelGGL((filter_pairs_kernel<K, S, 4>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_cutoff, d_coef, d_st, d_out); \
    else if (chunk == 32) hipLaunchKernelGGL((filter_pairs_kernel<K, S, 32>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_cutoff, d_coef, d_st, d_out); \
    else if (chunk == 16) hipLaunchKernelGGL((filter_pairs_kernel<K, S, 16>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_cutoff, d_coef, d_st, d_out); \
    else hipLaunchKernelGGL((filter_pairs_kernel<K, S, 8>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_cutoff, d_coef, d_st, d_out);
#define MXG_FLP(K)                     \
    if (rw == 2) { MXG_FLP2(K, 0) }    \
    else if (rw == 3) { MXG_FLP2(K, 2) } \
    else { MXG_FLP2(K, 1) }
            int chunk = tune_get("rw_chunk");  // samples per chunk of the pair-row kernels: 0 automatic, 8 / 16 / 32
            if (chunk == 0) chunk = 8;
            switch (kind) {
                case 0: MXG_FLP(0) break;
                case 1: MXG_FLP(1) break;
                case 2: MXG_FLP(2) break;
                case 3: MXG_FLP(3) break;
                default: MXG_FLP(4) break;
            }
#undef MXG_FLP2
#undef MXG_FLP
            return check_hip(hipGetLastError(), "filter_pairs_kernel launch");
        }
    }
#define MXG_FLT_LAUNCH(K)                                                                        \
    if (mod)                                                                                     \
        hipLaunchKernelGGL((filter_kernel<K, true>), grid_for(V, block), dim3(block), 0, st, V, N, \
                           d_in, d_cutoff, cps, d_res, rps, d_coef, d_st, d_out, sr);            \
    else                                                                                         \
        hipLaunchKernelGGL((filter_kernel<K, false>), grid_for(V, block), dim3(block), 0, st, V, N, \
                           d_in, d_cutoff, cps, d_res, rps, d_coef, d_st, d_out, sr);
    KernelTimer kt("filter_kernel", st);
    switch (kind) {
        case 0: MXG_FLT_LAUNCH(0) break;
        case 1: MXG_FLT_LAUNCH(1) break;
        case 2: MXG_FLT_LAUNCH(2) break;
        case 3: MXG_FLT_LAUNCH(3) break;
        case 4: MXG_FLT_LAUNCH(4) break;
    }
#undef MXG_FLT_LAUNCH
    return check_hip(hipGetLastError(), "filter_kernel launch");
}

int mxg_filter_render_coefs(int kind, size_t V, size_t N, const double *d_in, const double *d_coef_ps, double *d_st, double *d_out,
                            void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(kind >= MXG_FLT_LORES && kind <= MXG_FLT_BANDPASS, "lores, hires or bandpass (lopass / hipass take a per-sample cutoff in mxg_filter_render)");
    MXG_REQUIRE(d_in && d_coef_ps && d_st && d_out, "null device pointer");
    if (V == 0 || N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("filter_coefps_kernel", st);
    const int block = 256;
    switch (kind) {
        case 0: hipLaunchKernelGGL((filter_coefps_kernel<0>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_coef_ps, d_st, d_out); break;
        case 1: hipLaunchKernelGGL((filter_coefps_kernel<1>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_coef_ps, d_st, d_out); break;
        default: hipLaunchKernelGGL((filter_coefps_kernel<2>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_coef_ps, d_st, d_out); break;
    }
    return check_hip(hipGetLastError(), "filter_coefps_kernel launch");
}

// Host libm evaluation of C:456-461 / C:489-495 (no device involved).
int mxg_filter_coeffs_host(int kind, size_t V, const double *h_cutoff, const double *h_res,
                           double *h_coef) {
    MXG_REQUIRE(h_cutoff && h_res && h_coef, "null pointer");
    MXG_REQUIRE(kind >= MXG_FLT_LORES && kind <= MXG_FLT_BANDPASS, "kind has no coefficients");
    const size_t sr = settings().sampleRate;
    for (size_t v = 0; v < V; v++) {
        double cutoff = h_cutoff[v], resonance = h_res[v];
        if (kind == MXG_FLT_BANDPASS) {
            if (cutoff > (sr * 0.5)) cutoff = (sr * 0.5);
            if (resonance >= 1.) resonance = 0.999999;
            double z = cos(MXG_TWOPI * cutoff / sr);
            h_coef[v] = (1 - resonance) * (sqrt(resonance * (resonance - 4.0 * pow(z, 2.0) + 2.0) + 1));
            h_coef[V + v] = 2 * z * resonance;
            h_coef[2 * V + v] = pow((resonance * -1), 2);
        } else {
            if (cutoff < 10) cutoff = 10;
            if (cutoff > (sr)) cutoff = (sr);
            if (resonance < 1.) resonance = 1.;
            double z = cos(MXG_TWOPI * cutoff / sr);
            h_coef[v] = 2 - 2 * z;
            h_coef[V + v] = (sqrt(2.0) * sqrt(-pow((z - 1.0), 3.0)) + resonance * (z - 1)) /
                            (resonance * (z - 1));
            h_coef[2 * V + v] = 0.0;
        }
    }
    return MXG_OK;
}

int mxg_env_render(int mode, size_t V, size_t N, const double *d_in, const int32_t *d_trig,
                   int tpv, const double *d_par, const int64_t *d_holdtime, double *d_dst,
                   int64_t *d_ist, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (adsr) or 1 (ar)");
    MXG_REQUIRE(d_trig && d_par && d_holdtime && d_dst && d_ist && d_out, "null device pointer");
    if (V == 0 || N == 0) return MXG_OK;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;  // the bank kernels are compiled for <= 256 lanes per workgroup (512 VGPRs/lane budget)
    hipStream_t st = resolve_stream(stream);
    // 16-byte pair-row streams (knob rw_store, as mxg_filter_render: 0 automatic = write-through stores for blocks from 64 MB, 1 off,
    // 2 / 3 / 4 plain / write-through / non-temporal stores)
    int rw = tune_get("rw_store");
    const bool pairs_ok = V >= 2 && !(V & 1) && !(N & 1) && !(((uintptr_t)d_in) & 15) && !(((uintptr_t)d_out) & 15);
    if (rw == 0) rw = (V * N * sizeof(double) >= ((size_t)64 << 20)) ? 3 : 1;
    const int px = (rw >= 2 && pairs_ok) ? rw - 1 : 0;
#define MXG_ENV_LAUNCH(M, I, P, X)                                                                      \
    hipLaunchKernelGGL((env_kernel<M, I, P, X>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_trig, \
                       tpv, d_par, d_holdtime, d_dst, d_ist, d_out)
#define MXG_ENV_LAUNCH1(M, I, P)                        \
    switch (px) {                                       \
        case 1: MXG_ENV_LAUNCH(M, I, P, 1); break;      \
        case 2: MXG_ENV_LAUNCH(M, I, P, 2); break;      \
        case 3: MXG_ENV_LAUNCH(M, I, P, 3); break;      \
        default: MXG_ENV_LAUNCH(M, I, P, 0); break;     \
    }
#define MXG_ENV_LAUNCH2(M)                                                  \
    if (d_in) {                                                             \
        if (tpv) { MXG_ENV_LAUNCH1(M, true, true) } else { MXG_ENV_LAUNCH1(M, true, false) }   \
    } else {                                                                \
        if (tpv) { MXG_ENV_LAUNCH1(M, false, true) } else { MXG_ENV_LAUNCH1(M, false, false) } \
    }
    KernelTimer kt("env_kernel", st);
    if (mode == 0) { MXG_ENV_LAUNCH2(0) } else { MXG_ENV_LAUNCH2(1) }
#undef MXG_ENV_LAUNCH2
#undef MXG_ENV_LAUNCH1
#undef MXG_ENV_LAUNCH
    return check_hip(hipGetLastError(), "env_kernel launch");
}

// maxiEnv setters, host libm (C:1469-1494).
double mxg_env_coeff_host(int which, double ms) {
    const size_t sr = settings().sampleRate;
    switch (which) {
        case 0: return 1 - pow(0.01, 1.0 / (ms * sr * 0.001));  // setAttack   C:1480-1482
        case 1: return pow(0.01, 1.0 / (ms * sr * 0.001));      // setDecay    C:1475-1477
        case 2: return pow(0.01, 1.0 / (ms * sr * 0.001));      // setRelease  C:1470-1472
        case 3: return 1.0 / (ms / 1000.0 * sr);                // setAttackMS C:1486-1488
    }
    return 0.0;
}

// maxiConvert::mtof (H:941, C:1498-1500): mtofarray[midinote], the reference's literals (maxi_tables.h, generated from the
// compiled reference)
double mxg_mtof_host(int midinote) { return (midinote >= 0 && midinote <= 128) ? MAXI_MTOF[midinote] : 0.0; }

// The launch of K2f: d_pan / d_rows null = the block only; both given = the mixdown form (MIX, see voice_kernel), d_out optional.
static int voice_launch(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff, const double *d_res,
                        const double *d_coef, const int32_t *d_trig, int tpv, const double *d_par, const int64_t *d_holdtime,
                        double *d_ost, double *d_fst, double *d_dst, int64_t *d_ist, double *d_out, const double *d_pan, double *d_rows,
                        hipStream_t st) {
    const bool mix = d_rows != nullptr;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;  // the bank kernels are compiled for <= 256 lanes per workgroup (512 VGPRs/lane budget)
    // the store stream, as K1's (osc.hip, profiles/r03_osc_store.md): knob voice_store 0 = automatic (pair rows of write-through
    // 16-byte stores wherever whole pairs exist; XCD-contiguous numbering from 262 144 voices), 1 / 2 plain / nt 8-byte stores
    // (0 with voice_nt set: the round-2 rule), 3 / 4 / 5 pair rows plain / sc1 / nt
    const int nt_knob = tune_get("voice_nt");
    const size_t out_bytes = V * N * sizeof(double);
    const bool pairs_ok = !(V & 1) && !(((uintptr_t)d_out) & 15) && V >= 2;
    int store = tune_get("voice_store") - 1, xcd = tune_get("voice_xcd") - 1;  // (knob value 0 = automatic)
    if (store < 0) {
        // automatic, from tools/sweep_voice_store.py (profiles/r03_voice_store.md; rotated destinations, sustain segment, fraction of
        // 8 TB/s on 8.34 B per sample): below 49 152 voices plain stores (the block lives in the caches); 65 536: pair rows of
        // non-temporal 16-byte stores (54.5 -> 48.4 us, 0.64 -> 0.72); 98 304 ... 196 607: pair rows write-through, XCD-contiguous
        // (131 072: 113 -> 100 us, 0.62 -> 0.70); from 196 608: non-temporal 8-byte stores, XCD-contiguous (524 288: 460 -> 393 us,
        // 0.61 -> 0.71)
        if (nt_knob != 2) store = nt_knob == 1 ? 1 : 0;
        else if (out_bytes < ((size_t)192 << 20)) store = 0;
        // (round 6, re-measured on the current kernel, two boxes, interleaved: at 65 536 voices non-temporal 8-BYTE stores beat the pair rows
        // for mode A -- 47.2-47.7 against 48.1-48.7 us -- and clearly for the mixdown form, whose producers share their SIMDs with the
        // consumers -- 50.4-50.6 against 53.1-53.3 us; mode B is indifferent, 60.6-62.1 against 60.7-61.5, and keeps the pair rows)
        else if (V < 98304) store = (mode == 0 || mix) ? 1 : (pairs_ok ? 4 : 1);
        else if (V < 196608) store = pairs_ok ? 3 : 1;
        else store = 1;
        if (xcd < 0) xcd = V >= 98304 ? 1 : 0;
    }
    if (store >= 2 && !pairs_ok) store = store == 4 ? 1 : 0;
    if (xcd < 0) xcd = 0;
    if (mix) {
        // the mixdown form's own store knob (voice_mix_store: 0 automatic = the rule above, 1 ... 5 = voice_store's flavours)
        const int ms = tune_get("voice_mix_store") - 1;
        if (ms >= 0) store = (ms >= 2 && !pairs_ok) ? (ms == 4 ? 1 : 0) : ms;
        if (!d_out) store = 5;  // no per-voice block
        block = 512;            // 256 voices per workgroup: four producer + four consumer wavefronts
    }
    const dim3 grid = mix ? grid_for(V, 256) : grid_for(V, block);
    double sr = (double)settings().sampleRate;
#define MXG_VOICE_LAUNCH(M, S, X, P, MX, D)                                                            \
    hipLaunchKernelGGL((voice_kernel<M, S, X, P, MX, D>), grid, dim3(block), 0, st, V, N, d_freq, \
                       d_cutoff, d_res, d_coef, d_trig, tpv, d_par, d_holdtime, d_ost, d_fst,   \
                       d_dst, d_ist, d_out, sr, xcd, d_pan, d_rows, pace_ctl, pace_arg)
#define MXG_VOICE_LAUNCH1(M, S, X, P, MX) \
    if ((M) == 1 || (MX) || diet) MXG_VOICE_LAUNCH(M, S, X, P, MX, true); else MXG_VOICE_LAUNCH(M, S, X, P, MX, ((M) == 1 || (MX)))
#define MXG_VOICE_LAUNCH2(M, S, X, MX) \
    if (tpv) MXG_VOICE_LAUNCH1(M, S, X, true, MX); else MXG_VOICE_LAUNCH1(M, S, X, false, MX)
#define MXG_VOICE_LAUNCH3(M, MX)                           \
    switch (store) {                                       \
        case 1: MXG_VOICE_LAUNCH2(M, 1, false, MX); break; \
        case 2: MXG_VOICE_LAUNCH2(M, 0, true, MX); break;  \
        case 3: MXG_VOICE_LAUNCH2(M, 2, true, MX); break;  \
        case 4: MXG_VOICE_LAUNCH2(M, 1, true, MX); break;  \
        default: MXG_VOICE_LAUNCH2(M, 0, false, MX); break; \
    }
    // the fast paths' instruction diet (voice_kernel, DIET): always for mode B and the mixdown form; plain mode A unless the knob
    // voice_diet says 1 (the round-5 instruction stream, kept for comparison)
    // the paced schedule (voice_kernel, PACE): knob voice_pace 0 = automatic, 1 = never, >= 2 = a fixed period of that many 10 ns ticks
    // per 8-sample chunk (sweeps).  Automatic = the controller wherever the store stream is the bound AND the whole grid is resident at
    // once (a schedule per workgroup means nothing to workgroups that wait for a CU): from 45 056 voices (below, the kernel's time is
    // its own instruction stream's: 40 960 voices 32.7 -> 33.3 us) up to 262 144 for mode A (five wavefronts of 92 registers per SIMD: four
    // workgroups per CU; 393 216 voices on fixed periods: 313 -> 426 us), 131 072 for mode B (two of 160-190), 65 536 for the mixdown form (one workgroup of
    // 146 KB of LDS per CU: at 131 072 voices half the grid waits, 110 -> 142 us).  Measured with the controller, mode A:
    // 49 152 voices 36.1 -> 32.7 us, 65 536 51.3 -> 42.4 (the round-5 stream without it: 47.7), 81 920 65.7 -> 54.5, 98 304 74.4 -> 68.4,
    // 131 072 103.0 -> 87.7, 196 608 166.5 -> 132.8: the collapse of the store stream is not a matter of one wavefront per SIMD.
    const int pace_knob = tune_get("voice_pace");
    unsigned *pace_ctl = nullptr;
    unsigned pace_arg = 0;
    if (pace_knob >= 2) {
        pace_arg = (unsigned)pace_knob;
    } else if (pace_knob == 0 && V >= 45056 && V <= (mix ? (size_t)65536 : (mode ? (size_t)131072 : (size_t)294911))) {
        // the starting period: the chip's 8 rows at 6.6 TB/s, in ticks of the device's constant counter (pace_start_period); one controller per stream and form
        pace_arg = pace_start_period(V * 8 * 8);
        unsigned *base = pace_words(SCR_VOICE_PACE, st, 4 * kPaceWords);
        if (base && pace_arg) pace_ctl = base + kPaceWords * ((mode ? 1 : 0) + (mix ? 2 : 0));
        else pace_arg = 0;  // (inside a graph capture before the first eager launch, or a device whose counter's rate is unknown: not paced)
        // (a paced launch wants the NATURAL workgroup numbering -- the whole chip walking down the same rows: the XCD-contiguous one was
        // round 3's answer to the free-running kernel's drift at large banks.  Fixed periods, mode A: 98 304 voices 67.2 -> 64.4 us,
        // 196 608 146 -> 142, 262 144 192 -> 171 where the free-running kernel takes 198: profiles/r06_pace.md)
        if (pace_ctl && tune_get("voice_xcd") == 0) xcd = 0;
    }
    // (plain mode A where every SIMD holds one wavefront and the launch could NOT be paced -- a capture before the stream's first eager
    // launch, a device that is not the whole 256-CU chip, the knob: the short stream by itself is the slower one there, 50-51 us against
    // the round-5 stream's 47.7, so that launch keeps the round-5 stream)
    const int diet_knob = tune_get("voice_diet");
    const bool diet = diet_knob == 2 || (diet_knob == 0 && !(!pace_arg && V >= 57344 && V < 73728));
    KernelTimer kt("voice_kernel", st);
    if (mix) {
        if (store == 5) {
            if (mode == 0) { MXG_VOICE_LAUNCH2(0, 3, false, true); } else { MXG_VOICE_LAUNCH2(1, 3, false, true); }
        } else if (mode == 0) {
            MXG_VOICE_LAUNCH3(0, true)
        } else {
            MXG_VOICE_LAUNCH3(1, true)
        }
    } else if (mode == 0) {
        MXG_VOICE_LAUNCH3(0, false)
    } else {
        MXG_VOICE_LAUNCH3(1, false)
    }
#undef MXG_VOICE_LAUNCH3
#undef MXG_VOICE_LAUNCH2
#undef MXG_VOICE_LAUNCH1
#undef MXG_VOICE_LAUNCH
    return check_hip(hipGetLastError(), "voice_kernel launch");
}

int mxg_voice_render(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff,
                     const double *d_res, const double *d_coef, const int32_t *d_trig, int tpv,
                     const double *d_par, const int64_t *d_holdtime, double *d_ost, double *d_fst,
                     double *d_dst, int64_t *d_ist, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 or 1");
    MXG_REQUIRE(d_freq && d_trig && d_par && d_holdtime && d_ost && d_fst && d_dst && d_ist && d_out,
                "null device pointer");
    MXG_REQUIRE(mode == 1 || d_coef, "mode 0 needs d_coef from mxg_filter_coeffs_host");
    MXG_REQUIRE(mode == 0 || (d_cutoff && d_res), "mode 1 needs d_cutoff and d_res");
    if (V == 0 || N == 0) return MXG_OK;
    return voice_launch(mode, V, N, d_freq, d_cutoff, d_res, d_coef, d_trig, tpv, d_par, d_holdtime, d_ost, d_fst, d_dst, d_ist, d_out,
                        nullptr, nullptr, resolve_stream(stream));
}

// K2f + the fused maxiMix::stereo mixdown (C:503-509 per voice, the sum over voices of 15.polysynth/main.cpp:54-70): the block (d_out,
// optional) and the per-workgroup rows d_rows[mxg_osc_mix_groups(V)][N][2] -- the layout of mxg_osc_render_mix_rows, what a grouped mix
// queue's slot takes (mxg_mixq_create_grouped).
int mxg_voice_render_mix_rows(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff, const double *d_res,
                              const double *d_coef, const int32_t *d_trig, int tpv, const double *d_par, const int64_t *d_holdtime,
                              double *d_ost, double *d_fst, double *d_dst, int64_t *d_ist, double *d_out, const double *d_pan,
                              double *d_rows, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 or 1");
    MXG_REQUIRE(d_freq && d_trig && d_par && d_holdtime && d_ost && d_fst && d_dst && d_ist && d_pan && d_rows, "null device pointer");
    MXG_REQUIRE(mode == 1 || d_coef, "mode 0 needs d_coef from mxg_filter_coeffs_host");
    MXG_REQUIRE(mode == 0 || (d_cutoff && d_res), "mode 1 needs d_cutoff and d_res");
    if (V == 0 || N == 0) return MXG_OK;
    return voice_launch(mode, V, N, d_freq, d_cutoff, d_res, d_coef, d_trig, tpv, d_par, d_holdtime, d_ost, d_fst, d_dst, d_ist, d_out,
                        d_pan, d_rows, resolve_stream(stream));
}

// The same with the mix itself as the result, d_mix[N][2]: rows in per-stream scratch + the row sum on the caller's stream.
int mxg_voice_render_mix(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff, const double *d_res,
                         const double *d_coef, const int32_t *d_trig, int tpv, const double *d_par, const int64_t *d_holdtime,
                         double *d_ost, double *d_fst, double *d_dst, int64_t *d_ist, double *d_out, const double *d_pan, double *d_mix,
                         void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 or 1");
    MXG_REQUIRE(d_freq && d_trig && d_par && d_holdtime && d_ost && d_fst && d_dst && d_ist && d_pan && d_mix, "null device pointer");
    MXG_REQUIRE(mode == 1 || d_coef, "mode 0 needs d_coef from mxg_filter_coeffs_host");
    MXG_REQUIRE(mode == 0 || (d_cutoff && d_res), "mode 1 needs d_cutoff and d_res");
    if (N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const size_t groups = (V + 255) / 256;
    if (V == 0) return check_hip(hipMemsetAsync(d_mix, 0, N * 2 * sizeof(double), st), "voice mix memset");
    if (groups == 1)  // a bank of one workgroup: its row IS the mix
        return voice_launch(mode, V, N, d_freq, d_cutoff, d_res, d_coef, d_trig, tpv, d_par, d_holdtime, d_ost, d_fst, d_dst, d_ist, d_out,
                            d_pan, d_mix, st);
    double *rows = nullptr;  // per-stream scratch: [groups][N][2]
    if (int s = scratch_get(SCR_OSC_MIX, st, sizeof(double) * (N * groups * 2 + 2), (void **)&rows)) return s;
    if (int s = voice_launch(mode, V, N, d_freq, d_cutoff, d_res, d_coef, d_trig, tpv, d_par, d_holdtime, d_ost, d_fst, d_dst, d_ist, d_out,
                             d_pan, rows, st))
        return s;
    return mxg_mix_rows_sum(groups, N * 2, rows, d_mix, stream);
}

}  // extern "C"
