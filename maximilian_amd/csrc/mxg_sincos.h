// mxg_sincos.h -- sin/cos of (phase * TWOPI) for maxiOsc::sinewave / coswave (C:230, C:278).
//
// Two routines.  The oscillators use the TABLE form at the end of this file (sincos_tab: 512-entry double-double table in LDS,
// correctly rounded on every argument tested, ~32 instructions).  The Cody-Waite + fdlibm-kernel form described next (sin_small /
// cos_small, no table) came first; it remains for callers without a table at hand (the modulated lores / bandpass coefficients of
// the fused voice) and is measured by the same host test.
//
// The reference evaluates glibc sin()/cos() on the ROUNDED product x = phase*TWOPI.  glibc's
// result is within 1 ULP of the true value (in practice correctly rounded almost always).
// The device path must therefore produce sin(x) to well under 1 ULP so the two differ by at
// most 1 ULP per sample (the contract stated in DESIGN.md; measured in tests/test_gpu_osc.py).
//
// An oscillator's argument is small (phase lives in [0, 1+inc), so x in [0, ~2*pi]); the generic
// device sin() carries a large-argument reduction it never needs here.  Fast path for |x| <= 64
// (phase up to ~10: audio-rate FM included): a three-constant Cody-Waite reduction by pi/2 (k <= 41, so k*pio2_hi
// is exact) that keeps a
// (hi, lo) remainder, then the classic minimax kernels for sin and cos on [-pi/4, pi/4] (the
// coefficient sets published with Sun's fdlibm, k_sin.c / k_cos.c), evaluated with FMAs -- this is
// libm-internal arithmetic, not one of the reference's expression trees.  Next to a zero of the function (remainder
// below 2^-18) the result's own ULP shrinks below that reduction's absolute error (~3e-25, the rounding of k*pio2_lo),
// so those arguments -- a wave-rare case -- repeat the reduction with fdlibm's three-iteration form (pi/2 in 33-bit
// pieces whose products with k are exact: 151 bits), which keeps the RELATIVE error of the remainder at 2^-70 or better
// even for x = fl(k*pi/2).  Error < 0.85 ULP of the result everywhere, measured against quad-precision sinq/cosq by
// tests/host_sincos_accuracy.cpp (0.77 on 8 M arguments): below 1 ULP, so the result and glibc's are the two doubles
// bracketing the true value -- at most 1 ULP apart, with no absolute-error escape near the zero crossings.
// Anything else (|x| > 64, NaN, Inf) goes to the device's generic sin()/cos().
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#else  // host build of the same text: tests/host_sincos_accuracy.cpp measures the error bound on the CPU
#include <math.h>
#define __device__
#define __forceinline__ inline
#define MXG_TWOPI 6.283185307179586476925286766559
#endif

namespace mxg {

namespace sincos_detail {
constexpr double kInvPio2 = 6.36619772367581382433e-01;  // 2/pi
constexpr double kPio2Hi = 1.57079632673412561417e+00;   // first 33 bits of pi/2
constexpr double kPio2Lo = 6.07710050650619224932e-11;   // pi/2 - kPio2Hi
constexpr double kPio2Lo2 = 3.52155982182414973774e-27;  // pi/2 - kPio2Hi - kPio2Lo (third piece; round 1 had fdlibm's
                                                         // pio2_2t here, the tail after the 33-bit SECOND piece: k*2e-21 off)
// fdlibm e_rem_pio2.c: pi/2 = kPio2Hi + kPio2_2 + kPio2_3 + kPio2_3t, the first three 33 bits each
constexpr double kPio2_2 = 6.07710050630396597660e-11, kPio2_2t = 2.02226624879595063154e-21;
constexpr double kPio2_3 = 2.02226624871116645580e-21, kPio2_3t = 8.47842766036889956997e-32;
constexpr double kSmallRem = 0x1p-18;  // below this the one-shot reduction's absolute error shows in the result's ULP
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

// One fused multiply-add, a*b + c.  On the device it is spelled as the three-address VOP3 instruction: left to itself hipcc
// turns a Horner chain over constants into v_fmac (two-address) and re-copies every coefficient into the accumulator register
// first -- 17 moves next to the 9 FMAs of the two chains below, a fifth of sinewave's instructions at one wavefront per SIMD.
// The coefficient goes in as the instruction's one scalar operand (`k`: wave-uniform, an SGPR pair set up outside the loop).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_k(double a, double b, double k) {  // a*b + k
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}
__device__ __forceinline__ double fma_kk(double a, double k1, double k0) {  // a*k1 + k0
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k1), "v"(k0));
    return d;
}
#else
__device__ __forceinline__ double fma_k(double a, double b, double k) { return fma(a, b, k); }
__device__ __forceinline__ double fma_kk(double a, double k1, double k0) { return fma(a, k1, k0); }
#endif

// sin and cos of y + t on [-pi/4, pi/4] (t = tail of the reduced argument)
__device__ __forceinline__ double k_sin(double y, double t) {
#pragma clang fp contract(fast)
    const double z = y * y, v = z * y;
    const double r = fma_k(z, fma_k(z, fma_k(z, fma_kk(z, S6, S5), S4), S3), S2);
    return y - ((z * (0.5 * t - v * r) - t) - v * S1);
}
__device__ __forceinline__ double k_cos(double y, double t) {
#pragma clang fp contract(fast)
    const double z = y * y;
    const double r = z * fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_kk(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - y * t));
}
// x = k*pi/2 + (y + t), |y| <= pi/4 (+ a hair); valid for |x| <= 64
__device__ __forceinline__ int reduce(double x, double &y, double &t) {
#pragma clang fp contract(fast)
    const double fn = rint(x * kInvPio2);
    const double z = x - fn * kPio2Hi;  // exact: fn has <= 6 bits, kPio2Hi 33 bits
    const double w = fn * kPio2Lo;
    y = z - w;
    t = ((z - y) - w) - fn * kPio2Lo2;
    return (int)fn;
}
// The same remainder to 151 bits of pi/2 (fdlibm's e_rem_pio2.c pieces): every fn*piece product is exact
// (33 + 6 bits) and every subtraction is an error-free TwoSum, so y + t = x - fn*pi/2 to ~2^-150 absolute.
__device__ __forceinline__ void two_diff(double a, double b, double &s, double &e) {
#pragma clang fp contract(off)
    s = a - b;
    const double bb = a - s;  // the part of b that was actually subtracted
    e = (a - (s + bb)) + (bb - b);
}
__device__ __forceinline__ void reduce_accurate(double x, double &y, double &t) {
#pragma clang fp contract(off)
    const double fn = rint(x * kInvPio2);
    const double r1 = x - fn * kPio2Hi;  // exact
    double r2, e2, r3, e3;
    two_diff(r1, fn * kPio2_2, r2, e2);
    two_diff(r2, fn * kPio2_3, r3, e3);
    const double tail = (e2 + e3) - fn * kPio2_3t;
    y = r3 + tail;
    t = (r3 - y) + tail;
}
}  // namespace sincos_detail

__device__ __forceinline__ double sin_small(double x) {
    using namespace sincos_detail;
    if (!(fabs(x) <= 64.0)) return sin(x);
    double y, t;
    const int n = reduce(x, y, t) & 3;
    if (fabs(y) < kSmallRem) reduce_accurate(x, y, t);
    const double s = k_sin(y, t), c = k_cos(y, t);
    const double r = (n & 1) ? c : s;
    return (n & 2) ? -r : r;
}
__device__ __forceinline__ double cos_small(double x) {
    using namespace sincos_detail;
    if (!(fabs(x) <= 64.0)) return cos(x);
    double y, t;
    const int n = reduce(x, y, t) & 3;
    if (fabs(y) < kSmallRem) reduce_accurate(x, y, t);
    const double s = k_sin(y, t), c = k_cos(y, t);
    const double r = (n & 1) ? s : c;
    return ((n + 1) & 2) ? -r : r;
}

// ---- the table form (round 2): sin / cos of a + b, a = k*pi/256 from a table, |b| <= pi/512 ----------------------------------------
// The oscillators call this one.  x = k*h + b with h = pi/256 held in three pieces (39 + 39 + 53 bits: k*P1 and k*P2 are exact for
// k < 2^13, i.e. |x| <= 64, so b = (bh, bl) is good to 2^-120 absolute -- relative 2^-60 even for the double nearest a zero of
// the function, with no special case); sin(a), cos(a) come as double-double (S, C) from a 512-entry table in LDS; sin b - b and
// cos b - 1 are three-term polynomials (|b| < 0.0062: the next terms are below 2^-75 of the result).  Then
//     sin(a + b) = S + (S (cos b - 1) + C sin b),     cos(a + b) = C + (C (cos b - 1) - S sin b),
// with the one term that can be as large as the leading one, C*bh (resp. S*bh), kept exact (product + FMA remainder) and added to
// the leading term with its rounding error recovered (Fast2Sum: |S_hi| >= sin(pi/256) > |C*bh| whenever S_hi is not exactly 0).
// ~32 instructions against ~50 for the reduction + both fdlibm kernels + quadrant select above, and MORE accurate: measured
// max error 0.5000 ULP against quad precision on 6 M arguments (whole domain, zero crossings and their neighbours, table
// points and their neighbours, tiny arguments: tests/host_sincos_accuracy.cpp) -- i.e. correctly rounded on every one of them.
// tab = MXG_SINTAB (mxg_sintab.h) in LDS / memory.  |x| > 64, NaN, Inf: the routines above.
#include "mxg_sintab.h"
#if defined(__HIPCC__)
#define MXG_NO_CONTRACT  // the library is built with -ffp-contract=off: only the fma() calls below fuse
#else
#define MXG_NO_CONTRACT __attribute__((optimize("fp-contract=off")))  // host builds of this text (tests): the same
#endif
// The two polynomial coefficients that enter an FMA next to another constant (z*k1 + k0: gfx950 takes ONE scalar operand per VALU
// instruction, so k0 sits in a vector register).  A caller with a loop passes them in, loaded once and made opaque to the optimizer --
// which otherwise re-creates each from its literal with a v_mov_b64 per sample (round 4: 1 of sinewave's 43 instructions).
struct SinTabK {
    double s120, c24;
};
__device__ __forceinline__ SinTabK sintab_k() {
    SinTabK k = {1.0 / 120, 1.0 / 24};
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(k.s120), "+v"(k.c24));
#endif
    return k;
}
// TRUST: the caller guarantees 0 <= x <= 4 pi (an oscillator whose phase stays in [0, 2]) AND a table of MXG_SINTAB_WIDE_LEN doubles
// (entries 0 .. 1024: the period and its repetition): no range test, no sign to restore, no index wrap.
#define MXG_SINTAB_WIDE_LEN (4 * 1025)
template <bool COS, bool TRUST = false>
MXG_NO_CONTRACT __device__ __forceinline__ double sincos_tab(double x, const double *tab, const SinTabK k = {1.0 / 120, 1.0 / 24}) {
    if constexpr (!TRUST)
        if (!(fabs(x) <= 64.0)) return COS ? cos(x) : sin(x);
    const double ax = TRUST ? x : fabs(x);
    const double fk = rint(ax * MXG_SINTAB_INVH);
    const double r1 = fma(-fk, MXG_SINTAB_P1, ax);  // exact
    const double w = fk * MXG_SINTAB_P2;            // exact
    const double bh = r1 - w;
    const double bl = fma(-fk, MXG_SINTAB_P3, (r1 - bh) - w);
    const double *e = tab + 4 * (TRUST ? (int)fk : ((int)fk & 511));
    const double Sh = e[0], Sl = e[1], Ch = e[2], Cl = e[3];
    const double z = bh * bh;
    using sincos_detail::fma_k;
    using sincos_detail::fma_kk;
    const double sc = fma(bh * z, fma_k(z, fma_kk(z, -1.0 / 5040, k.s120), -1.0 / 6), bl);  // sin b - bh
    const double cm1 = z * fma_k(z, fma_kk(z, -1.0 / 720, k.c24), -0.5);                    // cos b - 1
    const double Ah = COS ? Ch : Sh, Al = COS ? Cl : Sl, Bh = COS ? -Sh : Ch, Bl = COS ? -Sl : Cl;
    double t = fma(Bh, sc, Ah * cm1);
    t = t + fma(Bl, bh, Al);
    const double ph = Bh * bh, pl = fma(Bh, bh, -ph);
    const double s = Ah + ph, err = ph - (s - Ah);
    const double r = s + (err + (pl + t));
    if constexpr (TRUST) return r;
    return (!COS && x < 0) ? -r : r;
}

template <bool TRUST = false>
__device__ __forceinline__ double sin_2pi_phase(double phase, const double *sintab, const SinTabK k = {1.0 / 120, 1.0 / 24}) {
    double x = phase * (MXG_TWOPI);
    return sincos_tab<false, TRUST>(x, sintab, k);
}
template <bool TRUST = false>
__device__ __forceinline__ double cos_2pi_phase(double phase, const double *sintab, const SinTabK k = {1.0 / 120, 1.0 / 24}) {
    double x = phase * (MXG_TWOPI);
    return sincos_tab<true, TRUST>(x, sintab, k);
}

}  // namespace mxg
