// mxg_sincos.h -- sin/cos of (phase * TWOPI) for maxiOsc::sinewave / coswave (C:230, C:278).
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
// libm-internal arithmetic, not one of the reference's expression trees.  Error < 0.85 ULP of the result (+ 2^-62
// absolute next to a zero of the function), measured on 8 M arguments by tests/host_sincos_accuracy.cpp (0.77): below
// 1 ULP, so the result and glibc's are the two doubles bracketing the true value -- at most 1 ULP apart.
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
constexpr double kPio2Lo2 = 2.02226624879595063154e-21;  // pi/2 - kPio2Hi - kPio2Lo (third piece)
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

// sin and cos of y + t on [-pi/4, pi/4] (t = tail of the reduced argument)
__device__ __forceinline__ double k_sin(double y, double t) {
#pragma clang fp contract(fast)
    const double z = y * y, v = z * y;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return y - ((z * (0.5 * t - v * r) - t) - v * S1);
}
__device__ __forceinline__ double k_cos(double y, double t) {
#pragma clang fp contract(fast)
    const double z = y * y;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
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
}  // namespace sincos_detail

__device__ __forceinline__ double sin_small(double x) {
    using namespace sincos_detail;
    if (!(fabs(x) <= 64.0)) return sin(x);
    double y, t;
    const int n = reduce(x, y, t) & 3;
    const double s = k_sin(y, t), c = k_cos(y, t);
    const double r = (n & 1) ? c : s;
    return (n & 2) ? -r : r;
}
__device__ __forceinline__ double cos_small(double x) {
    using namespace sincos_detail;
    if (!(fabs(x) <= 64.0)) return cos(x);
    double y, t;
    const int n = reduce(x, y, t) & 3;
    const double s = k_sin(y, t), c = k_cos(y, t);
    const double r = (n & 1) ? s : c;
    return ((n + 1) & 2) ? -r : r;
}

__device__ __forceinline__ double sin_2pi_phase(double phase) {
    double x = phase * (MXG_TWOPI);
    return sin_small(x);
}
__device__ __forceinline__ double cos_2pi_phase(double phase) {
    double x = phase * (MXG_TWOPI);
    return cos_small(x);
}

}  // namespace mxg
