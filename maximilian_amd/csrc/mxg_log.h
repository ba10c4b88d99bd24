// mxg_log.h -- natural logarithm for the mel band energies, log(mb * mb) of L/maxiMFCC.cpp:63.
//
// The reference calls glibc log() (within 1 ULP of the true value).  The device's generic log() is as accurate but costs
// about 95 instructions, which made the 42 logs of a frame a sixth of the fused FFT+MFCC kernel's instruction stream.  The
// band energies that reach the log are finite, positive and normal (the reference's own guard: mb > 1e-6), so the fast path
// is the classic reduction x = 2^k (1 + f), sqrt(1/2) < 1 + f < sqrt(2), log(1 + f) = 2s + s R(s^2) with s = f / (2 + f) and
// the degree-14 minimax polynomial published with Sun's fdlibm (e_log.c; error of the polynomial < 2^-58.45), evaluated with
// FMAs -- libm-internal arithmetic, not one of the reference's expression trees.  The quotient is formed from the
// low-precision hardware reciprocal by two Newton steps and one residual correction (<= 1 ULP).  About 40 instructions,
// error < 1 ULP (tests/host_log_accuracy.cpp measures it against quad precision: 0.8 ULP on 16 M arguments); the tolerance
// of the mfcc outputs (1e-12 of the largest band) is unchanged.  Zero, negative, subnormal, Inf and NaN
// arguments go to the generic log().
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#else  // host build of the same text (accuracy test on the CPU)
#include <math.h>
#include <stdint.h>
#include <string.h>
#define __device__
#define __forceinline__ inline
#endif

namespace mxg {

namespace log_detail {
constexpr double kLn2Hi = 6.93147180369123816490e-01, kLn2Lo = 1.90821492927058770002e-10;  // fdlibm e_log.c
constexpr double kLg1 = 6.666666666666735130e-01, kLg2 = 3.999999999940941908e-01, kLg3 = 2.857142874366239149e-01,
                 kLg4 = 2.222219843214978396e-01, kLg5 = 1.818357216161805012e-01, kLg6 = 1.531383769920937332e-01,
                 kLg7 = 1.479819860511658591e-01;
constexpr double kSqrtHalf = 7.07106781186547524401e-01;

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1 / d for d in (1.7, 3.5): the hardware seed is good to ~24 bits; two Newton steps square the error twice
__device__ __forceinline__ double recip(double d) {
#if defined(__HIPCC__)
    double r = __builtin_amdgcn_rcp(d);
#else
    double r = (double)(1.0f / (float)d);  // a seed of the same quality
#endif
    r = fma_(fma_(-d, r, 1.0), r, r);
    r = fma_(fma_(-d, r, 1.0), r, r);
    return r;
}
}  // namespace log_detail

// log(x) for finite normal x > 0 (anything else: the generic routine)
__device__ __forceinline__ double fast_log(double x) {
    using namespace log_detail;
    uint64_t bits;
#if defined(__HIPCC__)
    bits = (uint64_t)__double_as_longlong(x);
#else
    memcpy(&bits, &x, 8);
#endif
    const uint32_t hi = (uint32_t)(bits >> 32);
    if (__builtin_expect(hi - 0x00100000u >= 0x7ff00000u - 0x00100000u, 0)) return log(x);  // zero, subnormal, negative, Inf, NaN
    // x = 2^k * m, m in [sqrt(1/2), sqrt(2)): fdlibm's exponent trick on the high word
    int k = (int)(hi >> 20) - 1023;
    uint32_t mh = hi & 0x000fffffu;
    const uint32_t up = (mh + 0x95f64u) & 0x100000u;  // mantissa >= sqrt(2): halve it
    mh |= up ^ 0x3ff00000u;
    k += (int)(up >> 20);
    const uint64_t mbits = ((uint64_t)mh << 32) | (bits & 0xffffffffu);
    double m;
#if defined(__HIPCC__)
    m = __longlong_as_double((long long)mbits);
#else
    memcpy(&m, &mbits, 8);
#endif
    const double f = m - 1.0;
    const double d = 2.0 + f;
    const double r = recip(d);
    double s = f * r;
    s = fma_(fma_(-d, s, f), r, s);  // residual correction: s = f / d to within an ulp
    const double z = s * s, w = z * z;
    const double t1 = w * fma_(w, fma_(w, kLg6, kLg4), kLg2);
    const double t2 = z * fma_(w, fma_(w, fma_(w, kLg7, kLg5), kLg3), kLg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * kLn2Hi - ((hfsq - fma_(dk, kLn2Lo, s * (hfsq + R))) - f);
}

}  // namespace mxg
