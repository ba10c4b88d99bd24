// tests/host_sincos_accuracy.cpp -- error bound of maximilian_amd/csrc/mxg_sincos.h (the sin/cos behind
// maxiOsc::sinewave / coswave and the modulated lores/bandpass coefficients on the device), measured on the host against
// quad-precision sinq/cosq (libquadmath) and against glibc's double sin/cos (what the reference calls).  Built with FMA contraction on
// (-mfma -ffp-contract=fast), the closest host equivalent of the device's `#pragma clang fp contract(fast)` kernels.
// Run by tests/test_sincos_host.py.
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>

#include "mxg_sincos.h"

typedef __float128 quad;
static double ulp_of(quad v) {
    int e;
    frexpq(fabsq(v), &e);          // |v| = m * 2^e, m in [0.5, 1)
    if (e < -1021) e = -1021;
    return ldexp(1.0, e - 53);
}
static long long bits_diff(double a, double b) {
    long long x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    if (x < 0) x = (long long)0x8000000000000000ULL - x;
    if (y < 0) y = (long long)0x8000000000000000ULL - y;
    return llabs(x - y);
}

static const double kSinTab[MXG_SINTAB_LEN] = {MXG_SINTAB_VALUES};

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 2000000;
    double worst_ts = 0, worst_tc = 0, arg_ts = 0, arg_tc = 0;
    std::mt19937_64 g(0x4D415849);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    double worst_s = 0, worst_c = 0, arg_s = 0, arg_c = 0;
    long long worst_vs_libm = 0;
    for (long i = 0; i < cases; i++) {
        double x;
        switch (i % 7) {
            case 6: x = (double)(long)(5000.0 * fabs(u(g))) * 0.01227184630308513 + ldexp(u(g), -(int)(8 + g() % 45)); break;  // the table's points and their neighbours
            case 0: x = 64.0 * u(g); break;                                // the whole fast-path domain
            case 1: x = (0.5 + 0.5 * u(g)) * MXG_TWOPI; break;               // an oscillator's phase * TWOPI
            case 2: x = (double)(long)(40.0 * u(g)) * 1.5707963267948966 + 1e-3 * u(g); break;  // near multiples of pi/2
            case 3: {  // the doubles AT and right next to fl(k*pi/2): the zero crossings themselves (phase 0.25, 0.5 ...)
                x = (double)(long)(40.0 * u(g)) * 1.5707963267948966;
                const int steps = (int)(g() % 9) - 4;
                for (int k = 0; k < (steps < 0 ? -steps : steps); k++) x = nextafter(x, steps < 0 ? -1e9 : 1e9);
                break;
            }
            case 4: x = (double)(long)(40.0 * u(g)) * 1.5707963267948966 + ldexp(u(g), -(int)(10 + g() % 40)); break;  // closing in on them
            default: x = ldexp(u(g), -(int)(g() % 60)); break;              // tiny arguments
        }
        const double s = mxg::sin_small(x), c = mxg::cos_small(x);
        const quad ts = sinq((quad)x), tc = cosq((quad)x);
        // Contract: |result - f(x)| < 0.85 ulp(f(x)) everywhere, zero crossings included.  Below 1 ULP is what matters:
        // the result and glibc's (itself within 1 ULP of f) are then the two doubles that bracket f(x), at most 1 ULP apart.
        const double es = (double)(fabsq((quad)s - ts) / ulp_of(ts));
        const double ec = (double)(fabsq((quad)c - tc) / ulp_of(tc));
        if (es > worst_s) { worst_s = es; arg_s = x; }
        if (ec > worst_c) { worst_c = ec; arg_c = x; }
        const long long d1 = bits_diff(s, sin(x)), d2 = bits_diff(c, cos(x));
        if (d1 > worst_vs_libm) worst_vs_libm = d1;
        if (d2 > worst_vs_libm) worst_vs_libm = d2;
        // the table form the oscillators use (sincos_tab): same contract, and it is held to correct rounding + a hair
        const double st = mxg::sincos_tab<false>(x, kSinTab), ct = mxg::sincos_tab<true>(x, kSinTab);
        const double ets = (double)(fabsq((quad)st - ts) / ulp_of(ts)), etc = (double)(fabsq((quad)ct - tc) / ulp_of(tc));
        if (ets > worst_ts) { worst_ts = ets; arg_ts = x; }
        if (etc > worst_tc) { worst_tc = etc; arg_tc = x; }
        const long long d3 = bits_diff(st, sin(x)), d4 = bits_diff(ct, cos(x));
        if (d3 > worst_vs_libm) worst_vs_libm = d3;
        if (d4 > worst_vs_libm) worst_vs_libm = d4;
    }
    printf("table form: sin max error %.4f ULP at x=%a; cos %.4f ULP at x=%a\n", worst_ts, arg_ts, worst_tc, arg_tc);
    if (!(worst_ts < 0.51 && worst_tc < 0.51)) return 1;
    printf("sin: max error %.4f ULP at x=%a; cos: %.4f ULP at x=%a; vs glibc sin/cos: max %lld ULP (%ld cases)\n",
           worst_s, arg_s, worst_c, arg_c, worst_vs_libm, cases);
    return (worst_s < 0.85 && worst_c < 0.85 && worst_vs_libm <= 1) ? 0 : 1;
}
