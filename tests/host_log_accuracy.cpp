// tests/host_log_accuracy.cpp -- error bound of maximilian_amd/csrc/mxg_log.h (the log behind maxiMFCC's log(mb*mb),
// L/maxiMFCC.cpp:63, on the device), measured on the host against quad-precision logq (libquadmath) and against glibc's
// double log (what the reference calls).  Run by tests/test_sincos_host.py.
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>

#include "mxg_log.h"

typedef __float128 quad;
static double ulp_of(quad v) {
    int e;
    frexpq(fabsq(v), &e);
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

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 g(0x4D415849);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    double worst = 0, arg = 0;
    long long worst_vs_libm = 0;
    for (long i = 0; i < cases; i++) {
        double x;
        switch (i % 5) {
            case 0: x = ldexp(0.5 + 0.5 * u(g), (int)(g() % 120) - 60); break;   // band energies: 1e-18 .. 1e18
            case 1: x = 1.0 + ldexp(u(g) - 0.5, -(int)(g() % 52)); break;         // closing in on 1 (log -> 0, cancellation)
            case 2: x = ldexp(0.70710678118654752 + 1e-6 * (u(g) - 0.5), (int)(g() % 40) - 20); break;  // the sqrt(1/2) seam
            case 3: { double mb = 1e-6 + u(g) * 50.0; x = mb * mb; break; }      // exactly what log_square feeds it
            default: x = ldexp(0.5 + 0.5 * u(g), (int)(g() % 2040) - 1020); break;  // every binade of the normal range
        }
        const double r = mxg::fast_log(x);
        const quad t = logq((quad)x);
        const double e = t == 0 ? (r == 0 ? 0.0 : 1e9) : (double)(fabsq((quad)r - t) / ulp_of(t));
        if (e > worst) { worst = e; arg = x; }
        const long long d = bits_diff(r, log(x));
        if (d > worst_vs_libm) worst_vs_libm = d;
    }
    // the slow path's arguments return what log() returns
    const double special[] = {0.0, -1.0, 4.9e-324, 2.2e-308 / 4, INFINITY, NAN};
    int ok = 1;
    for (double s : special) {
        const double a = mxg::fast_log(s), b = log(s);
        if (!((isnan(a) && isnan(b)) || a == b)) ok = 0;
    }
    printf("log: max error %.4f ULP at x=%a; vs glibc log: max %lld ULP (%ld cases); specials %s\n", worst, arg, worst_vs_libm,
           cases, ok ? "ok" : "WRONG");
    return (worst < 0.9 && worst_vs_libm <= 1 && ok) ? 0 : 1;
}
