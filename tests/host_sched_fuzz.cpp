// tests/host_sched_fuzz.cpp -- host fuzz of maximilian_amd/csrc/mxg_advance.h (the exact multi-step forms the granular
// scheduler K8a uses) against the step-by-step recurrences of the reference (L/maxiGrains.h:341-355, 359-367, 412-430:
// `position += rate`, `looper++`, `floor(fmod(looper, cycle)) == 0`).  Built and run by tests/test_sched_host.py.
#include <stdint.h>
#include <stdio.h>
#include <random>

#include "mxg_advance.h"

using namespace mxg;

static int naive_advance(double &x, double r, double limit, bool inclusive, int kmax, bool &crossed) {
    crossed = false;
    int done = 0;
    while (done < kmax) {
        x = x + r;
        done++;
        if (inclusive ? x >= limit : x > limit) {
            crossed = true;
            break;
        }
    }
    return done;
}

static uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 g(0x4D415849);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    long bad = 0, jumps = 0;
    for (long i = 0; i < cases; i++) {
        // rates: full 53-bit mantissas, short mantissas, tiny, huge; starts: 0, small, near the limit, fractional
        double r;
        switch (g() % 6) {
            case 0: r = 0.25 + 1.5 * u01(g); break;
            case 1: r = (double)(1 + g() % 512) / 256.0; break;          // multiples of 2^-8
            case 2: r = ldexp(1.0 + u01(g), -(int)(g() % 40)); break;     // down to 2^-40
            case 3: r = 1.0; break;
            case 4: r = 1.0 + (double)(g() % 5000) + u01(g); break;       // large steps
            default: r = u01(g) * 1e-3 + 1e-9; break;
        }
        double limit;
        switch (g() % 4) {
            case 0: limit = 44100.0 * (1 + g() % 100); break;
            case 1: limit = 551.25 + (double)(g() % 10); break;
            case 2: limit = 1.0 + u01(g) * 1e6; break;
            default: limit = (double)(1 + g() % 5000); break;
        }
        double x0;
        switch (g() % 5) {
            case 0: x0 = 0.0; break;
            case 1: x0 = limit * u01(g); break;
            case 2: x0 = limit - r * (double)(g() % 8) * u01(g); if (x0 < 0) x0 = 0; break;
            case 3: x0 = (double)(g() % 4096); break;
            default: x0 = limit; break;
        }
        const bool inclusive = g() & 1;
        const int kmax = 1 + (int)(g() % 3000);
        double xa = x0, xb = x0;
        bool ca, cb;
        const int da = advance_until(xa, r, limit, inclusive, kmax, ca);
        const int db = naive_advance(xb, r, limit, inclusive, kmax, cb);
        if (da != db || ca != cb || bits(xa) != bits(xb)) {
            if (bad < 5)
                printf("advance_until mismatch: x0=%a r=%a limit=%a incl=%d kmax=%d -> (%d,%d,%a) vs (%d,%d,%a)\n", x0,
                       r, limit, (int)inclusive, kmax, da, (int)ca, xa, db, (int)cb, xb);
            bad++;
        }
        jumps += da;
    }
    // add_line: every one of the next n sums, on its own, against the recurrence
    long line_bad = 0, line_ok = 0;
    for (long i = 0; i < cases; i++) {
        double r;
        switch (g() % 6) {
            case 0: r = 0.25 + 1.5 * u01(g); break;
            case 1: r = (double)(1 + g() % 512) / 256.0; break;
            case 2: r = ldexp(1.0 + u01(g), -(int)(g() % 70)); break;     // far below an ulp of x too
            case 3: r = 1.0; break;
            case 4: r = ldexp(1.0 + (double)(g() % 3) * 0.5, -(int)(g() % 12)) ; break;  // exact halves of an ulp: ties
            default: r = u01(g) * 1e-3 + 1e-9; break;
        }
        double x0;
        switch (g() % 5) {
            case 0: x0 = 1.0 + u01(g) * 4.4e6; break;
            case 1: x0 = ldexp(1.0, (int)(g() % 24)) - r * (double)(g() % 70); if (!(x0 > 0)) x0 = 1.0; break;  // about to leave its binade
            case 2: x0 = (double)(1 + g() % 100000); break;
            case 3: x0 = u01(g); break;
            default: x0 = ldexp(1.0 + u01(g), (int)(g() % 22)); break;
        }
        const double limit = (g() & 1) ? 4.41e6 : x0 + r * (double)(g() % 200);
        const int n = 1 + (int)(g() % 128);
        const AddLine l = add_line(x0, r, limit, n);
        if (!l.ok) continue;
        line_ok++;
        double x = x0;
        for (int k = 1; k <= n; k++) {
            x = x + r;
            if (bits(add_line_at(l, k)) != bits(x) || !(x < limit)) {
                if (line_bad < 5) printf("add_line mismatch: x0=%a r=%a n=%d step %d: %a vs %a (limit %a)\n", x0, r, n, k,
                                         add_line_at(l, k), x, limit);
                line_bad++;
                break;
            }
        }
    }
    printf("add_line: %ld of %ld cases accepted, %ld wrong\n", line_ok, cases, line_bad);
    bad += line_bad;
    if (line_ok < cases / 4) bad++;  // the fuzz is meant to exercise the accepted case
    long nb_bad = 0, nb_fallback = 0;
    for (long i = 0; i < cases / 4; i++) {
        double cyc;
        switch (g() % 4) {
            case 0: cyc = 551.25; break;
            case 1: cyc = 2.0 + u01(g) * 3000.0; break;
            case 2: cyc = (double)(3 + g() % 2000); break;               // integer cycles
            default: cyc = 551.25 + (double)(g() % 10); break;           // cycleLength + rand()%10
        }
        const double L = (double)(g() % 2000000);
        bool ok = true;
        const double Lc = next_birth(L, cyc, ok);
        if (!ok) { nb_fallback++; continue; }
        double Ln = L + 1.0;  // the reference's walk: one sample at a time
        while (!(0 == floor(fmod(Ln, cyc)))) Ln += 1.0;
        if (Ln != Lc) {
            if (nb_bad < 5) printf("next_birth mismatch: L=%.0f cyc=%a -> %.0f vs %.0f\n", L, cyc, Lc, Ln);
            nb_bad++;
        }
    }
    printf("advance_until: %ld cases, %ld mismatches, %ld steps; next_birth: %ld cases, %ld mismatches, %ld fallbacks\n",
           cases, bad, jumps, cases / 4, nb_bad, nb_fallback);
    return (bad || nb_bad) ? 1 : 0;
}
