// mxg_advance.h -- exact multi-step forms of the granular scheduler's recurrences (used by grains.hip's K8a).
// Plain arithmetic on doubles and 64-bit integers, no device state: the same text compiles for the host, which is how
// tests/test_sched_host.py fuzzes it against the step-by-step recurrences (tests/host_sched_fuzz.cpp).
#pragma once
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define MXG_HD __device__ __forceinline__
#define MXG_HD_NOINLINE __device__
#else
#define MXG_HD static inline
#define MXG_HD_NOINLINE static
static inline long long __double_as_longlong(double x) { long long b; memcpy(&b, &x, 8); return b; }
static inline double __longlong_as_double(long long b) { double x; memcpy(&x, &b, 8); return x; }
#endif

namespace mxg {
namespace {

// ---- exact multi-step advance of x <- fl(x + r) --------------------------------------------------
// The scheduler's two recurrences (position += rate, looper += 1.0) are plain repeated additions
// between rare events (a wrap, a spawn).  They can be advanced by many steps at once WITHOUT
// changing a single bit of the result, because IEEE addition is predictable in two situations:
//  (a) "grid" case: x and r are both multiples of g = 2^q and every partial sum stays below
//      g*2^53  =>  every addition is exact, x_k = x + k*r  (looper += 1.0 always lands here; so
//      do rates with short mantissas such as 0.265625);
//  (b) "binade" case: while the sums stay inside x's binade [2^e, 2^(e+1)) they live on the grid
//      u = 2^(e-52); r = R*u + f, and round-to-nearest turns each step into +R*u (f < u/2) or
//      +(R+1)*u (f > u/2), a CONSTANT integer number of ulps  =>  X_k = X + k*c on the mantissa.
//      (f == u/2, the tie, is resolved by parity one step at a time.)
// Anything else (binade crossings, the step that crosses the limit, x = 0, subnormals) is done
// as one ordinary floating-point step, so the sequence is the reference's, bit for bit.
// Requires x >= 0, r > 0.  Advances at most kmax steps and stops right after the first step
// whose result crosses the limit (x > limit, or x >= limit if `inclusive`); returns steps taken.
MXG_HD int lowbit_exp(double v) {  // exponent of the lowest set bit of v (v > 0, normal)
    const long long b = __double_as_longlong(v);
    const int e = (int)((b >> 52) & 0x7ff);
    const unsigned long long m = ((unsigned long long)b & 0xFFFFFFFFFFFFFULL) | (1ULL << 52);
    return e - 1075 + (int)__builtin_ctzll(m);
}

// floor(num / den) for 0 <= num < 2^53, 0 < den < 2^53: a double division (both operands exact) is off by at most
// one, which two multiply-compare fix-ups repair -- a fraction of the cost of the emulated 64-bit integer division.
MXG_HD long long floor_div53(long long num, long long den) {
    long long q = (long long)((double)num / (double)den);
    while (q * den > num) q--;
    while ((q + 1) * den <= num) q++;
    return q;
}

MXG_HD_NOINLINE int advance_until(double &x, const double r, const double limit, const bool inclusive, int kmax,
                             bool &crossed) {
    crossed = false;
    int done = 0;
    while (done < kmax) {
        bool jumped = false;
        const long long xb = __double_as_longlong(x);
        const int e = (int)((xb >> 52) & 0x7ff);
        const int er = (int)((__double_as_longlong(r) >> 52) & 0x7ff);
        if (er > 0 && er < 0x7ff && e < 0x7ff && (e > 0 || x == 0.0)) {
            // ---- (a) grid case
            int q = lowbit_exp(r);
            if (x != 0.0) {
                const int qx = lowbit_exp(x);
                q = qx < q ? qx : q;
            }
            // all partial sums up to the one that crosses are <= limit + r: exact if that is < 2^(q+53)
            const double top = ldexp(1.0, q + 53);
            if (limit + r < top && x < top) {
                const double Xs = ldexp(x, -q), Rs = ldexp(r, -q), Ls = ldexp(limit, -q);  // exact scalings
                // largest k with x + k*r NOT crossed: X + k*R <= L (exclusive) / < L (inclusive)
                double kk = floor((Ls - Xs) / Rs);  // estimate, fixed up exactly below
                if (kk < 0) kk = 0;
                if (kk > (double)(kmax - done)) kk = (double)(kmax - done);
                long long k = (long long)kk;
                // integers below 2^53: the products/sums here are exact in double
                auto not_crossed = [&](long long j) {
                    const double v = Xs + (double)j * Rs;
                    return inclusive ? (v < Ls) : (v <= Ls);
                };
                while (k > 0 && !not_crossed(k)) k--;
                while (k < (long long)(kmax - done) && not_crossed(k + 1)) k++;
                if (k >= 1) {
                    x = ldexp(Xs + (double)k * Rs, q);
                    done += (int)k;
                    jumped = true;
                }
            } else if (e > 0) {
                // ---- (b) binade case
                const double ru = ldexp(r, 1075 - e);  // r in ulps of x (exact scaling)
                if (ru < 4503599627370496.0) {         // < 2^52, else the sum leaves the binade at once
                    const double Rf = floor(ru), fr = ru - Rf;
                    const long long X = (xb & 0xFFFFFFFFFFFFFLL) | (1LL << 52);
                    const long long R = (long long)Rf;
                    const bool tie = fr == 0.5;
                    const long long c = fr < 0.5 ? R : (fr > 0.5 ? R + 1 : (((X + R) & 1) ? R + 1 : R));
                    long long j = (long long)(kmax - done);
                    if (c > 0) {
                        const long long jb = floor_div53((1LL << 53) - 1 - X, c);  // stay inside the binade
                        j = jb < j ? jb : j;
                        const double lu = ldexp(limit, 1075 - e);
                        if (lu < 9007199254740992.0) {  // limit inside/below this binade's mantissa range
                            const long long Lq = inclusive ? (long long)ceil(lu) - 1 : (long long)floor(lu);
                            const long long jl = Lq >= X ? floor_div53(Lq - X, c) : 0;  // steps that do not cross
                            j = jl < j ? jl : j;
                        }
                    } else if (inclusive ? x >= limit : x > limit) {
                        j = 0;  // x is stuck (r < ulp/2) on a value that already passes the test: the next step crosses
                    }
                    if (tie && j > 1) j = 1;
                    if (j >= 1) {
                        const long long Xn = X + j * c;
                        x = __longlong_as_double(((long long)e << 52) | (Xn & 0xFFFFFFFFFFFFFLL));
                        done += (int)j;
                        jumped = true;
                    }
                }
            }
        }
        if (!jumped) {  // one ordinary step
            x = x + r;
            done++;
            if (inclusive ? x >= limit : x > limit) {
                crossed = true;
                return done;
            }
        }
    }
    return done;
}

// ---- the next n additions of x <- fl(x + r) as a LINE on x's mantissa grid -----------------------------------------
// The binade case of advance_until stated for a fixed number of steps, so that step k can be evaluated on its own (a lane
// per step): inside x's binade [2^e, 2^(e+1)) every partial sum is an integer multiple of u = 2^(e-52); r = (R + f) u with
// 0 <= f < 1, and round-to-nearest adds R ulps (f < 1/2) or R + 1 ulps (f > 1/2) on EVERY step:  X_k = X + k c,
// x_k = X_k * u.  ok = false when that does not hold for all k <= n: r <= 0 or not normal, x not a positive normal number,
// r >= 2^52 ulps, the tie f == 1/2 (then the parity of X decides step by step), a sum leaving the binade, or a sum
// reaching `limit` (the caller's wrap).  The value after k steps is add_line_at(l, k).
struct AddLine {
    long long X, c;
    int e;
    bool ok;
};
MXG_HD AddLine add_line(const double x, const double r, const double limit, const int n) {
    AddLine l = {0, 0, 0, false};
    const long long xb = __double_as_longlong(x), rb = __double_as_longlong(r);
    const int e = (int)((xb >> 52) & 0x7ff), er = (int)((rb >> 52) & 0x7ff);
    if (xb < 0 || rb <= 0 || e == 0 || e == 0x7ff || er == 0 || er == 0x7ff) return l;  // signs, zero, subnormal, Inf, NaN
    const double ru = ldexp(r, 1075 - e);  // r in ulps of x (exact scaling; may underflow to 0 or a subnormal: f then < 1/2)
    if (!(ru < 4503599627370496.0)) return l;
    const double Rf = floor(ru), fr = ru - Rf;
    if (fr == 0.5) return l;
    l.X = (xb & 0xFFFFFFFFFFFFFLL) | (1LL << 52);
    l.c = (long long)Rf + (fr > 0.5 ? 1 : 0);
    l.e = e;
    const long long Xn = l.X + (long long)n * l.c;  // < 2^53 + 2^7 * 2^52: no overflow for n <= 128
    if (n < 0 || n > 128 || Xn > (1LL << 53) - 1) return l;
    const double last = __longlong_as_double(((long long)e << 52) | (Xn & 0xFFFFFFFFFFFFFLL));
    l.ok = last < limit;
    return l;
}
MXG_HD double add_line_at(const AddLine &l, const int k) {
    const long long Xk = l.X + (long long)k * l.c;
    return __longlong_as_double(((long long)l.e << 52) | (Xk & 0xFFFFFFFFFFFFFLL));
}

// Smallest integer Lc > L (L an integer-valued double) with floor(fmod(Lc, cyc)) == 0, cyc > 2.  The candidate comes
// from one multiplication; fmod is exact, so the predicate itself confirms it (and that the integer before it is not
// a birth, and that no whole cycle was jumped).  ok = false: could not be confirmed, the caller walks sample by sample.
MXG_HD double next_birth(double L, double cyc, bool &ok) {
    auto born = [&](double x) { return 0 == floor(fmod(x, cyc)); };
    if (born(L + 1.0)) return L + 1.0;
    const double m = floor((L + 1.0) / cyc) + 1.0;
    double Lc = ceil(m * cyc);
    if (Lc <= L + 1.0) Lc = ceil((m + 1.0) * cyc);  // the quotient was rounded down across an integer
    if (Lc - 1.0 > L + 1.0 && born(Lc - 1.0)) Lc -= 1.0;
    else if (!born(Lc)) Lc += 1.0;
    if (!born(Lc) || (Lc - 1.0 > L && born(Lc - 1.0)) || Lc - L > cyc + 1.0) ok = false;
    return Lc;
}

}  // namespace
}  // namespace mxg
