// mxg_smp.h -- the maxiSample players (src/maximilian.cpp:740-1075) split into head-advance / gather-index
// generation (smp_gen) and interpolation (smp_eval); shared by sample.hip and sampler.hip.
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#else  // host build of the same text (tests/host_smp.cpp: the players against the oracle over wide parameter ranges)
#include <math.h>
#include <stddef.h>
#define __host__
#define __device__
#define __forceinline__ inline
namespace mxg {
constexpr double kChandiv = 1.0;
}
#endif

namespace mxg {
namespace {

// Layout of an uploaded sample (mxg_sample_upload / mxg_sample_load_wav): kSmpGuardLo zeros, the len samples, kSmpGuardHi
// zeros; the pointer handed around is element 0.  [-1], [len] and [len+1] are what the reference's players read next to the
// buffer inside the ranges it is defined on (C:898, C:1063-1064; zero is the parity convention).  The remaining guards
// back the index clamp below: outside those ranges (a play4 step longer than its loop, playLoop with end > 1, a head
// uploaded far outside the sample) the reference indexes outside its vector -- undefined -- and the device reads a
// guard zero instead of whatever lies, or does not lie, beyond the allocation.
// kSmpWindow: sample.hip's time-part kernel fetches the indices of a chunk as one window of that many consecutive doubles per
// voice, starting at the chunk's lowest index (<= len+4): the high guard covers a window that starts on the last index.
constexpr int kSmpWindow = 16;
constexpr int kSmpGuardLo = 4, kSmpGuardHi = 6 + kSmpWindow;

// The head as the index computations see it: unchanged on [-2, len+2] (every position the reference is defined on),
// pinned to that interval otherwise (NaN -> -2), so derived indices stay within [-4, len+4].
__device__ __forceinline__ double smp_safe_head(double pos, size_t len) {
    return fmin(fmax(pos, -2.0), (double)len + 2.0);
}

struct Smp {
    const double *amp;
    size_t len;
    double pos;
    double step_div;  // (double)(sampleRate / mySampleRate), the INTEGER quotient of C:1070
    // trigger-driven modes (9-14): maxiTrigger::previousValue/firstTrigger (H:593-594), or
    // phasorPrev/phasorFirst (H:731-732) for playWithPhasor; p0/p1 = the per-voice offset/length/pos
    double tprev;
    bool tfirst;
    double p0, p1;
};

// Modes 9-13 are a maxiTrigger::onZX test (H:569-579) in front of one of the plain players.
__host__ __device__ constexpr int smp_base(int mode) {
    return mode == 9 ? 1 : (mode == 10 || mode == 11) ? 5 : mode == 12 ? 6 : mode == 13 ? 0 : mode;
}
__host__ __device__ constexpr int smp_loads(int mode) {
    return mode == 14 ? 2 : (smp_base(mode) <= 3 ? 1 : (smp_base(mode) == 7 ? 4 : 2));
}

// ---- pipelined form ------------------------------------------------------------------------
// Every player is split in two: smp_gen advances the play head and emits the gather indices (it
// never needs a loaded sample value), smp_eval turns the gathered values into the output.  The
// kernel issues the gathers of chunk k+1 before the stores of chunk k, so waiting for them is a
// counted vmcnt and the store stream is never drained (loads and stores retire in order on one
// counter).  Guarded reads of the reference (`cond ? A[i] : 0`) become a read of a clamped index
// plus a select, which loads the same value whenever the reference loads at all.
template <int MODE>
struct SmpReq {
    static constexpr int L = smp_loads(MODE);
    long long idx[L];
    double rem;
    bool ok;   // modes 1,3,4,5,6: the reference's bounds test; modes 7,8: "backward" branch
};

template <int MODE>
__device__ __forceinline__ void smp_gen(Smp &s, double x, double t, double start, double end,
                                        double sr, SmpReq<MODE> &q) {
    constexpr int B = smp_base(MODE);
    q.rem = 0.0;
    q.ok = true;
    if constexpr (MODE >= 9 && MODE <= 13) {  // C:1006-1042
        const bool zx = (s.tprev <= 0.0 || s.tfirst) && t > 0;  // H:572
        s.tprev = t;
        s.tfirst = false;
        if (zx) {
            if constexpr (MODE == 13) {  // setPosition(pos) C:749-751, maxiMap::clamp H:843-854
                double c = s.p0;
                if (c > 1.0) c = 1.0;
                else if (c < 0.0) c = 0.0;
                s.pos = c * (double)s.len;
            } else {
                s.pos = 0;  // trigger() C:597-600
                if constexpr (MODE == 11 || MODE == 12) s.pos = s.p0 * (double)s.len;  // C:1024, C:1032
            }
        }
    }
    if constexpr (MODE == 14) {  // playWithPhasor C:753-816 (pos1/pos2 are size_t there)
        const unsigned long long amplen = s.len;
        double pha = t;
        if (pha > 1) pha = 1;
        if (pha < 0) pha = 0;
        const double pos = pha * (double)amplen * 0.99999999999999;
        if (s.tfirst) {
            s.tfirst = false;
            s.tprev = pos;
        }
        unsigned long long pos1 = (unsigned long long)(round(s.tprev));
        unsigned long long pos2 = (unsigned long long)(round(pos));
        if (pos1 == pos2) {
            if (pos >= s.tprev) pos2++;
            else pos1--;  // 0 wraps to 2^64-1 and is caught by the next test, as in the reference
        }
        if (pos2 >= amplen) pos2 = 0;
        if (pos1 >= amplen) pos1 = 0;
        double q1;
        if (pos2 > pos1) {
            const double dist = (double)(pos2 - pos1);
            q1 = (dist == 0) ? 0 : (pos - (double)pos1) / dist;
        } else {
            const double dist = (double)((amplen - pos1) + pos2);
            if (dist == 0) q1 = 0;
            else if (pos > (double)pos1) q1 = (pos - (double)pos1) / dist;
            else q1 = ((double)(amplen - pos1) + pos) / dist;
        }
        q.rem = q1;
        q.idx[0] = (long long)pos1;
        q.idx[1] = (long long)pos2;
        s.tprev = pos;
    } else if constexpr (B == 0) {  // C:740-747
        q.idx[0] = (long long)smp_safe_head(s.pos, s.len);
        s.pos += 1.0;
        if ((size_t)(long long)s.pos >= s.len) s.pos = 0;
    } else if constexpr (B == 1) {  // C:982-991
        q.ok = (size_t)(long long)s.pos < s.len;
        q.idx[0] = q.ok ? (long long)s.pos : 0;
        s.pos += 1.0;
    } else if constexpr (B == 2) {  // C:960-967
        s.pos += 1.0;
        double lo = (double)s.len * start;
        if (s.pos < lo) s.pos = lo;
        if ((double)(long long)s.pos >= (double)s.len * end) s.pos = lo;
        q.idx[0] = (long long)smp_safe_head(s.pos, s.len);
    } else if constexpr (B == 3) {  // C:969-978
        s.pos += 1.0;
        if (end > 1.0) end = 1.0;
        q.ok = (double)(long long)s.pos < (double)s.len * end;
        q.idx[0] = q.ok ? (long long)smp_safe_head(s.pos, s.len) : 0;
    } else if constexpr (B == 4 || B == 5 || B == 6) {  // C:1060-1075, C:994-1003, C:1047-1058
        long long i = (long long)s.pos;
        q.rem = s.pos - (double)i;
        if constexpr (B == 4) q.ok = (size_t)i < s.len;
        if constexpr (B == 5) q.ok = (size_t)(i + 1) < s.len;
        if constexpr (B == 6) {
            if (end > 1.0) end = 1.0;
            q.ok = (double)i < (double)s.len * end;
        }
        long long first = (B == 5) ? i : 1 + i;
        // playAtSpeed / playOnceAtSpeed: their bounds test already keeps `first` inside [-1, len]; playUntilAtSpeed only
        // tests the upper end (a head driven below 0 by a negative speed passes it)
        if constexpr (B == 6) first = 1 + (long long)smp_safe_head(s.pos, s.len);
        q.idx[0] = q.ok ? first : 0;
        q.idx[1] = q.idx[0] + 1;
        s.pos = s.pos + ((x * kChandiv) / s.step_div);
        if constexpr (B == 4)
            if ((size_t)(long long)s.pos >= s.len) s.pos -= (double)s.len;
    } else if constexpr (B == 7) {  // C:884-956; idx = {a, b, c, d}
        double frequency = x;
        if (frequency > 0.) {
            if (s.pos < start) s.pos = start;
            if (s.pos >= end) s.pos = start;
            s.pos += ((end - start) / (sr / (frequency * kChandiv)));
            q.rem = s.pos - floor(s.pos);
            const double ps = smp_safe_head(s.pos, s.len);
            q.idx[0] = (s.pos > 0) ? (long long)((int)(floor(ps)) - 1) : 0;
            q.idx[1] = (long long)ps;
            q.idx[2] = (s.pos < end - 2) ? (long long)ps + 1 : 0;
            q.idx[3] = (s.pos < end - 3) ? (long long)ps + 2 : 0;
            q.ok = false;
        } else {
            frequency *= -1.;
            if (s.pos <= start) s.pos = end;
            s.pos -= ((end - start) / (sr / (frequency * kChandiv)));
            q.rem = s.pos - floor(s.pos);
            const double ps = smp_safe_head(s.pos, s.len);
            q.idx[0] = (s.pos > start && s.pos < end - 1) ? (long long)ps + 1 : 0;
            q.idx[1] = (long long)ps;
            q.idx[2] = (s.pos > start) ? (long long)ps - 1 : 0;
            q.idx[3] = (s.pos > start + 1) ? (long long)ps - 2 : 0;
            q.ok = true;
        }
    } else {  // C:823-880: `position` is a by-value parameter there, the head never advances
        double frequency = x, pos = s.pos;
        const size_t amplen = s.len;
        if (end >= (double)amplen) end = (double)(amplen - 1);
        if (frequency > 0.) {
            if (pos < start) pos = start;
            if (pos >= end) pos = start;
            pos += ((end - start) / ((sr) / (frequency * kChandiv)));
            q.rem = pos - floor(pos);
            long long posl = (long long)floor(smp_safe_head(pos, s.len));
            q.idx[0] = ((size_t)(posl + 1) < amplen) ? posl + 1 : posl - 1;
            q.idx[1] = ((size_t)(posl + 2) < amplen) ? posl + 2 : (long long)amplen - 1;
            q.ok = false;
        } else {
            frequency *= -1.;
            if (pos <= start) pos = end;
            pos -= ((end - start) / (sr / (frequency * kChandiv)));
            q.rem = pos - floor(pos);
            long long posl = (long long)floor(smp_safe_head(pos, s.len));
            q.idx[0] = (posl - 1 >= 0) ? posl - 1 : 0;
            q.idx[1] = (posl - 2 >= 0) ? posl - 2 : 0;
            q.ok = true;
        }
    }
}

template <int MODE>
__device__ __forceinline__ double smp_eval(const SmpReq<MODE> &q, const double *val) {
    constexpr int B = smp_base(MODE);
    if constexpr (MODE == 14) {
        const double q2 = 1 - q.rem;
        return (q.rem * val[0] + q2 * val[1]);  // C:810-811
    } else if constexpr (B == 0 || B == 2) {
        return val[0];
    } else if constexpr (B == 1 || B == 3) {
        return q.ok ? val[0] : 0.0;
    } else if constexpr (B == 4 || B == 5 || B == 6) {
        double o = ((1 - q.rem) * val[0] + q.rem * val[1]);
        return q.ok ? o : 0.0;
    } else if constexpr (B == 7) {
        const double a = val[0], b = val[1], c = val[2], d = val[3];
        double a1 = 0.5 * (c - a);
        double a2 = a - 2.5 * b + 2. * c - 0.5 * d;
        double a3 = 0.5 * (d - a) + 1.5 * (b - c);
        const double m = q.ok ? -q.rem : q.rem;  // C:950 multiplies by -remainder going backwards
        return (((a3 * q.rem + a2) * m + a1) * m + b);
    } else {
        const double w = q.ok ? (-1 - q.rem) : (1 - q.rem);  // C:872 / C:850
        return (w * val[0] + q.rem * val[1]);
    }
}

}  // namespace
}  // namespace mxg
