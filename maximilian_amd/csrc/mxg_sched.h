// mxg_sched.h -- the scheduler of a granular stream (L/maxiGrains.h:341-355 maxiTimeStretch::play, :359-367
// playAtPosition, :412-430 maxiPitchShift::play, :512-530 maxiStretch::play): constants, state, the one-sample step and the
// event-driven walk.  Plain arithmetic; shared by grains.hip's kernels and -- compiled for the host -- by
// tests/host_sched_events.cpp, which checks the event-driven walk against the one-sample step.
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#else
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif
#endif
#include "mxg_advance.h"

namespace mxg {
namespace {

// ---- the per-stream scheduler, one sample at a time ---------------------------------------------
// MODE 0 maxiTimeStretch::play :341-355, 1 maxiStretch::play :512-530 (loop = whole sample),
// 2 maxiTimeStretch::playAtPosition :359-367 (a = per-sample pos [T][S]; `position` untouched),
// 3 maxiPitchShift::play :412-430 (the `looper` slot holds the member `cycles`, a long).
struct SchedConst {
    double dlen, cycleLength, grainLength, sr, pm, speed, rate;
    int sampleDur;
    const double *a_ps;   // MODE 2: &pos[0][s], stride S
    size_t S;
    const int32_t *rnd;   // this stream's rand()%10 draws (or null)
    size_t R;
};
struct SchedState {
    double position, looper, randomOffset, thr;
    size_t cursor;
};

// maxiGrain ctor (L/maxiGrains.h:160-181): where the grain starts and its per-sample increment
// A step the reference could not survive either -- NaN/Inf, or longer than the sample so that the single wrap of
// maxiGrain::play (:219-224) leaves pos outside [0, len) and the reads go out of bounds -- is refused: failed = 5,
// the grain is born with inc = 0 (it stays inside the buffer) and the launch reports MXG_ERR_INVALID.
__device__ __forceinline__ void grain_birth(const SchedConst &c, double p01, const double grainSpeed,
                                            double &pos0, double &inc, int &failed) {
    p01 = 1.0 < p01 ? 1.0 : p01;  // max(min(1.0, .), 0.0)
    p01 = p01 < 0.0 ? 0.0 : p01;
    const double startPos = floor(c.dlen * p01);  // (unsigned long)(len*pos), pos >= 0
    double endPos = startPos + (double)c.sampleDur;
    endPos = c.dlen < endPos ? c.dlen : endPos;
    const double frequency = (1.0 / c.grainLength) * grainSpeed;
    pos0 = frequency > 0 ? startPos : endPos;
    inc = (frequency != 0) ? (double)c.sampleDur / (c.sr / frequency) : 0.0;
    if (!(fabs(inc) <= c.dlen)) {
        failed = 5;
        inc = 0.0;
    }
}

// `randomOffset = rand() % 10` (:352 / :525) from the caller-supplied draws
__device__ __forceinline__ void sched_draw(SchedState &q, const SchedConst &c, int &failed) {
    if (c.rnd) {
        if (q.cursor < c.R) q.randomOffset = (double)c.rnd[q.cursor]; else { failed = 2; q.randomOffset = 0; }
        q.cursor++;
    } else {
        q.randomOffset = 0;
    }
    q.thr = c.cycleLength + q.randomOffset;
}

// modes 0/1: the body of `if (looper > cycleLength + randomOffset)` (:347-353 / :519-526)
template <int MODE>
__device__ __forceinline__ void sched_spawn01(SchedState &q, const SchedConst &c, double &pos0, double &inc,
                                              int &failed) {
    q.looper -= q.thr;
    const double grainSpeed = MODE == 0 ? (c.speed > 0 ? 1.0 : -1.0) : c.speed;  // :350
    grain_birth(c, (q.position / c.dlen) + c.pm, grainSpeed, pos0, inc, failed);
    sched_draw(q, c, failed);
}

// advances the scheduler by one sample; true if a grain (pos0, inc) is born at this sample
template <int MODE>
__device__ __forceinline__ bool sched_step(SchedState &q, const SchedConst &c, size_t n, double &pos0,
                                           double &inc, int &failed) {
    if constexpr (MODE == 2) {
        q.looper += 1.0;
        double pos = c.a_ps[n * c.S];
        pos *= c.dlen;
        if (0 == floor(fmod(q.looper, c.cycleLength))) {  // :362
            grain_birth(c, (pos / c.dlen), 1.0, pos0, inc, failed);
            return true;
        }
        return false;
    } else if constexpr (MODE == 3) {
        q.position = q.position + 1;
        q.looper += 1.0;  // cycles++
        if (q.position > c.dlen) q.position = 0;
        if (q.position < 0) q.position = c.dlen;
        const double cycleMod = fmod(q.looper, c.cycleLength + q.randomOffset);
        if (0 == floor(cycleMod)) {
            const double sp = c.speed - ((cycleMod / c.cycleLength) * 0.1);  // :421
            grain_birth(c, (q.position / c.dlen) + c.pm, sp, pos0, inc, failed);
            return true;
        }
        return false;
    } else {
        q.position = q.position + c.rate;
        q.looper += 1.0;
        if (MODE == 0) {
            if (q.position > c.dlen) q.position -= c.dlen;
            if (q.position < 0) q.position += c.dlen;
        } else {  // loopStart 0, loopEnd = loopLength = len (maxiStretch ctor :469-477)
            if (q.position >= c.dlen) q.position -= c.dlen;
            if (q.position < 0.0) q.position += c.dlen;
        }
        if (q.looper > q.thr) {
            sched_spawn01<MODE>(q, c, pos0, inc, failed);
            return true;
        }
        return false;
    }
}

template <int MODE>
__device__ __forceinline__ SchedConst sched_const(size_t s, size_t S, size_t len, size_t R, const double *a,
                                                  const double *b, const double *posMod, const int32_t *rnd,
                                                  double cycleLength, double grainLength, double sr,
                                                  int sampleDur) {
    SchedConst c;
    c.dlen = (double)len;
    c.cycleLength = cycleLength;
    c.grainLength = grainLength;
    c.sr = sr;
    c.pm = posMod ? posMod[s] : 0.0;
    c.speed = MODE == 2 ? 1.0 : a[s];
    c.rate = MODE == 1 ? b[s] : c.speed;  // what advances `position` each sample (modes 0/1)
    c.sampleDur = sampleDur;
    c.a_ps = a + s;
    c.S = S;
    c.rnd = rnd ? rnd + s * R : nullptr;
    c.R = R;
    return c;
}

// The event-driven part of a stream's walk over Tn samples: jumps from birth to birth with the exact multi-step forms
// of mxg_advance.h wherever the state allows it and calls record(n, pos0, inc) for every grain born (n = sample index).
// Returns the number of samples it has consumed; the caller walks the rest one sample at a time with sched_step --
// from a state that is, bit for bit, the one sched_step would have reached itself (tests/host_sched_events.cpp).
template <int MODE, typename Record>
__device__ __forceinline__ int sched_run_events(SchedState &q, const SchedConst &sc, int Tn, bool fast_enabled, int &failed,
                                                Record record) {
    int nstart = 0;
    const double dlen = sc.dlen, rate = sc.rate;
    if constexpr (MODE <= 1) {
        const bool fast = fast_enabled && rate > 0.0 && q.position >= 0.0 && q.position <= dlen && q.looper >= 0.0 &&
                          q.thr > 1.0;
        if (fast) {
            // event-driven: jump from spawn to spawn (looper), dragging position along (with its wraps)
            int n = 0;
            while (n < Tn) {
                bool spawn;
                const int k = advance_until(q.looper, 1.0, q.thr, false, Tn - n, spawn);
                int rem = k;
                while (rem > 0) {
                    bool wrapped;
                    rem -= advance_until(q.position, rate, dlen, MODE == 1, rem, wrapped);
                    if (wrapped) q.position -= dlen;  // :344 / :516 (the `< 0` branch cannot fire for rate > 0)
                }
                n += k;
                if (spawn) {
                    double pos0, inc;
                    sched_spawn01<MODE>(q, sc, pos0, inc, failed);
                    record(n - 1, pos0, inc);
                }
            }
            nstart = Tn;
        }
    }
    if constexpr (MODE == 2 || MODE == 3) {
        // playAtPosition / maxiPitchShift: the counter (`looper` / `cycles`) just counts samples and a grain is born
        // when floor(fmod(counter, cycle)) == 0 (:362, :419-420); nothing else happens in between (playAtPosition
        // reads its position signal on birth samples only; maxiPitchShift's `position` is a +1 ramp with a reset).
        // With an integer-valued counter the additions are exact, so the kernel jumps from birth to birth
        // (next_birth: one multiplication, then the exact predicate itself as the judge).
        const double cyc = MODE == 2 ? sc.cycleLength : sc.cycleLength + q.randomOffset;
        bool eligible = fast_enabled && cyc > 2.0 && q.looper >= 0.0 && q.looper == floor(q.looper) &&
                        q.looper + (double)Tn < 4.0e15;
        if constexpr (MODE == 3) eligible = eligible && q.position >= 0.0 && q.position <= dlen;
        if (eligible) {
            const double L0 = q.looper;  // counter before sample 0; sample n sees L0 + n + 1
            double L = L0;
            int ndone = 0;               // samples whose `position` step has been applied (MODE 3)
            bool ok = true;
            auto ramp = [&](int upto) {  // maxiPitchShift :415-417 for samples [ndone, upto)
                int rem = upto - ndone;
                while (rem > 0) {
                    bool wrapped;
                    rem -= advance_until(q.position, 1.0, dlen, false, rem, wrapped);
                    if (wrapped) q.position = 0;  // `if (position > len) position = 0`
                }
                ndone = upto;
            };
            for (;;) {
                const double Lc = next_birth(L, cyc, ok);
                if (!ok) break;
                const double nn = Lc - L0 - 1.0;  // sample index of that birth
                if (!(nn < (double)Tn)) break;
                const int n = (int)nn;
                double pos0, inc;
                if constexpr (MODE == 2) {
                    double pos = sc.a_ps[(size_t)n * sc.S];
                    pos *= sc.dlen;
                    grain_birth(sc, (pos / sc.dlen), 1.0, pos0, inc, failed);
                } else {
                    ramp(n + 1);
                    const double cycleMod = fmod(Lc, cyc);
                    const double sp = sc.speed - ((cycleMod / sc.cycleLength) * 0.1);  // :421
                    grain_birth(sc, (q.position / sc.dlen) + sc.pm, sp, pos0, inc, failed);
                }
                record(n, pos0, inc);
                L = Lc;
            }
            if (ok) {
                if constexpr (MODE == 3) ramp(Tn);
                q.looper = L0 + (double)Tn;
                nstart = Tn;
            } else {  // resume one sample at a time right after the last confirmed birth
                q.looper = L;
                nstart = (int)(L - L0);
                if constexpr (MODE == 3) ramp(nstart);
            }
        }
    }
    return nstart;
}

}  // namespace
}  // namespace mxg
