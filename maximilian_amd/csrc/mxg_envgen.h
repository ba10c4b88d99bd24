// mxg_envgen.h -- maxiEnvGen::play (src/maximilian.h:2277-2356) for one envelope: per-lane state and one sample of the
// three-state machine (the reference's switch with its fall-through made explicit).  Plain arithmetic, shared by
// envgen.hip's kernel and -- compiled for the host -- by tests/host_envgen.cpp, which runs it against the oracle from
// arbitrary states.
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#else
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#define __device__
#define __forceinline__ inline
#endif

namespace mxg {
namespace {

enum { EG_WAITING = 0, EG_TRIGGERED = 1, EG_HOLDING = 2 };

// One envelope.  The reference keeps counter / currentlevel per stage; only the stage `phase` points at is ever non-zero
// (they are zeroed when a stage is left and by reset(), H:2402-2410), so one pair per envelope carries the same state.
struct EgState {
    double envval, currentlevel;
    double tprev, hprev, rprev;   // previousValue of trigDetector / holdDetector / retriggerDetector (H:593)
    bool tfirst, hfirst, rfirst;  // their firstTrigger (H:594)
    long long phase, counter;
    int state;
    bool nxc;                     // nxcHappened
};

__device__ __forceinline__ bool on_zx(double &prev, bool &first, double input) {  // H:569-579
    const bool zx = (prev <= 0.0 || first) && input > 0;
    prev = input;
    first = false;
    return zx;
}

// tab: [S][6] = startlevel, endlevel, gradient, curve, length, hold (mxg_envgen_stages_host).  Returns envval.
__device__ __forceinline__ double envgen_tick(EgState &e, const double *tab, long long S, bool loop, bool retrigger,
                                              double trigger) {
    int entry = e.state;  // the switch's fall-through, made explicit
    if (entry == EG_WAITING) {  // H:2279-2292
        if (on_zx(e.tprev, e.tfirst, trigger)) {
            e.state = EG_TRIGGERED;
            e.nxc = false;
            entry = EG_TRIGGERED;
        } else {
            entry = -1;
        }
    }
    if (entry == EG_TRIGGERED) {  // H:2293-2329
        const double *cs = tab + 6 * e.phase;
        if (on_zx(e.hprev, e.hfirst, -trigger)) e.nxc = true;
        if (cs[5] != 0) {
            e.state = EG_HOLDING;
            entry = EG_HOLDING;
        } else {
            const double curve = cs[3];
            double val = (curve == 1.0) ? e.currentlevel : pow(e.currentlevel, curve);
            val = (1.0 < val) ? 1.0 : val;  // linlin: max(min(val, inMax), inMin)
            val = (val < 0.0) ? 0.0 : val;
            e.envval = ((val - 0.0) / (1.0 - 0.0) * (cs[1] - cs[0])) + cs[0];
            e.counter++;
            if (e.counter == (long long)cs[4]) {
                e.counter = 0;
                e.currentlevel = 0;
                e.phase++;
            } else {
                e.currentlevel += cs[2];
            }
            if (retrigger && on_zx(e.rprev, e.rfirst, trigger)) {
                e.nxc = false;
                e.counter = 0; e.currentlevel = 0; e.phase = 0; e.state = EG_TRIGGERED;  // reset() H:2402-2410
            }
            entry = -1;
        }
    }
    if (entry == EG_HOLDING) {  // H:2330-2348
        if (on_zx(e.hprev, e.hfirst, -trigger)) e.nxc = true;
        if (e.nxc) {
            e.state = EG_TRIGGERED;
            e.phase++;
        }
        if (retrigger && on_zx(e.rprev, e.rfirst, trigger)) {
            e.nxc = false;
            e.counter = 0; e.currentlevel = 0; e.phase = 0; e.state = EG_TRIGGERED;
        }
    }
    if (e.phase == S) {  // H:2349-2355: reset() / resetAndArm()
        e.counter = 0; e.currentlevel = 0;
        e.phase = 0;
        e.state = loop ? EG_TRIGGERED : EG_WAITING;
    }
    return e.envval;
}

// The steady chunk: U samples during which this envelope neither changes state nor leaves its stage, whatever it is doing.
// Read off play() (H:2277-2356), a lane that is
//   RAMP   TRIGGERED inside a timed linear stage with more than U samples of it left: per sample the value of the stage's line,
//          counter++, currentlevel += gradient; the hold detector may set nxcHappened (it only matters later, in the HOLD
//          stage); a retrigger crossing would reset the envelope;
//   HOLD   HOLDING with no negative crossing seen: the value stands; a crossing of -trigger leaves the stage;
//   WAIT   WAITING: the value stands; a positive crossing of the trigger starts the envelope;
// runs the chunk from the stage row it keeps in registers (EgRow, re-read only when the stage machine has moved `phase`), with
// the three zero-crossing detectors (H:569-579) advanced in ONE step: a detector fed x[0..U) from (previousValue, firstTrigger)
// fires somewhere in the chunk iff  ((previousValue <= 0 || firstTrigger) && x[0] > 0)  or  x[i-1] <= 0 && x[i] > 0  for some
// 0 < i < U, and ends at (x[U-1], false) either way; the second term does not depend on the envelope (EgCross -- per chunk,
// per wavefront for a shared gate).  Straight-line code, ~1/10 of the stage machine's instructions; the lanes of a wavefront
// may each be in a different one of the three.  Computed on a copy of the state: returns false -- discard the copy -- if an
// exit condition fires inside the chunk (a crossing that changes the state, the stage running out) or the lane is in none of
// the three (a HOLD stage entered while TRIGGERED, phase == S, a curved stage, currentlevel outside [0, 1] where linlin's clamp
// would act); the caller commits only if every lane returned true, else the chunk goes through envgen_tick.
struct EgRow {
    double start, span, gradient;
    long long length;
    bool timed_linear;  // a timed stage (no HOLD) with curve == 1 (every stage of setupAR / setupASR / setupADSR) and gradient >= 0
    bool valid;         // phase inside the table (phase == S -- reachable through an uploaded state -- is play()'s end-of-envelope
                        // test, H:2349-2355, which runs on every sample in every state: never a steady chunk)
};
__device__ __forceinline__ EgRow envgen_row(const double *tab, long long S, long long phase) {
    EgRow r = {0.0, 0.0, 0.0, 0, false, false};
    if (phase < 0 || phase >= S) return r;
    r.valid = true;
    const double *cs = tab + 6 * phase;
    r.start = cs[0];
    r.span = cs[1] - cs[0];
    r.gradient = cs[2];
    r.length = (long long)cs[4];
    r.timed_linear = cs[5] == 0 && cs[3] == 1.0 && cs[2] >= 0.0;
    return r;
}
struct EgCross {
    double first, last;  // t[0], t[U-1]
    bool pos, neg;       // a crossing between two samples of the chunk: of the trigger (trigDetector / retriggerDetector), of -trigger (holdDetector)
};
template <int U>
__device__ __forceinline__ EgCross envgen_cross(const double (&t)[U]) {
    EgCross c = {t[0], t[U - 1], false, false};
#pragma unroll
    for (int i = 1; i < U; i++) {
        c.pos = c.pos || (t[i - 1] <= 0.0 && t[i] > 0);
        c.neg = c.neg || (-t[i - 1] <= 0.0 && -t[i] > 0);
    }
    return c;
}
template <int U>
__device__ __forceinline__ bool envgen_steady_chunk(EgState &e, const EgRow &r, bool retrigger, const EgCross &x, double (&o)[U]) {
    // linlin's clamp (H:801-805) is the identity on [0, 1]; currentlevel only grows (gradient >= 0), so the first and the last
    // level used bound the chunk.  (U - 1) additions in the reference's order:
    double cl[U + 1];
    cl[0] = e.currentlevel;
#pragma unroll
    for (int i = 0; i < U; i++) cl[i + 1] = cl[i] + r.gradient;
    const bool ramp = e.state == EG_TRIGGERED && r.timed_linear && e.counter >= 0 && e.counter + (long long)U < r.length &&
                      cl[0] >= 0.0 && cl[U - 1] <= 1.0;
    const bool hold = e.state == EG_HOLDING && !e.nxc;
    const bool wait = e.state == EG_WAITING;
    const bool trig_fires = ((e.tprev <= 0.0 || e.tfirst) && x.first > 0) || x.pos;    // H:2281
    const bool hold_fires = ((e.hprev <= 0.0 || e.hfirst) && -x.first > 0) || x.neg;   // H:2294 / H:2331
    const bool retr_fires = ((e.rprev <= 0.0 || e.rfirst) && x.first > 0) || x.pos;    // H:2323 / H:2341
    const bool ok = r.valid && (wait ? !trig_fires : ((ramp || (hold && !hold_fires)) && !(retrigger && retr_fires)));
#pragma unroll
    for (int i = 0; i < U; i++) {
        const double nv = cl[i] * r.span + r.start;  // H:2302-2304 with curve == 1 and the clamp an identity
        o[i] = ramp ? nv : e.envval;
    }
    e.envval = o[U - 1];
    e.currentlevel = ramp ? cl[U] : e.currentlevel;
    e.counter = ramp ? e.counter + (long long)U : e.counter;
    e.tprev = wait ? x.last : e.tprev;
    e.tfirst = e.tfirst && !wait;
    e.nxc = e.nxc || (!wait && hold_fires);
    e.hprev = wait ? e.hprev : -x.last;
    e.hfirst = e.hfirst && wait;
    const bool re = retrigger && !wait;
    e.rprev = re ? x.last : e.rprev;
    e.rfirst = e.rfirst && !re;
    return ok;
}

// the [5][V] / [7][V] state arrays of mxg_envgen_render
__device__ __forceinline__ void envgen_load(EgState &e, size_t V, size_t v, const double *dst, const int64_t *ist) {
    e.envval = dst[v]; e.currentlevel = dst[V + v];
    e.tprev = dst[2 * V + v]; e.hprev = dst[3 * V + v]; e.rprev = dst[4 * V + v];
    e.phase = ist[v]; e.state = (int)ist[V + v]; e.nxc = ist[2 * V + v] != 0; e.counter = ist[3 * V + v];
    e.tfirst = ist[4 * V + v] != 0; e.hfirst = ist[5 * V + v] != 0; e.rfirst = ist[6 * V + v] != 0;
}
__device__ __forceinline__ void envgen_store(const EgState &e, long long S, size_t V, size_t v, double *dst, int64_t *ist) {
    const bool in = e.phase < S;
    dst[v] = e.envval;
    dst[V + v] = in ? e.currentlevel : 0.0;
    dst[2 * V + v] = e.tprev; dst[3 * V + v] = e.hprev; dst[4 * V + v] = e.rprev;
    ist[v] = e.phase;
    ist[V + v] = e.state;
    ist[2 * V + v] = e.nxc ? 1 : 0;
    ist[3 * V + v] = in ? e.counter : 0;
    ist[4 * V + v] = e.tfirst ? 1 : 0; ist[5 * V + v] = e.hfirst ? 1 : 0; ist[6 * V + v] = e.rfirst ? 1 : 0;
}

}  // namespace
}  // namespace mxg
