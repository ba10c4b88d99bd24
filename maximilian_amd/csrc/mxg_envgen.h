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
// runs the chunk with the stage row read once, the three zero-crossing detectors (H:569-579) advanced exactly as the state
// machine advances them, and ~1/5 of the instructions.  The lanes of a wavefront may each be in a different one of the three.
// Computed on a copy of the state: returns false -- discard the copy -- if an exit condition fired on any of the U samples
// (a crossing that changes the state, the stage running out) or the lane is in none of them (a HOLD stage entered while
// TRIGGERED, phase == S); the caller commits only if every lane returned true, else the chunk goes through envgen_tick.
template <int U>
__device__ __forceinline__ bool envgen_steady_chunk(EgState &e, const double *tab, long long S, bool retrigger,
                                                    const double (&t)[U], double (&o)[U]) {
    if (e.phase < 0 || e.phase >= S) return false;  // (H:2349-2355 fires on phase == S in every state)
    const double *cs = tab + 6 * e.phase;
    const double start = cs[0], span = cs[1] - cs[0], gradient = cs[2], curve = cs[3];
    const long long length = (long long)cs[4];
    // (linear stages only -- every stage of setupAR / setupASR / setupADSR: a curved stage is dominated by its pow() either way)
    const bool ramp = e.state == EG_TRIGGERED && cs[5] == 0 && curve == 1.0 && e.counter >= 0 && e.counter + (long long)U < length;
    const bool hold = e.state == EG_HOLDING && !e.nxc;
    const bool wait = e.state == EG_WAITING;
    bool ok = ramp || hold || wait;
    double envval = e.envval, cl = e.currentlevel;
    bool nxc = e.nxc;
#pragma unroll
    for (int i = 0; i < U; i++) {
        if (wait) {
            if (on_zx(e.tprev, e.tfirst, t[i])) ok = false;  // H:2281: the envelope starts
        } else {
            const bool neg = on_zx(e.hprev, e.hfirst, -t[i]);  // H:2294 / H:2331
            nxc = nxc || neg;
            if (hold && neg) ok = false;                        // H:2334: the hold ends
            if (retrigger && on_zx(e.rprev, e.rfirst, t[i])) ok = false;  // H:2323 / H:2341: reset
        }
        double val = cl;  // H:2302-2304 with curve == 1, as envgen_tick
        val = (1.0 < val) ? 1.0 : val;
        val = (val < 0.0) ? 0.0 : val;
        const double nv = ((val - 0.0) / (1.0 - 0.0) * span) + start;
        envval = ramp ? nv : envval;
        cl = ramp ? cl + gradient : cl;
        o[i] = envval;
    }
    e.envval = envval;
    e.currentlevel = cl;
    e.nxc = nxc;
    if (ramp) e.counter += (long long)U;
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
