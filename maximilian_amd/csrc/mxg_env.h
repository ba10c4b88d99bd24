// mxg_env.h -- maxiEnv (src/maximilian.cpp:1319-1494) per-lane state and tick functions, shared by the
// envelope / fused-voice kernels (voice.hip) and the sampler kernel (sampler.hip).
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#include "mxg_gate.h"
#else  // host build of the same text (tests/host_env.cpp: state machine and steady-state paths against the oracle)
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#define __device__
#define __forceinline__ inline
#endif

namespace mxg {
namespace {

// ---- maxiEnv ----------------------------------------------------------------------------
struct Env {
    double attack, decay, sustain, release, amplitude, output;
    long long holdtime, holdcount;
    int attackphase, decayphase, sustainphase, holdphase, releasephase;
};

__device__ __forceinline__ void env_load(Env &e, size_t V, size_t v, const double *par,
                                         const int64_t *holdtime, const double *dst,
                                         const int64_t *ist) {
    e.attack = par[v];
    e.decay = par[V + v];
    e.sustain = par[2 * V + v];
    e.release = par[3 * V + v];
    e.holdtime = holdtime[v];
    e.amplitude = dst[v];
    e.output = dst[V + v];
    e.holdcount = ist[v];
    e.attackphase = (int)ist[V + v];
    e.decayphase = (int)ist[2 * V + v];
    e.sustainphase = (int)ist[3 * V + v];
    e.holdphase = (int)ist[4 * V + v];
    e.releasephase = (int)ist[5 * V + v];
}
__device__ __forceinline__ void env_store(const Env &e, size_t V, size_t v, double *dst,
                                          int64_t *ist) {
    dst[v] = e.amplitude;
    dst[V + v] = e.output;
    ist[v] = e.holdcount;
    ist[V + v] = e.attackphase;
    ist[2 * V + v] = e.decayphase;
    ist[3 * V + v] = e.sustainphase;
    ist[4 * V + v] = e.holdphase;
    ist[5 * V + v] = e.releasephase;
}

// C:1415-1466.  Statement order is the reference's: a sample can pass through several of
// the `if`s (e.g. attack -> decay in the same call).
// Written as straight-line selects (same statement order, same arithmetic): the lanes of a wave sit
// in different envelope phases, and as nested `if`s this compiled to ~13 exec-mask regions per
// sample (26 s_cbranch_execz per 4 samples), 4x slower than the predicated form.
// GAIN form (round 6): every assignment of C:1425-1463 to `output` is input * (the amplitude at that statement), so one call leaves
// output = input * g with g the amplitude at the LAST such statement that fired, or leaves output alone (upd = false) -- a select among
// products of one input is the product with the selected factor, the same bits.  The state machine is this function; env_adsr
// multiplies once.  The fused voice's two-stage form (voice.hip, voice_split_kernel) runs the gains a stage ahead of the filter.
__device__ __forceinline__ double env_adsr_gain(Env &e, int trigger, bool &upd) {
    const bool t1 = trigger == 1;
    // C:1417-1423
    const bool c1 = t1 && e.attackphase != 1 && e.holdphase != 1 && e.decayphase != 1;
    e.holdcount = c1 ? 0 : e.holdcount;
    e.decayphase = c1 ? 0 : e.decayphase;
    e.sustainphase = c1 ? 0 : e.sustainphase;
    e.releasephase = c1 ? 0 : e.releasephase;
    e.attackphase = c1 ? 1 : e.attackphase;
    double amp = e.amplitude, g = 0.0;
    // C:1425-1435  attack
    const bool a = e.attackphase == 1;
    e.releasephase = a ? 0 : e.releasephase;
    const double ampA = amp + (1 * e.attack);
    amp = a ? ampA : amp;
    g = a ? amp : g;
    const bool a2 = a && amp >= 1;
    amp = a2 ? 1.0 : amp;
    e.attackphase = a2 ? 0 : e.attackphase;
    e.decayphase = a2 ? 1 : e.decayphase;
    // C:1438-1444  decay
    const bool d = e.decayphase == 1;
    const double ampD = amp * e.decay;
    amp = d ? ampD : amp;
    g = d ? amp : g;
    const bool d2 = d && amp <= e.sustain;
    e.decayphase = d2 ? 0 : e.decayphase;
    e.holdphase = d2 ? 1 : e.holdphase;
    // C:1446-1449  hold
    const bool h = e.holdcount < e.holdtime && e.holdphase == 1;
    g = h ? amp : g;
    e.holdcount += h ? 1 : 0;
    // C:1451-1458
    const bool ge = e.holdcount >= e.holdtime;
    const bool s = ge && t1;
    g = s ? amp : g;
    const bool rel = ge && !t1;
    e.holdphase = rel ? 0 : e.holdphase;
    e.releasephase = rel ? 1 : e.releasephase;
    // C:1460-1463  release
    const bool r = e.releasephase == 1 && amp > 0.;
    const double ampR = amp * e.release;
    amp = r ? ampR : amp;
    g = r ? amp : g;
    e.amplitude = amp;
    upd = a || d || h || s || r;
    return g;
}
__device__ __forceinline__ double env_adsr(Env &e, double input, int trigger) {
    bool upd;
    const double g = env_adsr_gain(e, trigger, upd);
    e.output = upd ? input * g : e.output;
    return e.output;
}

// Two steady states of the ADSR in which a sample is one multiply and no flag moves (read off
// C:1415-1466): SUSTAIN = gate held after the hold time ran out (only C:1451-1453 fires:
// output = input*amplitude), RELEASE = gate off after it (only C:1455-1463: amplitude *= release
// while it is > 0).  A wavefront whose lanes are all in one of them, with the (shared, scalar) gate
// constant over a chunk, runs the chunk without the predicated state machine -- same operations on
// the same values, so bit-identical, at ~1/6 of the VALU work.
__device__ __forceinline__ bool env_in_sustain(const Env &e) {
    return e.attackphase != 1 && e.decayphase != 1 && e.releasephase != 1 && e.holdphase == 1 &&
           e.holdcount >= e.holdtime;
}
__device__ __forceinline__ bool env_in_release(const Env &e) {
    // holdphase == 0, not merely != 1: C:1456 stores 0 into it on every such sample, so any other value (a state
    // uploaded by the host) has to go through the state machine once
    return e.attackphase != 1 && e.decayphase != 1 && e.releasephase == 1 && e.holdphase == 0 &&
           e.holdcount >= e.holdtime;
}
__device__ __forceinline__ double env_sustain_tick(Env &e, double input) {  // gate == 1
    e.output = input * e.amplitude;  // C:1452
    return e.output;
}
__device__ __forceinline__ double env_release_tick(Env &e, double input) {  // gate != 1
    const bool r = e.amplitude > 0.;  // C:1460
    const double ampR = e.amplitude * e.release;
    e.amplitude = r ? ampR : e.amplitude;
    e.output = r ? input * e.amplitude : e.output;
    return e.output;
}
// The general steady chunk: U samples during which no flag of this envelope moves, whatever stage it is in.  Read off
// C:1415-1466 with a constant gate, a lane that stays inside ONE stage for the whole chunk executes per sample
//   attack   amplitude += attack                 output = input*amplitude     (while amplitude < 1)
//   decay    amplitude *= decay                  output = input*amplitude     (while amplitude > sustain)
//   hold     holdcount++ (up to holdtime)        output = input*amplitude
//   sustain  (hold over, gate on)                output = input*amplitude
//   release  amplitude *= release (gate off)     output = input*amplitude     (while amplitude > 0)
//   idle     nothing                              output unchanged
// i.e. amplitude = amplitude*m + a with (m, a) = (1, attack), (decay, 0), (1, 0), (release, 0) -- x*1 and x+0 are exact for
// the non-negative amplitudes admitted here, so the two roundings are the reference's one -- and one multiply.  The lanes of a
// wavefront may each be in a DIFFERENT stage (a polyphonic bank is), which is what the sustain / release paths above cannot
// take.  The chunk is computed speculatively on a copy of the state; it is valid if the stage's own exit test would have been
// false on every sample, which for these monotone recurrences is a test of the LAST value (attack: < 1, decay: > sustain,
// release: > 0) plus the conditions on the flags below.  Returns false (and must then be discarded) otherwise; the caller
// commits only if every lane of the wavefront returned true, else the chunk goes through env_adsr.  ~6 operations per sample
// instead of ~100.
// (gain form, as env_adsr_gain: g[i] = the amplitude sample i multiplies its input by, upd = false for an idle lane, whose output stays)
template <int U>
__device__ __forceinline__ bool env_steady_gains(Env &e, const bool gate, double (&g)[U], bool &upd) {
    const bool ap = e.attackphase == 1, dp = e.decayphase == 1, hp = e.holdphase == 1, rp = e.releasephase == 1;
    const bool ge = e.holdcount >= e.holdtime;
    long long hb;
    {
        double t = e.amplitude;
#if defined(__HIPCC__)
        hb = __double_as_longlong(t);
#else
        memcpy(&hb, &t, 8);
#endif
    }
    const bool nonneg = hb >= 0;  // sign bit clear: +0 and positive values (and positive NaNs, which fail every test below)
    // stage of this lane (exactly one phase flag set, the others exactly as the stage leaves them)
    const bool A = ap && !dp && !hp && e.releasephase == 0 && (gate || !ge) && e.attack > 0.0;  // (C:1426 stores 0 there)
    const bool D = dp && !ap && !hp && !rp && (gate || !ge) && e.decay > 0.0 && e.decay <= 1.0;
    // hold / sustain: with the gate on the count simply saturates at holdtime (C:1446-1453 give the same output either side);
    // with the gate off the sample on which it reaches holdtime starts the release (C:1455-1458), so that chunk is not steady
    const bool H = hp && !ap && !dp && !rp && (gate || e.holdcount + (long long)U < e.holdtime);
    const bool R = !gate && rp && !ap && !dp && e.holdphase == 0 && ge && e.amplitude > 0.0 && e.release > 0.0 && e.release <= 1.0;
    // idle: gate off, no stage active, and neither C:1455-1458 nor C:1460-1463 would change anything
    const bool Z = !gate && !ap && !dp && !hp && !(rp && e.amplitude > 0.0) && (!ge || (e.holdphase == 0 && rp));
    const double m = D ? e.decay : (R ? e.release : 1.0);
    const double a = A ? e.attack : 0.0;
    double amp = e.amplitude;
#pragma unroll
    for (int i = 0; i < U; i++) {
        amp = (amp * m) + a;
        g[i] = amp;
    }
    bool ok = nonneg && (A || D || H || R || Z);
    ok = ok && (!A || amp < 1.0) && (!D || amp > e.sustain) && (!R || amp > 0.0);
    e.amplitude = amp;
    if (H && e.holdcount < e.holdtime) {
        const long long c = e.holdcount + (long long)U;
        e.holdcount = c < e.holdtime ? c : e.holdtime;
    }
    upd = !Z;
    return ok;
}
template <int U>
__device__ __forceinline__ bool env_steady_chunk(Env &e, const double (&x)[U], const bool gate, double (&o)[U]) {
    double g[U];
    bool upd;
    const bool ok = env_steady_gains<U>(e, gate, g, upd);
    double out = e.output;
#pragma unroll
    for (int i = 0; i < U; i++) {
        out = upd ? x[i] * g[i] : out;
        o[i] = out;
    }
    e.output = out;
    return ok;
}

// gate state of a full chunk from its (already fetched, wave-uniform) trigger values:
// +1 all == 1, -1 all != 1, 0 mixed.  Scalar work.
template <int U>
__device__ __forceinline__ int gate_of_chunk(const int (&t)[U]) {
    int on = 0;
#pragma unroll
    for (int i = 0; i < U; i++) on += (t[i] == 1) ? 1 : 0;
    return on == U ? 1 : (on == 0 ? -1 : 0);
}

// C:1319-1358
__device__ __forceinline__ double env_ar(Env &e, double input, int trigger) {
    const double attack = e.attack, release = e.release;
    const long long holdtime = e.holdtime;
    if (trigger == 1 && e.attackphase != 1 && e.holdphase != 1) {
        e.holdcount = 0;
        e.releasephase = 0;
        e.attackphase = 1;
    }
    if (e.attackphase == 1) {
        e.amplitude += (1 * attack);
        e.output = input * e.amplitude;
    }
    if (e.amplitude >= 1) {
        e.amplitude = 1;
        e.attackphase = 0;
        e.holdphase = 1;
    }
    if (e.holdcount < holdtime && e.holdphase == 1) {
        e.output = input;
        e.holdcount++;
    }
    if (e.holdcount == holdtime && trigger == 1) {
        e.output = input;
    }
    if (e.holdcount == holdtime && trigger != 1) {
        e.holdphase = 0;
        e.releasephase = 1;
    }
    if (e.releasephase == 1 && e.amplitude > 0.) {
        e.amplitude *= release;
        e.output = input * e.amplitude;
    }
    return e.output;
}

}  // namespace
}  // namespace mxg
