// mxg_osc.h -- one sample of a maxiOsc voice (src/maximilian.cpp:228-373): the block-constant sub-expressions (osc_pre) and
// the per-sample tick of each waveform (osc_tick).  Plain arithmetic over the two static tables, shared by osc.hip's
// kernels and -- compiled for the host -- by tests/host_osc.cpp, which runs it against the oracle from arbitrary phases.
#pragma once
#if defined(__HIPCC__)
#include "mxg_common.h"
#include "mxg_sincos.h"
#else
#include <math.h>
#include <stddef.h>
#include "../../include/maxigpu.h"
#include "mxg_sincos.h"
namespace mxg {
constexpr double kChandiv = 1.0;
}
#endif

namespace mxg {
namespace {

// Per-voice values that depend only on (frequency, p1, p2): hoisted out of the sample loop
// when the frequency is block-constant.  Each is the exact sub-expression of the reference.
struct OscPre {
    double inc;  // phase increment
    double k;    // sawn: 8820.22/frequency            (C:346)
    double p1;   // pulse: clamped duty (C:304-305); phasorBetween: startphase
    double p2;   // phasorBetween: endphase
    SinTabK sk;  // sinewave / coswave: two coefficients of sincos_tab in vector registers (mxg_sincos.h; a kernel's loop sets them opaque)
};

template <int WF>
__device__ __forceinline__ OscPre osc_pre(double f, double sr, double p1, double p2) {
    OscPre q;
    q.k = 0.0;
    q.p1 = p1;
    q.p2 = p2;
    q.sk = {1.0 / 120, 1.0 / 24};
    if constexpr (WF == MXG_OSC_SINEBUF) {
        q.inc = 512. / (sr / (f * kChandiv));  // C:269
    } else if constexpr (WF == MXG_OSC_SINEBUF4) {
        q.inc = 512. / (sr / (f));  // C:241
    } else if constexpr (WF == MXG_OSC_SAW) {
        q.inc = (1. / (sr / (f))) * 2.0;  // C:337
    } else if constexpr (WF == MXG_OSC_SAWN) {
        q.inc = (1. / (sr / (f)));  // C:345
        q.k = (8820.22 / f);        // C:346
    } else if constexpr (WF == MXG_OSC_PHASORBETWEEN) {
        q.inc = ((p2 - p1) / (sr / (f)));  // C:328
    } else if constexpr (WF == MXG_OSC_PULSE) {
        double duty = p1;
        if (duty < 0.) duty = 0;  // C:304
        if (duty > 1.) duty = 1;  // C:305
        q.p1 = duty;
        q.inc = (1. / (sr / (f)));  // C:307
    } else {
        q.inc = (1. / (sr / (f)));  // C:232, 280, 289, 297, 314, 365
    }
    return q;
}

// `if (phase >= 1.0) phase -= 1.0;` (C:231, C:279) as compare + ONE select + subtract: the subtrahend is 1.0 or +0.0, assembled from
// its high word (x - 0.0 is x for every x, -0.0 and NaN included), instead of a subtract and a two-word select.
// UNIT (the TRUST forms of sinewave / coswave): the caller guarantees +0.0 <= phase < 2, where the statement is the fractional part --
// v_fract_f64, one instruction: x - floor(x) is x itself below 1 and the exact x - 1 from 1 to 2 (the instruction's clamp to the double
// below 1.0 never acts on exact differences; a -0.0, which the subtraction would keep and fract would not, is excluded by the caller).
template <bool UNIT = false>
__device__ __forceinline__ double wrap_at_one(double phase) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (UNIT) return __builtin_amdgcn_fract(phase);
    return phase - __hiloint2double(phase >= 1.0 ? 0x3FF00000 : 0, 0);
#else
    return phase >= 1.0 ? phase - 1.0 : phase;
#endif
}

// `if (phase >= 511) phase -= 512;` (C:240, C:269) the same way: compare, one select of the subtrahend's high word (512.0 or +0.0), subtract.
// FAST = false: the plain statement as hipcc compiles it (subtract, compare, two-word select) -- K1's sinebuf keeps it: that kernel is
// bound by its store stream, and every leaner form of its loop measured SLOWER (profiles/r04_heavy_osc.md).
template <bool FAST>
__device__ __forceinline__ double wrap_at_511(double phase) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (FAST) return phase - __hiloint2double(phase >= 511 ? 0x40800000 : 0, 0);
#endif
    if (phase >= 511) phase -= 512;
    return phase;
}

// One sample of one voice.  `phase`/`hold` are the members `phase`/`output` (H:173,176).
// s_sine[i+1] == sineBuffer[i] (i=-1..513), s_trans[i] == transition[i] (i=0..1001); for sinewave / coswave s_sine is the
// sin / cos table MXG_SINTAB of mxg_sincos.h instead.
// TRUST (sinewave / coswave only): the caller has checked +0.0 <= inc < 1 and +0.0 <= phase < 2 (sign bits clear), which the
// recurrence then keeps: fract(phase) <= 1 - 2^-53 and inc <= 1 - 2^-53 add up to at most 2 - 2^-52, a double below 2.
#ifndef MXG_SB4_PAIRS
#define MXG_SB4_PAIRS 0  // 1: sinebuf4's four table values as two aligned 16-byte LDS reads from a parity copy of the table (A/B: measured
                        // SLOWER on the same device, 59.0 against 57.3 us for 65 536 x 512 -- the kernel is not bound by LDS cycles)
#endif
// Tick flavours (osc_tick's FL, a kernel's choice): bit 0 = sinebuf / sawn read their second table value from a copy of the table
// tab_copy doubles on (two ds_read_b64, ~7 LDS cycles each at unrelated addresses, where hipcc merges two reads off one base into a
// ds_read2_b64 of ~20); bit 1 = the three-instruction phase wrap (wrap_at_511<true>).  The VALU- and LDS-bound kernels take both
// (K1m, K1's sawn: 52 -> 46 us); K1's sinebuf takes neither (see wrap_at_511).
constexpr int kTickLean = 3;
template <int WF, int FL>
constexpr int tab_copy() {
    return (FL & 1) ? (WF == MXG_OSC_SINEBUF ? 520 : (WF == MXG_OSC_SAWN ? 1008 : 0)) : 0;
}
constexpr int kSb4Copy = 520;     // sinebuf4 on the device: four copies of the 515-entry table this far apart (osc_tick; osc.hip load_tab)
constexpr int kSineOddOff = 515;  // (odd and >= the table's 515 entries: element i of the second copy is 16-byte aligned for odd i; osc.hip asserts)
template <int WF, bool TRUST = false, int FL = 0>
__device__ __forceinline__ double osc_tick(double &phase, double &hold, const OscPre &q,
                                           const double *s_sine, const double *s_trans) {
    if constexpr (WF == MXG_OSC_SINEWAVE) {  // C:228-235
        double r = sin_2pi_phase<TRUST>(phase, s_sine, q.sk);
        hold = r;
        phase = wrap_at_one<TRUST>(phase);
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_COSWAVE) {  // C:276-283
        double r = cos_2pi_phase<TRUST>(phase, s_sine, q.sk);
        hold = r;
        phase = wrap_at_one<TRUST>(phase);
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_PHASOR) {  // C:285-291
        double r = phase;
        hold = r;
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_SAW) {  // C:333-340
        double r = phase;
        hold = r;
        if (phase >= 1.0) phase -= 2.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_TRIANGLE) {  // C:362-373
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        // (the branch of C:366-370 selects the operand, then ONE subtract-multiply: the same two operations on the taken side)
        const double t = (phase <= 0.5) ? phase : (1.0 - phase);
        const double r = (t - 0.25) * 4;
        hold = r;
        return r;
    } else if constexpr (WF == MXG_OSC_SQUARE) {  // C:293-300 (output held at phase==0.5)
        if (phase < 0.5) hold = -1;
        if (phase > 0.5) hold = 1;
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        return hold;
    } else if constexpr (WF == MXG_OSC_PULSE) {  // C:302-311 (output held at phase==duty)
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        // (measured, round 4: the sign of phase - duty copied onto 1.0, with the equal / NaN case behind a wave-level branch, is 3
        // instructions shorter and 25 % SLOWER -- 57.5 against 46.4 us at 65 536 voices: a scalar branch per voice and sample)
        if (phase < q.p1) hold = -1.;
        if (phase > q.p1) hold = 1.;
        return hold;
    } else if constexpr (WF == MXG_OSC_IMPULSE) {  // C:312-319 (member `output` untouched)
        if (phase >= 1.0) phase -= 1.0;
        double r = phase < q.inc ? 1.0 : 0.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_PHASORBETWEEN) {  // C:321-330
        double r = phase;
        hold = r;
        if (phase < q.p1) phase = q.p1;
        if (phase >= q.p2) phase = q.p1;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_SINEBUF) {  // C:266-274
        phase += q.inc;
        phase = wrap_at_511<(FL & 2) != 0>(phase);
        double remainder = phase - floor(phase);
        int i = (int)phase;  // (long)phase: truncation toward zero; |phase| < 2^31 here
        double r = (1 - remainder) * s_sine[1 + i + 1] + remainder * s_sine[2 + i + 1 + tab_copy<WF, FL>()];
        hold = r;
        return r;
    } else if constexpr (WF == MXG_OSC_SINEBUF4) {  // C:237-264
        phase += q.inc;
        phase = wrap_at_511<true>(phase);
        double remainder = phase - floor(phase);
        int i = (int)phase;
#if defined(__HIP_DEVICE_COMPILE__) && MXG_SB4_PAIRS
        // a, b, c, d = s_sine[i .. i + 3]: two aligned 16-byte LDS reads from the copy of the table in which element i starts a
        // 16-byte slot (even i: the table itself; odd i: the second copy, kSineOddOff elements on -- see load_tab) instead of four
        // 8-byte ones; phase == 0 reads sineBuffer[512] for `a` (C:245-256; index -1 is the 0.0 guard s_sine[0])
        typedef double pair2 __attribute__((ext_vector_type(2)));
        const double *t = s_sine + i + (i & 1) * kSineOddOff;
        const pair2 ab = *reinterpret_cast<const pair2 *>(t), cd = *reinterpret_cast<const pair2 *>(t + 2);
        double a = ab.x;
        if (__builtin_expect(phase == 0, 0)) a = s_sine[513];
        const double b = ab.y, c = cd.x, d = cd.y;
#elif defined(__HIP_DEVICE_COMPILE__)
        // a, b, c, d = sineBuffer[i-1 .. i+2] = s_sine[i .. i+3] from ONE address; `phase == 0` -- a voice at rest: frequency 0 -- reads
        // sineBuffer[512] for `a` instead (C:245-256; index -1 is the 0.0 guard s_sine[0]): a select of the VALUE after the read (as a
        // select of the INDEX it costs a second address on every sample of every voice).
        // Four ds_read_b64 (~7 LDS cycles each at 64 unrelated addresses), not the two ds_read2_b64 hipcc makes of four reads off one
        // base (~20 each: that instruction is served 16 lanes at a time over 32 banks, and the LDS pipe of a CU, shared by its four
        // SIMDs, would become the bound in place of the VALU): the kernels keep FOUR copies of the table, kSb4Copy doubles apart --
        // further than a ds_read2's offsets reach -- and element i + j comes from copy j.
        const double *t = s_sine + i;
        double a = t[0];
        const double b = t[kSb4Copy + 1], c = t[2 * kSb4Copy + 2], d = t[3 * kSb4Copy + 3];
#ifndef MXG_SB4_BRANCH
#define MXG_SB4_BRANCH 0  // A/B (tools/build_ab.sh): 1 = the rest case out of line behind a wave-level test -- two VALU instructions fewer
                          // and 1.5 % SLOWER (53.4 against 52.6 us): a scalar branch per sample costs more than it saves (pulse: 25 %)
#endif
#if MXG_SB4_BRANCH
        if (__builtin_expect(__any(phase == 0), 0)) {
            if (phase == 0) a = s_sine[513];
        }
#else
        a = (phase == 0) ? s_sine[513] : a;
#endif
#else
        int ia = (phase == 0) ? 512 : i - 1;  // C:245-256; index -1 is the 0.0 guard
        double a = s_sine[ia + 1];
        double b = s_sine[i + 1];
        double c = s_sine[i + 1 + 1];
        double d = s_sine[i + 2 + 1];
#endif
        double a1 = 0.5 * (c - a);
        double a2 = a - 2.5 * b + 2. * c - 0.5 * d;
        double a3 = 0.5 * (d - a) + 1.5 * (b - c);
        double r = ((a3 * remainder + a2) * remainder + a1) * remainder + b;
        hold = r;
        return r;
    } else {  // MXG_OSC_SAWN  C:342-359
        if (phase >= 0.5) phase -= 1.0;
        phase += q.inc;
        double temp = q.k * phase;
        if (temp < -0.5) temp = -0.5;
        if (temp > 0.5) temp = 0.5;
        temp *= 1000.0;
        temp += 500.0;
        double remainder = temp - floor(temp);
        int i = (int)temp;
        double r = ((1.0 - remainder) * s_trans[i] + remainder * s_trans[1 + i + tab_copy<WF, FL>()]) - phase;
        hold = r;
        return r;
    }
}

// K consecutive samples of one voice in three stages, for kernels that run ONE wavefront per SIMD: (1) the K phase steps, with index
// and remainder; (2) the table reads of all K samples; (3) the K interpolations.  The same operations per sample as osc_tick -- only
// operations of DIFFERENT samples change places, so the bits are osc_tick's -- but every LDS read of the chunk is in flight before the
// first result is needed, and a caller can put other work between (2) and (3) while the (bank-conflicted) reads drain.  Left to
// itself the scheduler finishes a sample before it starts the next to save registers, and a lone wavefront waits out the LDS
// latency K times.  Table forms with two reads per sample only: sinebuf (C:266-274) and sawn (C:342-359).
template <int WF>
constexpr bool osc_has_pipe() {
    return WF == MXG_OSC_SINEBUF || WF == MXG_OSC_SAWN;
}
template <int K>
struct OscPipe {
    int idx[K];
    double rem[K], aux[K];  // aux: sawn's phase (subtracted from the interpolated value)
    double t0[K], t1[K];
};
// (the K phase steps in two halves, for a caller that puts other instructions between them)
template <int WF, int K, int HALF>
__device__ __forceinline__ void osc_pipe_phase_half(double &phase, const OscPre &q, OscPipe<K> &p) {
#pragma unroll
    for (int i = HALF * (K / 2); i < (HALF + 1) * (K / 2); i++) {
        if constexpr (WF == MXG_OSC_SINEBUF) {
            phase += q.inc;
            phase = wrap_at_511<true>(phase);
            p.rem[i] = phase - floor(phase);
            p.idx[i] = (int)phase + 2;
        } else {
            if (phase >= 0.5) phase -= 1.0;
            phase += q.inc;
            double temp = q.k * phase;
            if (temp < -0.5) temp = -0.5;
            if (temp > 0.5) temp = 0.5;
            temp *= 1000.0;
            temp += 500.0;
            p.rem[i] = temp - floor(temp);
            p.idx[i] = (int)temp;
            p.aux[i] = phase;
        }
    }
}
template <int WF, int K>
__device__ __forceinline__ void osc_pipe_phase(double &phase, const OscPre &q, OscPipe<K> &p) {
    osc_pipe_phase_half<WF, K, 0>(phase, q, p);
    osc_pipe_phase_half<WF, K, 1>(phase, q, p);
}
// FL: the caller's table layout (bit 0: the second value comes from the table's copy, tab_copy; K1m's kernels) -- 0: one plain table
// (the per-voice tables of osctab.hip)
// FL bit 2 (a table without a copy, e.g. a voice's own): the second value is read at the first one's LDS address plus an 8 the optimizer
// cannot see through -- one more integer addition per sample, but two ds_read_b64 instead of one ds_read2_b64 (see tab_copy).
template <int WF, int K, int FL = kTickLean>
__device__ __forceinline__ void osc_pipe_fetch(OscPipe<K> &p, const double *s_tab) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr ((FL & 4) != 0) {
        typedef __attribute__((address_space(3))) const double lds_cd;
        unsigned eight = 8;
        asm volatile("" : "+v"(eight));
#pragma unroll
        for (int i = 0; i < K; i++) {
            lds_cd *a = (lds_cd *)(s_tab + p.idx[i]);  // (s_tab is LDS: the generic pointer's low 32 bits are the LDS address)
            p.t0[i] = *a;
            p.t1[i] = *(lds_cd *)(uintptr_t)((unsigned)(uintptr_t)a + eight);
        }
        return;
    }
#endif
    const double *s_next = s_tab + 1 + tab_copy<WF, FL>();
#pragma unroll
    for (int i = 0; i < K; i++) {
        p.t0[i] = s_tab[p.idx[i]];
        p.t1[i] = s_next[p.idx[i]];
    }
}
template <int WF, int K>
__device__ __forceinline__ void osc_pipe_finish(const OscPipe<K> &p, double (&r)[K], double &hold) {
#pragma unroll
    for (int i = 0; i < K; i++) {
        if constexpr (WF == MXG_OSC_SINEBUF)
            r[i] = (1 - p.rem[i]) * p.t0[i] + p.rem[i] * p.t1[i];
        else
            r[i] = ((1.0 - p.rem[i]) * p.t0[i] + p.rem[i] * p.t1[i]) - p.aux[i];
    }
    hold = r[K - 1];
}

// K samples at once where the waveform has the staged form (all LDS reads of the chunk in flight), tick by tick elsewhere
template <int WF, int K>
__device__ __forceinline__ void osc_tick_chunk(double &phase, double &hold, const OscPre &q, const double *s_sine,
                                               const double *s_trans, double (&r)[K]) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (osc_has_pipe<WF>()) {
        OscPipe<K> p;
        osc_pipe_phase<WF, K>(phase, q, p);
        __builtin_amdgcn_sched_barrier(0);
        osc_pipe_fetch<WF, K>(p, WF == MXG_OSC_SINEBUF ? s_sine : s_trans);
        __builtin_amdgcn_sched_barrier(0);
        osc_pipe_finish<WF, K>(p, r, hold);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < K; i++) r[i] = osc_tick<WF, false, kTickLean>(phase, hold, q, s_sine, s_trans);
}

// The recurrence of one sample WITHOUT its output: exactly the phase operations of osc_tick, in its order.  For the
// waveforms whose tick always overwrites `hold` it is left alone (a later tick sets it); square / pulse / impulse update
// `hold` conditionally, so they run their (cheap) tick.
template <int WF>
__device__ __forceinline__ void osc_skip(double &phase, double &hold, const OscPre &q, const double *s_sine,
                                         const double *s_trans) {
    if constexpr (WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE || WF == MXG_OSC_PHASOR || WF == MXG_OSC_TRIANGLE) {
        phase = wrap_at_one(phase);
        phase += q.inc;
    } else if constexpr (WF == MXG_OSC_SAW) {
        if (phase >= 1.0) phase -= 2.0;
        phase += q.inc;
    } else if constexpr (WF == MXG_OSC_PHASORBETWEEN) {
        if (phase < q.p1) phase = q.p1;
        if (phase >= q.p2) phase = q.p1;
        phase += q.inc;
    } else if constexpr (WF == MXG_OSC_SINEBUF || WF == MXG_OSC_SINEBUF4) {
        phase += q.inc;
        phase = wrap_at_511<true>(phase);
    } else if constexpr (WF == MXG_OSC_SAWN) {
        if (phase >= 0.5) phase -= 1.0;
        phase += q.inc;
    } else {
        (void)osc_tick<WF>(phase, hold, q, s_sine, s_trans);
    }
}

}  // namespace
}  // namespace mxg
