/* oracle/maxi_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the per-sample DSP hot path of micknoise/Maximilian
 * (the reference, mounted read-only at /root/reference; never copied).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the CHECKER -- the product (libmaxigpu.so) never links, loads or falls back
 * to it.
 *
 * PARITY PINNING: the reference's own tests hold no golden vectors / KATs for this
 * path.  This restatement is therefore pinned against outputs of
 * the reference itself, run in the build container: (1) oracle/_ref/libmaxiref.so
 * (the unmodified reference sources + oracle/ref_harness.cpp) is compared with this
 * file bit-for-bit on seeded inputs in tests/test_oracle_golden.py, and (2) fixtures
 * generated from _ref by oracle/gen_golden.py are committed under tests/golden/.
 *
 * Every function cites the reference lines it follows (C = src/maximilian.cpp,
 * H = src/maximilian.h, L/ = src/libs/).  Build flags are part of the contract:
 * -O2 -ffp-contract=off, no -march=native, no -ffast-math (oracle/Makefile).
 *
 * Layout convention: banks are rendered sample-major / voice-minor, out[n*V + v];
 * state is SoA, one array of length V per reference member.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "maxi_oracle_tables.h"

/* H:55-58 */
#define MX_PI 3.1415926535897932384626433832795
#define MX_TWOPI 6.283185307179586476925286766559

/* C:53  `float chandiv = 1;`  (promoted to double wherever it multiplies a double) */
static const float chandiv = 1;

/* C:57-59  maxiSettings statics (size_t!) */
static size_t g_sampleRate = 44100;
static size_t g_channels = 2;
static size_t g_bufferSize = 1024;

/* sineBuffer[i] for i = -1..513 ; transition[i] for i = 0..1001 (guards, see tables.h) */
#define SINEBUF(i) (MAXI_SINE_TAB[(i) + 1])
#define TRANSITION(i) (MAXI_TRANS_TAB[(i)])

const char *mxo_kind(void) { return "port"; }

/* H:138-143 maxiSettings::setup */
void mxo_settings(size_t sr, size_t ch, size_t buf) {
    g_sampleRate = sr;
    g_channels = ch;
    g_bufferSize = buf;
}

const double *mxo_sine_table(void) { return &MAXI_SINE_TAB[1]; }
const double *mxo_transition_table(void) { return &MAXI_TRANS_TAB[0]; }
double mxo_sine_table_guard(void) { return MAXI_SINE_TAB[0]; }
double mxo_transition_guard(void) { return MAXI_TRANS_TAB[1001]; }

/* ------------------------------------------------------------------------------------
 * maxiOsc (H:169-215, C:209-373).  State per voice: phase, output.
 * One call == one sample of one voice; returns the sample.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    double phase, output;
} osc_t;

/* C:228-235 */
static double osc_sinewave(osc_t *o, double frequency) {
    o->output = sin(o->phase * (MX_TWOPI));
    if (o->phase >= 1.0) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    return o->output;
}
/* C:276-283 */
static double osc_coswave(osc_t *o, double frequency) {
    o->output = cos(o->phase * (MX_TWOPI));
    if (o->phase >= 1.0) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    return o->output;
}
/* C:285-291 */
static double osc_phasor(osc_t *o, double frequency) {
    o->output = o->phase;
    if (o->phase >= 1.0) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    return o->output;
}
/* C:333-340 */
static double osc_saw(osc_t *o, double frequency) {
    o->output = o->phase;
    if (o->phase >= 1.0) o->phase -= 2.0;
    o->phase += (1. / (g_sampleRate / (frequency))) * 2.0;
    return o->output;
}
/* C:362-373 */
static double osc_triangle(osc_t *o, double frequency) {
    if (o->phase >= 1.0) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    if (o->phase <= 0.5) {
        o->output = (o->phase - 0.25) * 4;
    } else {
        o->output = ((1.0 - o->phase) - 0.25) * 4;
    }
    return o->output;
}
/* C:293-300 : output is HELD when phase == 0.5 exactly */
static double osc_square(osc_t *o, double frequency) {
    if (o->phase < 0.5) o->output = -1;
    if (o->phase > 0.5) o->output = 1;
    if (o->phase >= 1.0) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    return o->output;
}
/* C:302-311 : output is HELD when phase == duty exactly */
static double osc_pulse(osc_t *o, double frequency, double duty) {
    if (duty < 0.) duty = 0;
    if (duty > 1.) duty = 1;
    if (o->phase >= 1.0) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    if (o->phase < duty) o->output = -1.;
    if (o->phase > duty) o->output = 1.;
    return o->output;
}
/* C:312-319 : uses a LOCAL `output`, the member is left untouched */
static double osc_impulse(osc_t *o, double frequency) {
    if (o->phase >= 1.0) o->phase -= 1.0;
    double phaseInc = (1. / (g_sampleRate / (frequency)));
    double output = o->phase < phaseInc ? 1.0 : 0.0;
    o->phase += phaseInc;
    return output;
}
/* C:321-330 */
static double osc_phasorBetween(osc_t *o, double frequency, double startphase, double endphase) {
    o->output = o->phase;
    if (o->phase < startphase) {
        o->phase = startphase;
    }
    if (o->phase >= endphase) o->phase = startphase;
    o->phase += ((endphase - startphase) / (g_sampleRate / (frequency)));
    return o->output;
}
/* C:266-274 : wrap at 511 (not 512); table indices offset by +1/+2; (long) truncates
 * toward zero so phase in (-1,0) indexes 0. */
static double osc_sinebuf(osc_t *o, double frequency) {
    double remainder;
    o->phase += 512. / (g_sampleRate / (frequency * chandiv));
    if (o->phase >= 511) o->phase -= 512;
    remainder = o->phase - floor(o->phase);
    o->output = (double)((1 - remainder) * SINEBUF(1 + (long)o->phase) +
                         remainder * SINEBUF(2 + (long)o->phase));
    return o->output;
}
/* C:237-264 : 4-point interpolation; index (long)phase-1 is -1 for phase in (-1,1)\{0}
 * (out of bounds in the reference; our table carries a 0.0 guard there). */
static double osc_sinebuf4(osc_t *o, double frequency) {
    double remainder;
    double a, b, c, d, a1, a2, a3;
    o->phase += 512. / (g_sampleRate / (frequency));
    if (o->phase >= 511) o->phase -= 512;
    remainder = o->phase - floor(o->phase);
    if (o->phase == 0) {
        a = SINEBUF((long)512);
        b = SINEBUF((long)o->phase);
        c = SINEBUF((long)o->phase + 1);
        d = SINEBUF((long)o->phase + 2);
    } else {
        a = SINEBUF((long)o->phase - 1);
        b = SINEBUF((long)o->phase);
        c = SINEBUF((long)o->phase + 1);
        d = SINEBUF((long)o->phase + 2);
    }
    a1 = 0.5f * (c - a);
    a2 = a - 2.5 * b + 2.f * c - 0.5f * d;
    a3 = 0.5f * (d - a) + 1.5f * (b - c);
    o->output = (double)(((a3 * remainder + a2) * remainder + a1) * remainder + b);
    return o->output;
}
/* C:342-359 : band-limited saw through transition[]; index 1001 is read (x remainder 0)
 * when temp clamps to exactly +0.5 (guard 0.0). */
static double osc_sawn(osc_t *o, double frequency) {
    if (o->phase >= 0.5) o->phase -= 1.0;
    o->phase += (1. / (g_sampleRate / (frequency)));
    double temp = (8820.22 / frequency) * o->phase;
    if (temp < -0.5) {
        temp = -0.5;
    }
    if (temp > 0.5) {
        temp = 0.5;
    }
    temp *= 1000.0f;
    temp += 500.0f;
    double remainder = temp - floor(temp);
    o->output = (double)((1.0f - remainder) * TRANSITION((long)temp) +
                         remainder * TRANSITION(1 + (long)temp)) -
                o->phase;
    return o->output;
}

static double osc_tick(int wf, osc_t *o, double f, double p1, double p2) {
    switch (wf) {
        case 0: return osc_sinewave(o, f);
        case 1: return osc_coswave(o, f);
        case 2: return osc_phasor(o, f);
        case 3: return osc_saw(o, f);
        case 4: return osc_triangle(o, f);
        case 5: return osc_square(o, f);
        case 6: return osc_pulse(o, f, p1);
        case 7: return osc_impulse(o, f);
        case 8: return osc_sinebuf(o, f);
        case 9: return osc_sinebuf4(o, f);
        case 10: return osc_sawn(o, f);
        case 11: return osc_phasorBetween(o, f, p1, p2);
    }
    return 0.0;
}

int mxo_osc(int wf, size_t V, size_t N, const double *freq, int fps, const double *p1,
            const double *p2, double *phase, double *outhold, double *out) {
    if (wf < 0 || wf > 11) return -1;
    for (size_t v = 0; v < V; v++) {
        osc_t o = {phase[v], outhold[v]};
        double a = p1 ? p1[v] : 0.0, b = p2 ? p2[v] : 0.0;
        for (size_t n = 0; n < N; n++) {
            double f = fps ? freq[n * V + v] : freq[v];
            out[n * V + v] = osc_tick(wf, &o, f, a, b);
        }
        phase[v] = o.phase;
        outhold[v] = o.output;
    }
    return 0;
}

/* EXTENSION (no reference counterpart; the checker of mxg_osc_render_tables): maxiOsc::sinebuf, C:266-274, with the voice's own
 * 514-entry table T_v = tables + 514 v in the place of sineBuffer -- the statements of osc_sinebuf above, the table apart.  Pinned by
 * construction: with every T_v = sineBuffer it must return mxo_osc(8, ...)'s bits (tests/test_extra_cpu.py), and those are the
 * reference's (tests/test_oracle_golden.py). */
int mxo_osc_tables(size_t V, size_t N, const double *freq, const double *tables, double *phase, double *outhold, double *out) {
    for (size_t v = 0; v < V; v++) {
        const double *T = tables + 514 * v;
        double ph = phase[v], o = outhold[v];
        for (size_t n = 0; n < N; n++) {
            double remainder;
            ph += 512. / (g_sampleRate / (freq[v] * chandiv));
            if (ph >= 511) ph -= 512;
            remainder = ph - floor(ph);
            o = (double)((1 - remainder) * T[1 + (long)ph] + remainder * T[2 + (long)ph]);
            out[n * V + v] = o;
        }
        phase[v] = ph;
        outhold[v] = o;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiFilter (H:289-366, C:442-500, ctor C:1517: x,y,z,c = 0; outputs[] treated as 0).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    double x, y, o0, o1, o2;
} flt_t;

/* C:457-462 : the coefficient part of lores/hires */
static void lores_coeffs(double cutoff, double resonance, double *c, double *r) {
    if (cutoff < 10) cutoff = 10;
    if (cutoff > (g_sampleRate)) cutoff = (g_sampleRate);
    if (resonance < 1.) resonance = 1.;
    double z = cos(MX_TWOPI * cutoff / g_sampleRate);
    *c = 2 - 2 * z;
    *r = (sqrt(2.0) * sqrt(-pow((z - 1.0), 3.0)) + resonance * (z - 1)) / (resonance * (z - 1));
}
/* C:455-468 */
static double flt_lores(flt_t *f, double input, double cutoff1, double resonance) {
    double c, r;
    lores_coeffs(cutoff1, resonance, &c, &r);
    f->x = f->x + (input - f->y) * c;
    f->y = f->y + f->x;
    f->x = f->x * r;
    return f->y;
}
/* C:471-484 */
static double flt_hires(flt_t *f, double input, double cutoff1, double resonance) {
    double c, r;
    lores_coeffs(cutoff1, resonance, &c, &r);
    f->x = f->x + (input - f->y) * c;
    f->y = f->y + f->x;
    f->x = f->x * r;
    return input - f->y;
}
/* C:487-500 */
static double flt_bandpass(flt_t *f, double input, double cutoff1, double resonance) {
    double cutoff = cutoff1;
    if (cutoff > (g_sampleRate * 0.5)) cutoff = (g_sampleRate * 0.5);
    if (resonance >= 1.) resonance = 0.999999;
    double z = cos(MX_TWOPI * cutoff / g_sampleRate);
    double i0 = (1 - resonance) * (sqrt(resonance * (resonance - 4.0 * pow(z, 2.0) + 2.0) + 1));
    double i1 = 2 * z * resonance;
    double i2 = pow((resonance * -1), 2);
    double output = i0 * input + i1 * f->o1 + i2 * f->o2;
    f->o2 = f->o1;
    f->o1 = output;
    return output;
}
/* C:442-446 */
static double flt_lopass(flt_t *f, double input, double cutoff) {
    double output = f->o0 + cutoff * (input - f->o0);
    f->o0 = output;
    return output;
}
/* C:449-453 (sic: stores the high-passed value back into outputs[0]) */
static double flt_hipass(flt_t *f, double input, double cutoff) {
    double output = input - (f->o0 + cutoff * (input - f->o0));
    f->o0 = output;
    return output;
}

int mxo_filter(int kind, size_t V, size_t N, const double *in, const double *cutoff, int cps,
               const double *res, int rps, double *st, double *out) {
    if (kind < 0 || kind > 4) return -1;
    for (size_t v = 0; v < V; v++) {
        flt_t f = {st[0 * V + v], st[1 * V + v], st[2 * V + v], st[3 * V + v], st[4 * V + v]};
        for (size_t n = 0; n < N; n++) {
            double x = in[n * V + v];
            double c = cps ? cutoff[n * V + v] : cutoff[v];
            double r = res ? (rps ? res[n * V + v] : res[v]) : 0.0;
            double o = 0;
            switch (kind) {
                case 0: o = flt_lores(&f, x, c, r); break;
                case 1: o = flt_hires(&f, x, c, r); break;
                case 2: o = flt_bandpass(&f, x, c, r); break;
                case 3: o = flt_lopass(&f, x, c); break;
                case 4: o = flt_hipass(&f, x, c); break;
            }
            out[n * V + v] = o;
        }
        st[0 * V + v] = f.x;
        st[1 * V + v] = f.y;
        st[2 * V + v] = f.o0;
        st[3 * V + v] = f.o1;
        st[4 * V + v] = f.o2;
    }
    return 0;
}

/* Host-side coefficients of the hoisted path, [3][V]: lores/hires -> c, r, 0 (C:459-461);
 * bandpass -> inputs[0..2] (C:492-495). */
void mxo_filter_coeffs(int kind, size_t V, const double *cutoff, const double *res, double *coef) {
    for (size_t v = 0; v < V; v++) {
        if (kind == 2) {
            double cu = cutoff[v], resonance = res[v];
            if (cu > (g_sampleRate * 0.5)) cu = (g_sampleRate * 0.5);
            if (resonance >= 1.) resonance = 0.999999;
            double z = cos(MX_TWOPI * cu / g_sampleRate);
            coef[v] = (1 - resonance) * (sqrt(resonance * (resonance - 4.0 * pow(z, 2.0) + 2.0) + 1));
            coef[V + v] = 2 * z * resonance;
            coef[2 * V + v] = pow((resonance * -1), 2);
        } else {
            lores_coeffs(cutoff[v], res[v], &coef[v], &coef[V + v]);
            coef[2 * V + v] = 0.0;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * maxiEnv (H:888-932, C:1319-1494).  No constructor: all state starts 0, holdtime = 1.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    double attack, decay, sustain, release, amplitude, output;
    long holdtime, holdcount;
    int attackphase, decayphase, sustainphase, holdphase, releasephase;
} env_t;

/* C:1415-1466 (identical body to the 7-argument overload C:1362-1413) */
static double env_adsr(env_t *e, double input, int trigger) {
    if (trigger == 1 && e->attackphase != 1 && e->holdphase != 1 && e->decayphase != 1) {
        e->holdcount = 0;
        e->decayphase = 0;
        e->sustainphase = 0;
        e->releasephase = 0;
        e->attackphase = 1;
    }
    if (e->attackphase == 1) {
        e->releasephase = 0;
        e->amplitude += (1 * e->attack);
        e->output = input * e->amplitude;
        if (e->amplitude >= 1) {
            e->amplitude = 1;
            e->attackphase = 0;
            e->decayphase = 1;
        }
    }
    if (e->decayphase == 1) {
        e->output = input * (e->amplitude *= e->decay);
        if (e->amplitude <= e->sustain) {
            e->decayphase = 0;
            e->holdphase = 1;
        }
    }
    if (e->holdcount < e->holdtime && e->holdphase == 1) {
        e->output = input * e->amplitude;
        e->holdcount++;
    }
    if (e->holdcount >= e->holdtime && trigger == 1) {
        e->output = input * e->amplitude;
    }
    if (e->holdcount >= e->holdtime && trigger != 1) {
        e->holdphase = 0;
        e->releasephase = 1;
    }
    if (e->releasephase == 1 && e->amplitude > 0.) {
        e->output = input * (e->amplitude *= e->release);
    }
    return e->output;
}
/* C:1319-1358 */
static double env_ar(env_t *e, double input, double attack, double release, long holdtime,
                     int trigger) {
    if (trigger == 1 && e->attackphase != 1 && e->holdphase != 1) {
        e->holdcount = 0;
        e->releasephase = 0;
        e->attackphase = 1;
    }
    if (e->attackphase == 1) {
        e->amplitude += (1 * attack);
        e->output = input * e->amplitude;
    }
    if (e->amplitude >= 1) {
        e->amplitude = 1;
        e->attackphase = 0;
        e->holdphase = 1;
    }
    if (e->holdcount < holdtime && e->holdphase == 1) {
        e->output = input;
        e->holdcount++;
    }
    if (e->holdcount == holdtime && trigger == 1) {
        e->output = input;
    }
    if (e->holdcount == holdtime && trigger != 1) {
        e->holdphase = 0;
        e->releasephase = 1;
    }
    if (e->releasephase == 1 && e->amplitude > 0.) {
        e->output = input * (e->amplitude *= release);
    }
    return e->output;
}

static void env_load(env_t *e, size_t V, size_t v, const double *par, const int64_t *holdtime,
                     const double *dst, const int64_t *ist) {
    e->attack = par[0 * V + v];
    e->decay = par[1 * V + v];
    e->sustain = par[2 * V + v];
    e->release = par[3 * V + v];
    e->holdtime = (long)holdtime[v];
    e->amplitude = dst[0 * V + v];
    e->output = dst[1 * V + v];
    e->holdcount = (long)ist[0 * V + v];
    e->attackphase = (int)ist[1 * V + v];
    e->decayphase = (int)ist[2 * V + v];
    e->sustainphase = (int)ist[3 * V + v];
    e->holdphase = (int)ist[4 * V + v];
    e->releasephase = (int)ist[5 * V + v];
}
static void env_store(const env_t *e, size_t V, size_t v, double *dst, int64_t *ist) {
    dst[0 * V + v] = e->amplitude;
    dst[1 * V + v] = e->output;
    ist[0 * V + v] = e->holdcount;
    ist[1 * V + v] = e->attackphase;
    ist[2 * V + v] = e->decayphase;
    ist[3 * V + v] = e->sustainphase;
    ist[4 * V + v] = e->holdphase;
    ist[5 * V + v] = e->releasephase;
}

int mxo_env(int mode, size_t V, size_t N, const double *in, const int32_t *trig, int tpv,
            const double *par, const int64_t *holdtime, double *dst, int64_t *ist, double *out) {
    if (mode < 0 || mode > 1) return -1;
    for (size_t v = 0; v < V; v++) {
        env_t e;
        env_load(&e, V, v, par, holdtime, dst, ist);
        for (size_t n = 0; n < N; n++) {
            double x = in ? in[n * V + v] : 1.0;
            int t = tpv ? trig[n * V + v] : trig[n];
            out[n * V + v] = mode == 0 ? env_adsr(&e, x, t)
                                       : env_ar(&e, x, e.attack, e.release, e.holdtime, t);
        }
        env_store(&e, V, v, dst, ist);
    }
    return 0;
}

/* C:1469-1494 setters.  which: 0 setAttack 1 setDecay 2 setRelease 3 setAttackMS */
double mxo_env_coeff(int which, double ms) {
    switch (which) {
        case 0: return 1 - pow(0.01, 1.0 / (ms * g_sampleRate * 0.001));
        case 1: return pow(0.01, 1.0 / (ms * g_sampleRate * 0.001));
        case 2: return pow(0.01, 1.0 / (ms * g_sampleRate * 0.001));
        case 3: return 1.0 / (ms / 1000.0 * g_sampleRate);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Fused subtractive voice (config 3): composition of the three restatements above in the
 * call order of cpp/commandline/maximilian_examples/14.monosynth/main.cpp:50-55 (mode 1)
 * or osc -> filter -> envelope (mode 0).
 * ------------------------------------------------------------------------------------ */
int mxo_voice(int mode, size_t V, size_t N, const double *freq, const double *cutoff,
              const double *res, const int32_t *trig, int tpv, const double *par,
              const int64_t *holdtime, double *ost, double *fst, double *dst, int64_t *ist,
              double *out) {
    if (mode < 0 || mode > 1) return -1;
    for (size_t v = 0; v < V; v++) {
        osc_t o = {ost[v], ost[V + v]};
        flt_t f = {fst[0 * V + v], fst[1 * V + v], fst[2 * V + v], fst[3 * V + v], fst[4 * V + v]};
        env_t e;
        env_load(&e, V, v, par, holdtime, dst, ist);
        for (size_t n = 0; n < N; n++) {
            int t = tpv ? trig[n * V + v] : trig[n];
            double r;
            if (mode == 0) {
                double s = osc_saw(&o, freq[v]);
                double y = flt_lores(&f, s, cutoff[v], res[v]);
                r = env_adsr(&e, y, t);
            } else {
                double a = env_adsr(&e, 1.0, t);
                double s = osc_saw(&o, freq[v]);
                double y = flt_lores(&f, s, a * cutoff[v], res[v]);
                r = y * a;
            }
            out[n * V + v] = r;
        }
        ost[v] = o.phase;
        ost[V + v] = o.output;
        fst[0 * V + v] = f.x;
        fst[1 * V + v] = f.y;
        fst[2 * V + v] = f.o0;
        fst[3 * V + v] = f.o1;
        fst[4 * V + v] = f.o2;
        env_store(&e, V, v, dst, ist);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiMix::stereo (C:503-509) + the user-side running sum over voices.
 * ------------------------------------------------------------------------------------ */
int mxo_mix_stereo(size_t V, size_t N, const double *in, const double *pan, double *mix) {
    for (size_t n = 0; n < N; n++) {
        double l = 0, r = 0;
        for (size_t v = 0; v < V; v++) {
            double x = pan[v];
            if (x > 1) x = 1;
            if (x < 0) x = 0;
            double input = in[n * V + v];
            l += input * sqrt(1.0 - x);
            r += input * sqrt(x);
        }
        mix[2 * n] = l;
        mix[2 * n + 1] = r;
    }
    return 0;
}

/* CPU-baseline timer (single thread only in the C port; the reference harness has the
 * threaded one).  Returns seconds for N samples x V voices of waveform wf. */
#include <time.h>
double mxo_time_osc(int wf, size_t V, size_t N, const double *freq, int threads, double *sink) {
    (void)threads;
    osc_t *bank = (osc_t *)calloc(V, sizeof(osc_t));
    double *row = (double *)malloc(V * sizeof(double));
    struct timespec t0, t1;
    double a = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t n = 0; n < N; n++) {
        for (size_t v = 0; v < V; v++) row[v] = osc_tick(wf, &bank[v], freq[v], 0.5, 1.0);
        a += row[n % V];
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (sink) *sink = a;
    free(bank);
    free(row);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------
 * maxiDelayline (H:266-284, C:415-439).  memory[] is 88200*8 doubles per instance in the
 * reference; here the ring of a bank is mem[slot*V + v], slot < cap (cap >= every size).
 * `phase` is an int (unset by the ctor; static-storage objects start at 0).
 * ------------------------------------------------------------------------------------ */
/* mode 0: dl (C:420-429)   mode 1: dlFromPosition (C:431-439) */
int mxo_delay(int mode, size_t V, size_t N, const double *in, const int32_t *size,
              const double *feedback, const int32_t *position, double *mem, size_t cap,
              int32_t *phase, double *out) {
    if (mode < 0 || mode > 1) return -1;
    for (size_t v = 0; v < V; v++) {
        int ph = phase[v];
        const int sz = size[v];
        const double fb = feedback[v];
        if ((size_t)sz > cap) return -2;
        for (size_t n = 0; n < N; n++) {
            double input = in[n * V + v];
            double output;
            if (mode == 0) {
                if (ph >= sz) {
                    ph = 0;
                }
                output = mem[(size_t)ph * V + v];
                mem[(size_t)ph * V + v] = (mem[(size_t)ph * V + v] * fb) + (input * fb) * 0.5;
                ph += 1;
            } else {
                int pos = position[v];
                if (ph >= sz) ph = 0;
                if (pos >= sz) pos = 0;
                output = mem[(size_t)pos * V + v];
                mem[(size_t)ph * V + v] = (mem[(size_t)ph * V + v] * fb) + (input * fb) * chandiv;
                ph += 1;
            }
            out[n * V + v] = output;
        }
        phase[v] = ph;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiSample play family (H:602-783, C:740-1075).  One shared buffer `amp` of `len`
 * doubles for the whole bank (the way maxiGrains alias one maxiSample, L/maxiGrains.h:162).
 * Guards: the reference reads amplitudes[len], [len+1] (playAtSpeed family, C:1063-1064) and
 * [-1] (play4 with position in (0,1), C:898) out of bounds; `amp` must therefore be valid on
 * [-1, len+1] with those three elements 0.0 (the parity convention).
 * State per voice: position.  mySampleRate is the int member (44100 after setSample, H:676).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const double *amp;
    size_t len;
    int mySampleRate;
    double position;
} smp_t;

/* C:740-747 */
static double smp_play(smp_t *s) {
    double output = s->amp[(long)s->position];
    s->position++;
    if ((size_t)(long)s->position >= s->len) {
        s->position = 0;
    }
    return output;
}
/* C:982-991 */
static double smp_playOnce(smp_t *s) {
    double output;
    if ((size_t)(long)s->position < s->len)
        output = s->amp[(long)s->position];
    else {
        output = 0;
    }
    s->position++;
    return output;
}
/* C:960-967 */
static double smp_playLoop(smp_t *s, double start, double end) {
    s->position++;
    size_t sampleLength = s->len;
    if (s->position < sampleLength * start) s->position = sampleLength * start;
    if ((long)s->position >= sampleLength * end) s->position = sampleLength * start;
    return s->amp[(long)s->position];
}
/* C:969-978 */
static double smp_playUntil(smp_t *s, double end) {
    double output;
    s->position++;
    if (end > 1.0) end = 1.0;
    if ((long)s->position < s->len * end)
        output = s->amp[(long)s->position];
    else {
        output = 0;
    }
    return output;
}
/* C:1060-1075 : sr/mySampleRate is an INTEGER division (size_t / int) */
static double smp_playAtSpeed(smp_t *s, double speed) {
    double output;
    double remainder = s->position - (long)s->position;
    if ((size_t)(long)s->position < s->len) {
        output = ((1 - remainder) * s->amp[1 + (long)s->position] +
                  remainder * s->amp[2 + (long)s->position]);
    } else {
        output = 0;
    }
    s->position = s->position + ((speed * chandiv) / (g_sampleRate / (size_t)s->mySampleRate));
    if ((size_t)(long)s->position >= s->len) {
        s->position -= s->len;
    }
    return output;
}
/* C:994-1003 */
static double smp_playOnceAtSpeed(smp_t *s, double speed) {
    double output;
    double remainder = s->position - (long)s->position;
    if ((size_t)((long)s->position + 1) < s->len)
        output = ((1 - remainder) * s->amp[(long)s->position] +
                  remainder * s->amp[1 + (long)s->position]);
    else
        output = 0;
    s->position = s->position + ((speed * chandiv) / (g_sampleRate / (size_t)s->mySampleRate));
    return output;
}
/* C:1047-1058 */
static double smp_playUntilAtSpeed(smp_t *s, double end, double speed) {
    double output;
    double remainder = s->position - (long)s->position;
    if (end > 1.0) end = 1.0;
    if ((long)s->position < s->len * end)
        output = ((1 - remainder) * s->amp[1 + (long)s->position] +
                  remainder * s->amp[2 + (long)s->position]);
    else
        output = 0;
    s->position = s->position + ((speed * chandiv) / (g_sampleRate / (size_t)s->mySampleRate));
    return output;
}
/* C:884-956 : 4-point interpolated looped read; start/end in SAMPLES */
static double smp_play4(smp_t *s, double frequency, double start, double end) {
    double remainder;
    double a, b, c, d, a1, a2, a3;
    double output;
    if (frequency > 0.) {
        if (s->position < start) {
            s->position = start;
        }
        if (s->position >= end) s->position = start;
        s->position += ((end - start) / (g_sampleRate / (frequency * chandiv)));
        remainder = s->position - floor(s->position);
        if (s->position > 0) {
            a = s->amp[(int)(floor(s->position)) - 1];
        } else {
            a = s->amp[0];
        }
        b = s->amp[(long)s->position];
        if (s->position < end - 2) {
            c = s->amp[(long)s->position + 1];
        } else {
            c = s->amp[0];
        }
        if (s->position < end - 3) {
            d = s->amp[(long)s->position + 2];
        } else {
            d = s->amp[0];
        }
        a1 = 0.5f * (c - a);
        a2 = a - 2.5 * b + 2.f * c - 0.5f * d;
        a3 = 0.5f * (d - a) + 1.5f * (b - c);
        output = (((a3 * remainder + a2) * remainder + a1) * remainder + b);
    } else {
        frequency *= -1.;
        if (s->position <= start) s->position = end;
        s->position -= ((end - start) / (g_sampleRate / (frequency * chandiv)));
        remainder = s->position - floor(s->position);
        if (s->position > start && s->position < end - 1) {
            a = s->amp[(long)s->position + 1];
        } else {
            a = s->amp[0];
        }
        b = s->amp[(long)s->position];
        if (s->position > start) {
            c = s->amp[(long)s->position - 1];
        } else {
            c = s->amp[0];
        }
        if (s->position > start + 1) {
            d = s->amp[(long)s->position - 2];
        } else {
            d = s->amp[0];
        }
        a1 = 0.5f * (c - a);
        a2 = a - 2.5 * b + 2.f * c - 0.5f * d;
        a3 = 0.5f * (d - a) + 1.5f * (b - c);
        output = (((a3 * remainder + a2) * -remainder + a1) * -remainder + b);
    }
    return output;
}
/* C:823-880 : playAtSpeedBetweenPoints passes `position` BY VALUE (C:824) -- the member is
 * never advanced, every call recomputes from the same position.  Replicated as is. */
static double smp_playAtSpeedBetweenPoints(smp_t *s, double frequency, double start, double end) {
    double pos = s->position;
    double remainder, output;
    size_t amplen = s->len;
    if (end >= amplen) end = amplen - 1;
    long a, b;
    if (frequency > 0.) {
        if (pos < start) {
            pos = start;
        }
        if (pos >= end) pos = start;
        pos += ((end - start) / ((g_sampleRate) / (frequency * chandiv)));
        remainder = pos - floor(pos);
        long posl = floor(pos);
        if ((size_t)(posl + 1) < amplen) {
            a = posl + 1;
        } else {
            a = posl - 1;
        }
        if ((size_t)(posl + 2) < amplen) {
            b = posl + 2;
        } else {
            b = amplen - 1;
        }
        output = ((1 - remainder) * s->amp[a] + remainder * s->amp[b]);
    } else {
        frequency *= -1.;
        if (pos <= start) pos = end;
        pos -= ((end - start) / (g_sampleRate / (frequency * chandiv)));
        remainder = pos - floor(pos);
        long posl = floor(pos);
        if (posl - 1 >= 0) {
            a = posl - 1;
        } else {
            a = 0;
        }
        if (posl - 2 >= 0) {
            b = posl - 2;
        } else {
            b = 0;
        }
        output = ((-1 - remainder) * s->amp[a] + remainder * s->amp[b]);
    }
    return output;
}

/* mode: 0 play 1 playOnce 2 playLoop(start,end) 3 playUntil(end) 4 playAtSpeed(a)
 * 5 playOnceAtSpeed(a) 6 playUntilAtSpeed(end,a) 7 play4(a=frequency,start,end)
 * 8 playAtSpeedBetweenPoints(a=frequency,start,end).  a is [V] or [N][V] (aps). */
int mxo_sample(int mode, size_t V, size_t N, const double *amp, size_t len, int mySampleRate,
               const double *a, int aps, const double *start, const double *end, double *position,
               double *out) {
    if (mode < 0 || mode > 8) return -1;
    for (size_t v = 0; v < V; v++) {
        smp_t s = {amp, len, mySampleRate, position[v]};
        double st = start ? start[v] : 0.0, en = end ? end[v] : 1.0;
        for (size_t n = 0; n < N; n++) {
            double x = a ? (aps ? a[n * V + v] : a[v]) : 1.0;
            double o = 0;
            switch (mode) {
                case 0: o = smp_play(&s); break;
                case 1: o = smp_playOnce(&s); break;
                case 2: o = smp_playLoop(&s, st, en); break;
                case 3: o = smp_playUntil(&s, en); break;
                case 4: o = smp_playAtSpeed(&s, x); break;
                case 5: o = smp_playOnceAtSpeed(&s, x); break;
                case 6: o = smp_playUntilAtSpeed(&s, en, x); break;
                case 7: o = smp_play4(&s, x, st, en); break;
                case 8: o = smp_playAtSpeedBetweenPoints(&s, x, st, en); break;
            }
            out[n * V + v] = o;
        }
        position[v] = s.position;
    }
    return 0;
}

/* maxiTrigger::onZX (H:569-579): previousValue starts at 1, firstTrigger at 1 (H:593-594). */
typedef struct {
    double previousValue;
    int firstTrigger;
} zx_t;

static double zx_onZX(zx_t *z, double input) {
    double isZX = 0.0;
    if ((z->previousValue <= 0.0 || z->firstTrigger) && input > 0) isZX = 1.0;
    z->previousValue = input;
    z->firstTrigger = 0;
    return isZX;
}

/* maxiSample::setPosition C:749-751 (maxiMap::clamp H:843-854) */
static void smp_setPosition(smp_t *s, double newPos) {
    double c = newPos;
    if (c > 1.0) c = 1.0;
    else if (c < 0.0) c = 0.0;
    s->position = c * (double)s->len;
}

/* C:1006-1042.  trigger() (C:597-600) zeroes position. */
int mxo_sample_zx(int mode, size_t V, size_t N, const double *amp, size_t len, int mySampleRate,
                  const double *trig, const double *a, int aps, const double *p0, const double *p1,
                  double *position, double *zx_prev, int32_t *zx_first, double *out) {
    if (mode < 0 || mode > 4) return -1;
    for (size_t v = 0; v < V; v++) {
        smp_t s = {amp, len, mySampleRate, position[v]};
        zx_t z = {zx_prev[v], zx_first[v] != 0};
        for (size_t n = 0; n < N; n++) {
            const double t = trig[n * V + v];
            const double x = a ? (aps ? a[n * V + v] : a[v]) : 1.0;
            double o = 0;
            if (zx_onZX(&z, t) != 0.0) {
                if (mode == 4) {
                    smp_setPosition(&s, p0[v]); /* C:1038-1040 */
                } else {
                    s.position = 0; /* trigger() */
                    if (mode == 2 || mode == 3) s.position = p0[v] * (double)len; /* C:1024, C:1032 */
                }
            }
            switch (mode) {
                case 0: o = smp_playOnce(&s); break;                             /* C:1010 */
                case 1: case 2: o = smp_playOnceAtSpeed(&s, x); break;           /* C:1017, C:1026 */
                case 3: o = smp_playUntilAtSpeed(&s, p0[v] + p1[v], x); break;   /* C:1034 */
                case 4: o = smp_play(&s); break;                                 /* C:1041 */
            }
            out[n * V + v] = o;
        }
        position[v] = s.position;
        zx_prev[v] = z.previousValue;
        zx_first[v] = z.firstTrigger;
    }
    return 0;
}

/* maxiSample::playWithPhasor C:753-816.  pos1/pos2 are size_t in the reference: `pos1--` at 0
 * wraps to SIZE_MAX and is then caught by `pos1 >= amplen` -> 0 (the `< 0` tests are dead). */
int mxo_sample_phasor(size_t V, size_t N, const double *amp, size_t len, const double *pha_in,
                      double *phasor_prev, int32_t *phasor_first, double *out) {
    const size_t amplen = len;
    for (size_t v = 0; v < V; v++) {
        double phasorPrev = phasor_prev[v];
        int phasorFirst = phasor_first[v] != 0;
        for (size_t n = 0; n < N; n++) {
            double pha = pha_in[n * V + v];
            if (pha > 1) pha = 1;
            if (pha < 0) pha = 0;
            double pos = pha * amplen * 0.99999999999999;
            if (phasorFirst) {
                phasorFirst = 0;
                phasorPrev = pos;
            }
            size_t pos1 = (size_t)(round(phasorPrev));
            size_t pos2 = (size_t)(round(pos));
            if (pos1 == pos2) {
                if (pos >= phasorPrev) pos2++;
                else pos1--;
            }
            if (pos2 >= amplen) pos2 = 0;
            if (pos1 >= amplen) pos1 = 0;
            double q1, q2;
            if (pos2 > pos1) {
                double dist = pos2 - pos1;
                if (dist == 0) q1 = 0;
                else q1 = (pos - pos1) / dist;
            } else {
                double dist = (amplen - pos1) + pos2;
                if (dist == 0) q1 = 0;
                else {
                    if (pos > pos1) q1 = (pos - pos1) / dist;
                    else q1 = ((amplen - pos1) + pos) / dist;
                }
            }
            q2 = 1 - q1;
            out[n * V + v] = (q1 * amp[pos1] + q2 * amp[pos2]);
            phasorPrev = pos;
        }
        phasor_prev[v] = phasorPrev;
        phasor_first[v] = phasorFirst;
    }
    return 0;
}

/* maxiMix::stereo C:503-509, quad C:512-522, ambisonic C:525-541 (quirks kept: z never clamped,
 * z>1 / z<0 overwrite y, eight[0..3] = input*(sqrt(..)*1.0-z)) + voice-order sum. */
int mxo_mix_bus(int C, size_t V, size_t N, const double *in, const double *px, const double *py,
                const double *pz, double *bus, double *mix) {
    if (C != 2 && C != 4 && C != 8) return -1;
    for (size_t n = 0; n < N; n++) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t v = 0; v < V; v++) {
            const double input = in[n * V + v];
            double o[8], x = px[v], y = 0, z = 0;
            if (x > 1) x = 1;
            if (x < 0) x = 0;
            if (C >= 4) {
                y = py[v];
                if (y > 1) y = 1;
                if (y < 0) y = 0;
            }
            if (C == 2) {
                o[0] = input * sqrt(1.0 - x);
                o[1] = input * sqrt(x);
            } else if (C == 4) {
                o[0] = input * sqrt((1.0 - x) * y);
                o[1] = input * sqrt((1.0 - x) * (1.0 - y));
                o[2] = input * sqrt(x * y);
                o[3] = input * sqrt(x * (1.0 - y));
            } else {
                z = pz[v];
                if (z > 1) y = 1;
                if (z < 0) y = 0;
                o[0] = input * (sqrt((1.0 - x) * y) * 1.0 - z);
                o[1] = input * (sqrt((1.0 - x) * (1.0 - y)) * 1.0 - z);
                o[2] = input * (sqrt(x * y) * 1.0 - z);
                o[3] = input * (sqrt(x * (1.0 - y)) * 1.0 - z);
                o[4] = input * (sqrt((1.0 - x) * y) * z);
                o[5] = input * (sqrt((1.0 - x) * (1.0 - y)) * z);
                o[6] = input * sqrt((x * y) * z);
                o[7] = input * sqrt((x * (1.0 - y)) * z);
            }
            for (int c = 0; c < C; c++) {
                if (bus) bus[(n * C + c) * V + v] = o[c];
                acc[c] += o[c];
            }
        }
        for (int c = 0; c < C; c++) mix[n * C + c] = acc[c];
    }
    return 0;
}

/* maxiOsc::noise C:214-220 over libc rand(): see oracle/ref_harness.cpp for the draw protocol. */
int mxo_noise(unsigned seed, size_t V, size_t N, int32_t *rnd, double *out) {
    srand(seed);
    for (size_t i = 0; i < V * N; i++) {
        rnd[i] = rand();
        float r = rnd[i] / (float)RAND_MAX;
        out[i] = r * 2 - 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * fft / maxiFFT (L/fft.cpp:118-282, 390-534; L/maxiFFT.cpp:45-91).  Everything is fp32
 * except the trigonometric seeds (double sin/cos rounded to float).  x86-64 SSE evaluates
 * float expressions in float (FLT_EVAL_METHOD 0), so each `a*b - c*d` below is three
 * separately rounded float operations -- -ffp-contract=off keeps it that way.
 * ------------------------------------------------------------------------------------ */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static int fft_reverse_bits(int index, int NumBits) { /* L/fft.cpp:74-85 */
    int i, rev;
    for (i = rev = 0; i < NumBits; i++) {
        rev = (rev << 1) | (index & 1);
        index >>= 1;
    }
    return rev;
}

/* L/fft.cpp:118-211 */
static void fft_complex_dir(int NumSamples, int InverseTransform, const float *RealIn, const float *ImagIn,
                            float *RealOut, float *ImagOut) {
    int NumBits = 0, i, j, k, n, BlockSize, BlockEnd;
    double angle_numerator = 2.0 * M_PI;
    float tr, ti;
    if (InverseTransform) angle_numerator = -angle_numerator; /* :137-138 */
    while (!(NumSamples & (1 << NumBits))) NumBits++; /* NumberOfBitsNeeded, :60-72 */
    for (i = 0; i < NumSamples; i++) {
        j = fft_reverse_bits(i, NumBits);
        RealOut[j] = RealIn[i];
        ImagOut[j] = (ImagIn == NULL) ? 0.0 : ImagIn[i];
    }
    BlockEnd = 1;
    for (BlockSize = 2; BlockSize <= NumSamples; BlockSize <<= 1) {
        double delta_angle = angle_numerator / (double)BlockSize;
        float sm2 = sin(-2 * delta_angle);
        float sm1 = sin(-delta_angle);
        float cm2 = cos(-2 * delta_angle);
        float cm1 = cos(-delta_angle);
        float w = 2 * cm1;
        float ar0, ar1, ar2, ai0, ai1, ai2;
        for (i = 0; i < NumSamples; i += BlockSize) {
            ar2 = cm2;
            ar1 = cm1;
            ai2 = sm2;
            ai1 = sm1;
            for (j = i, n = 0; n < BlockEnd; j++, n++) {
                ar0 = w * ar1 - ar2;
                ar2 = ar1;
                ar1 = ar0;
                ai0 = w * ai1 - ai2;
                ai2 = ai1;
                ai1 = ai0;
                k = j + BlockEnd;
                tr = ar0 * RealOut[k] - ai0 * ImagOut[k];
                ti = ar0 * ImagOut[k] + ai0 * RealOut[k];
                RealOut[k] = RealOut[j] - tr;
                ImagOut[k] = ImagOut[j] - ti;
                RealOut[j] += tr;
                ImagOut[j] += ti;
            }
        }
        BlockEnd = BlockSize;
    }
    if (InverseTransform) { /* :201-209 */
        float denom = (float)NumSamples;
        for (i = 0; i < NumSamples; i++) {
            RealOut[i] /= denom;
            ImagOut[i] /= denom;
        }
    }
}

static void fft_complex(int NumSamples, const float *RealIn, const float *ImagIn, float *RealOut,
                        float *ImagOut) {
    fft_complex_dir(NumSamples, 0, RealIn, ImagIn, RealOut, ImagOut);
}

/* L/fft.cpp:228-282 */
static void fft_real(int NumSamples, const float *RealIn, float *RealOut, float *ImagOut) {
    int Half = NumSamples / 2;
    int i;
    float theta = M_PI / Half;
    float *tmpReal = (float *)malloc(Half * sizeof(float));
    float *tmpImag = (float *)malloc(Half * sizeof(float));
    for (i = 0; i < Half; i++) {
        tmpReal[i] = RealIn[2 * i];
        tmpImag[i] = RealIn[2 * i + 1];
    }
    fft_complex(Half, tmpReal, tmpImag, RealOut, ImagOut);
    float wtemp = (float)(sin(0.5 * theta));
    float wpr = -2.0 * wtemp * wtemp;
    float wpi = (float)(sin(theta));
    float wr = 1.0 + wpr;
    float wi = wpi;
    int i3;
    float h1r, h1i, h2r, h2i;
    for (i = 1; i < Half / 2; i++) {
        i3 = Half - i;
        h1r = 0.5 * (RealOut[i] + RealOut[i3]);
        h1i = 0.5 * (ImagOut[i] - ImagOut[i3]);
        h2r = 0.5 * (ImagOut[i] + ImagOut[i3]);
        h2i = -0.5 * (RealOut[i] - RealOut[i3]);
        RealOut[i] = h1r + wr * h2r - wi * h2i;
        ImagOut[i] = h1i + wr * h2i + wi * h2r;
        RealOut[i3] = h1r - wr * h2r + wi * h2i;
        ImagOut[i3] = -h1i + wr * h2i + wi * h2r;
        wtemp = wr;
        wr = wtemp * wpr - wi * wpi + wr;
        wi = wi * wpr + wtemp * wpi + wi;
    }
    h1r = RealOut[0];
    RealOut[0] = h1r + ImagOut[0];
    ImagOut[0] = h1r - ImagOut[0];
    free(tmpReal);
    free(tmpImag);
}

/* maxiFFT streamed over a signal: setup(fftSize, hopSize, windowSize) (L/maxiFFT.cpp:45-60),
 * then process(value, WITH_POLAR_CONVERSION) for every sample (L/maxiFFT.cpp:65-91); each time
 * it reports a new frame, real/imag/magnitudes/phases (bins = fftSize/2 each) are appended to
 * the outputs.  Returns the number of frames, or <0.  Any output pointer may be NULL.
 * windowSize > fftSize overruns the reference's window/buffer vectors -> rejected (-2). */
long mxo_fft_stream(const float *signal, size_t nsamples, int fftSize, int hopSize, int windowSize,
                    size_t max_frames, float *real, float *imag, float *mags, float *phases) {
    if (fftSize < 4 || (fftSize & (fftSize - 1))) return -1;
    int win = windowSize > fftSize ? windowSize : fftSize; /* :48 */
    if (win > fftSize || hopSize <= 0 || hopSize > win) return -2;
    int bins = fftSize / 2;
    float *buffer = (float *)calloc(fftSize, sizeof(float));
    float *window = (float *)calloc(fftSize, sizeof(float));
    float *in_real = (float *)calloc(fftSize, sizeof(float));
    float *out_real = (float *)calloc(fftSize, sizeof(float));
    float *out_img = (float *)calloc(fftSize, sizeof(float));
    for (int i = 0; i < win; i++) /* genWindow(3,...) Hanning, L/fft.cpp:409-413 */
        window[i] = 0.50 - 0.50 * cos(2 * M_PI * i / (win - 1));
    int pos = win - hopSize; /* :56 */
    size_t frames = 0;
    for (size_t s = 0; s < nsamples; s++) {
        buffer[pos++] = signal[s];
        if (pos == win) {
            if (frames >= max_frames) break;
            for (int i = 0; i < fftSize; i++) in_real[i] = buffer[0 + i] * window[i]; /* calcFFT :499-505 */
            fft_real(fftSize, in_real, out_real, out_img);
            for (int i = 0; i < bins; i++) { /* cartToPol :507-515 */
                float power = out_real[i] * out_real[i] + out_img[i] * out_img[i];
                if (mags) mags[frames * bins + i] = sqrtf(power);
                if (phases) phases[frames * bins + i] = atan2f(out_img[i], out_real[i]);
                if (real) real[frames * bins + i] = out_real[i];
                if (imag) imag[frames * bins + i] = out_img[i];
            }
            frames++;
            memmove(buffer, buffer + hopSize, (win - hopSize) * sizeof(float)); /* :85 */
            pos = win - hopSize;
        }
    }
    free(buffer);
    free(window);
    free(in_real);
    free(out_real);
    free(out_img);
    return (long)frames;
}

/* fft::convToDB (L/fft.cpp:526-534) over an array */
void mxo_fft_to_db(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (in[i] < 0.000001) {
            out[i] = 0;
        } else {
            /* C++: in[i]+1 is float, so log10 resolves to the float overload (log10f) */
            out[i] = 20.0 * log10f(in[i] + 1);
        }
    }
}

/* maxiFFT::spectralFlatness L/maxiFFT.cpp:113-123, spectralCentroid :125-132.  All float:
 * logf/expf; `fabs(magnitudes[i]) * i` resolves to the float overload (pinned against the compiled
 * reference), and size_t i converts to float. */
int mxo_fft_features(const float *mags, size_t nframes, int fftSize, float *flatness, float *centroid) {
    if (fftSize < 4 || (fftSize & (fftSize - 1))) return -1;
    const size_t bins = (size_t)fftSize / 2;
    for (size_t k = 0; k < nframes; k++) {
        const float *m = mags + k * bins;
        if (flatness) {
            float geometricMean = 0, arithmaticMean = 0;
            for (size_t i = 0; i < bins; i++) {
                if (m[i] != 0) geometricMean += logf(m[i]);
                arithmaticMean += m[i];
            }
            geometricMean = expf(geometricMean / (float)bins);
            arithmaticMean /= (float)bins;
            flatness[k] = arithmaticMean != 0 ? geometricMean / arithmaticMean : 0;
        }
        if (centroid) {
            float x = 0, y = 0;
            for (size_t i = 0; i < bins; i++) {
                x += fabsf(m[i]) * i;
                y += fabsf(m[i]);
            }
            centroid[k] = y != 0 ? x / y * ((float)g_sampleRate / fftSize) : 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiMFCCAnalyser<double> (L/maxiMFCC.h:30-211, L/maxiMFCC.cpp:48-66).
 * ------------------------------------------------------------------------------------ */
static double mfcc_hzToMel(double hz) { return 2595.0 * (log10(hz / 700.0 + 1.0)); }       /* :30-32 */
static double mfcc_melToHz(double mel) { return 700.0 * (pow(10, mel / 2595.0) - 1.0); } /* :36-38 */

/* setup() tables.  melFilters[filter + bin*numFilters] (L/maxiMFCC.h:118-182); filter 0 is
 * never written by the reference (loop starts at 1, :149) -- defined as 0 here, which is what
 * a fresh mmap'd malloc holds.  dct[i + j*numCoeffs] (:183-203). */
int mxo_mfcc_tables(unsigned numBins, unsigned numFilters, unsigned numCoeffs, double minFreq,
                    double maxFreq, double *melFilters, double *dct) {
    double mel, dMel, maxMel, minMel, nyquist, binFreq, start, thisF, nextF, prevF;
    unsigned int sampleRate_u = (unsigned int)g_sampleRate; /* `unsigned int sampleRate` member */
    double sampleRate = sampleRate_u;
    int numValidBins = numBins;
    nyquist = sampleRate / 2;
    if (maxFreq > nyquist) maxFreq = nyquist;
    maxMel = mfcc_hzToMel(maxFreq);
    minMel = mfcc_hzToMel(minFreq);
    dMel = (maxMel - minMel) / (numFilters + 2 - 1);
    double *filtPos = (double *)malloc(sizeof(double) * (numFilters + 2));
    mel = minMel;
    for (unsigned i = 0; i < numFilters + 2; i++) {
        filtPos[i] = mfcc_melToHz(mel);
        mel += dMel;
    }
    memset(melFilters, 0, sizeof(double) * numFilters * numValidBins);
    for (unsigned filter = 1; filter < numFilters; filter++) {
        for (int bin = 0; bin < numValidBins; bin++) {
            binFreq = (double)sampleRate / (double)numValidBins * (double)bin;
            thisF = filtPos[filter];
            nextF = filtPos[filter + 1];
            prevF = filtPos[filter - 1];
            int idx = filter + (bin * numFilters);
            if (binFreq > nextF || binFreq < prevF) {
                melFilters[idx] = 0;
            } else {
                double height = 2.0 / (nextF - prevF);
                if (binFreq < thisF) {
                    start = prevF;
                    melFilters[idx] = (binFreq - start) * (height / (thisF - start));
                } else {
                    melFilters[idx] = height + ((binFreq - thisF) * (-height / (nextF - thisF)));
                }
            }
        }
    }
    free(filtPos);
    double k = 3.14159265358979323846 / numFilters;
    double w1 = 1.0 / (sqrt(numFilters));
    double w2 = sqrt(2.0 / numFilters);
    for (unsigned i = 0; i < numCoeffs; i++) {
        for (unsigned j = 0; j < numFilters; j++) {
            int idx = i + (j * numCoeffs);
            if (i == 0)
                dct[idx] = w1 * cos(k * (i + 1) * (j + 0.5));
            else
                dct[idx] = w2 * cos(k * (i + 1) * (j + 0.5));
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiGrains (L/maxiGrains.h): window functors :18-90, maxiGrain :137-251,
 * maxiGrainPlayer :253-284, maxiTimeStretch::play :341-355, maxiStretch::play :512-530.
 * ------------------------------------------------------------------------------------ */
/* window kinds: 0 hann 1 hamming 2 cosine 3 rect 4 triangle 5 triangleNZ 6 blackmanHarris
 * 7 blackmanNutall 8 gaussian(kurtosis 0.3) */
static double grain_window_value(int kind, unsigned long windowLength, unsigned long windowPos) {
    switch (kind) {
        case 0: return 0.5 * (1.0 - cos((2.0 * MX_PI * windowPos) / (windowLength - 1)));           /* :19-21 */
        case 1: return 0.54 - (0.46 * cos((2.0 * MX_PI * windowPos) / (windowLength - 1)));         /* :26-28 */
        case 2: return sin((MX_PI * windowPos) / (windowLength - 1));                               /* :33-35 */
        case 3: return 1;                                                                          /* :40-42 */
        case 4:                                                                                    /* :47-49 */
            return (2.0 / (windowLength - 1.0)) *
                   (((windowLength - 1.0) / 2.0) - fabs(windowPos - ((windowLength - 1.0) / 2.0)));
        case 5:                                                                                    /* :54-56 */
            return (2.0 / windowLength) * ((windowLength / 2.0) - fabs(windowPos - ((windowLength - 1.0) / 2.0)));
        case 6:                                                                                    /* :61-66 */
            return 0.35875 - (0.48829 * cos((2 * MX_PI * windowPos) / (windowLength - 1))) +
                   (0.14128 * cos((4 * MX_PI * windowPos) / (windowLength - 1))) +
                   (0.01168 * cos((6 * MX_PI * windowPos) / (windowLength - 1)));
        case 7:                                                                                    /* :71-76 */
            return 0.3635819 - (0.4891775 * cos((2 * MX_PI * windowPos) / (windowLength - 1))) +
                   (0.1365995 * cos((4 * MX_PI * windowPos) / (windowLength - 1))) +
                   (0.0106411 * cos((6 * MX_PI * windowPos) / (windowLength - 1)));
        case 8: {                                                                                  /* :79-97 */
            double gausDivisor = (-2.0 * 0.3 * 0.3);
            double phase = ((windowPos / (double)windowLength) - 0.5) * 2.0;
            return exp((phase * phase) / gausDivisor);
        }
    }
    return 0;
}

/* maxiGrainWindowCache::getWindow (:112-120): the table for one length */
int mxo_grain_window(int kind, unsigned length, double *out) {
    if (kind < 0 || kind > 8) return -1;
    for (unsigned i = 0; i < length; i++) out[i] = grain_window_value(kind, length, i);
    return 0;
}

#define MXO_GRAIN_SLOTS 8
typedef struct {
    double pos, inc;
    unsigned long sampleIdx, sampleDur; /* sampleDur == 0: empty slot */
} grain_t;

/* maxiGrain ctor (:160-181) */
static void grain_init(grain_t *g, size_t len, int mySampleRate, double position, double duration,
                       double speed) {
    unsigned long sampleStartPos = len * position;
    unsigned long sampleDur = duration * (double)mySampleRate;
    double freq = 1.0 / duration;
    unsigned long sampleEndPos = len < sampleStartPos + sampleDur ? len : sampleStartPos + sampleDur;
    double frequency = freq * speed;
    if (frequency > 0) {
        g->pos = sampleStartPos;
    } else {
        g->pos = sampleEndPos;
    }
    if (frequency != 0) {
        g->inc = sampleDur / (g_sampleRate / frequency);
    } else
        g->inc = 0;
    g->sampleIdx = 0;
    g->sampleDur = sampleDur;
}

/* maxiGrain::play (:216-245), non-MAXIGRAINFAST path.  Returns 1 when the grain finished. */
static int grain_play(grain_t *g, const double *buffer, size_t len, const double *window, double *out) {
    double output = 0.0;
    {
        double envValue = window[g->sampleIdx];
        double remainder;
        g->pos += g->inc;
        if (g->pos >= len)
            g->pos -= len;
        else if (g->pos < 0)
            g->pos += len;
        long posl = floor(g->pos);
        remainder = g->pos - posl;
        long a = posl;
        long b = posl + 1;
        if ((size_t)b >= len) {
            b = 0;
        }
        output = (double)((1 - remainder) * buffer[a] + remainder * buffer[b]);
        output *= envValue;
    }
    g->sampleIdx++;
    *out = output;
    return g->sampleIdx == g->sampleDur;
}

/* S independent maxiTimeStretch<F> (mode 0) / maxiStretch<F> (mode 1, loop = whole sample)
 * objects over one shared sample, T samples each; out[n*S + s].
 *   mode 0: play(speed=a[s], grainLength, overlaps, posMod[s])                (:341-355)
 *   mode 1: play(pitchstretch=a[s], timestretch=b[s], grainLength, overlaps, posMod[s]) (:512-530)
 *   mode 2: maxiTimeStretch::playAtPosition(pos=a[n*S+s] (per sample), grainLength, overlaps) (:359-367)
 *   mode 3: maxiPitchShift::play(speed=a[s], grainLength, overlaps, posMod[s])  (:412-430); the
 *           `looper` slot of st holds the member `cycles` (long)
 * rnd: the values `rand() % 10` would have returned, consumed in order per stream, [S][R]
 * (NULL = all 0); the global rand() stream itself is not reproducible across a bank.
 * st = [4][S]: position, looper, randomOffset, rand cursor.  gst = [4][8][S]: per slot pos, inc,
 * sampleIdx, sampleDur (0 = empty); slots hold the live grains in creation order.
 * `amp` must be valid on [0, len] (one guard element, see mxo_sample).  Returns 0, or -3 if more
 * than 8 grains are alive at once, -4 if R is exhausted, -2 bad window length. */
int mxo_granular(int mode, int window_kind, size_t S, size_t T, const double *amp, size_t len,
                 int mySampleRate, double grainLength, int overlaps, const double *a, const double *b,
                 const double *posMod, const int32_t *rnd, size_t R, double *st, double *gst, double *out) {
    if (mode < 0 || mode > 3 || overlaps <= 0) return -1;
    unsigned long sampleDur = grainLength * (double)mySampleRate;
    if (sampleDur == 0 || sampleDur >= (unsigned long)(g_sampleRate / 2.0)) return -2; /* cacheSize :98 */
    double *window = (double *)malloc(sizeof(double) * sampleDur);
    mxo_grain_window(window_kind, (unsigned)sampleDur, window);
    int rc = 0;
    for (size_t s = 0; s < S && rc == 0; s++) {
        double position = st[0 * S + s], looper = st[1 * S + s], randomOffset = st[2 * S + s];
        size_t cursor = (size_t)st[3 * S + s];
        grain_t g[MXO_GRAIN_SLOTS];
        int count = 0;
        for (int k = 0; k < MXO_GRAIN_SLOTS; k++) {
            g[k].pos = gst[(0 * MXO_GRAIN_SLOTS + k) * S + s];
            g[k].inc = gst[(1 * MXO_GRAIN_SLOTS + k) * S + s];
            g[k].sampleIdx = (unsigned long)gst[(2 * MXO_GRAIN_SLOTS + k) * S + s];
            g[k].sampleDur = (unsigned long)gst[(3 * MXO_GRAIN_SLOTS + k) * S + s];
            if (g[k].sampleDur) count = k + 1;
        }
        const unsigned long loopStart = 0, loopEnd = len, loopLength = len; /* maxiStretch ctor :469-477 */
        for (size_t n = 0; n < T; n++) {
            double speed = mode == 2 ? 0.0 : a[s];
            int spawn = 0, draws = 0;
            double grainSpeed = 0, grainPos = 0;
            double cycleLength = grainLength * g_sampleRate / overlaps;
            if (mode == 2) { /* playAtPosition :359-367 (position is not touched) */
                double pos = a[n * S + s];
                looper++;
                pos *= len;
                if (0 == floor(fmod(looper, grainLength * g_sampleRate / overlaps))) {
                    grainPos = (pos / len);
                    grainSpeed = 1;
                    spawn = 1;
                }
            } else if (mode == 3) { /* maxiPitchShift::play :412-430 (looper holds `cycles`) */
                position = position + 1;
                looper++;
                if (position > len) position = 0;
                if (position < 0) position = len;
                double cycleMod = fmod((long)looper, cycleLength + randomOffset);
                if (0 == floor(cycleMod)) {
                    grainSpeed = speed - ((cycleMod / cycleLength) * 0.1);
                    grainPos = (position / len) + (posMod ? posMod[s] : 0.0);
                    spawn = 1;
                }
            } else {
                if (mode == 0) {
                    position = position + speed;
                    looper++;
                    if (position > len) position -= len;
                    if (position < 0) position += len;
                } else {
                    position = position + (1 * b[s]);
                    looper++;
                    if (position >= loopEnd) position -= loopLength;
                    if (position < loopStart) position += loopLength;
                }
                if (looper > cycleLength + randomOffset) {
                    looper -= (cycleLength + randomOffset);
                    grainSpeed = mode == 0 ? (speed > 0 ? 1 : -1) : speed;
                    grainPos = (position / len) + (posMod ? posMod[s] : 0.0);
                    spawn = 1;
                    draws = 1; /* randomOffset = rand() % 10, :352 / :525 */
                }
            }
            if (spawn) {
                if (count == MXO_GRAIN_SLOTS) {
                    rc = -3;
                    break;
                }
                double p01 = grainPos;
                p01 = 1.0 < p01 ? 1.0 : p01; /* min(1.0, x) */
                p01 = p01 < 0.0 ? 0.0 : p01; /* max(x, 0.0) */
                grain_init(&g[count++], len, mySampleRate, p01, grainLength, grainSpeed);
                if (draws) {
                    if (rnd) {
                        if (cursor >= R) {
                            rc = -4;
                            break;
                        }
                        randomOffset = rnd[s * R + cursor++];
                    } else
                        randomOffset = 0;
                }
            }
            /* maxiGrainPlayer::play (:270-283): sum in list order, erase the finished */
            double total = 0.0;
            int w = 0;
            for (int k = 0; k < count; k++) {
                double o;
                int fin = grain_play(&g[k], amp, len, window, &o);
                total += o;
                if (!fin) g[w++] = g[k];
            }
            for (int k = w; k < count; k++) g[k].sampleDur = 0;
            count = w;
            out[n * S + s] = total;
        }
        st[0 * S + s] = position;
        st[1 * S + s] = looper;
        st[2 * S + s] = randomOffset;
        st[3 * S + s] = (double)cursor;
        for (int k = 0; k < MXO_GRAIN_SLOTS; k++) {
            int live = k < count;
            gst[(0 * MXO_GRAIN_SLOTS + k) * S + s] = live ? g[k].pos : 0.0;
            gst[(1 * MXO_GRAIN_SLOTS + k) * S + s] = live ? g[k].inc : 0.0;
            gst[(2 * MXO_GRAIN_SLOTS + k) * S + s] = live ? (double)g[k].sampleIdx : 0.0;
            gst[(3 * MXO_GRAIN_SLOTS + k) * S + s] = live ? (double)g[k].sampleDur : 0.0;
        }
    }
    free(window);
    return rc;
}

/* mfcc() over nframes magnitude spectra (mags[f*mag_stride + bin]) -> melBands (after the
 * log-square, [nframes][numFilters], may be NULL) and coefficients [nframes][numCoeffs].
 * L/maxiMFCC.cpp:48-66 then L/maxiMFCC.h:98-111. */
int mxo_mfcc(unsigned numBins, unsigned numFilters, unsigned numCoeffs, double minFreq,
             double maxFreq, const float *mags, size_t mag_stride, size_t nframes, double *melbands,
             double *mfcc) {
    double *W = (double *)malloc(sizeof(double) * numFilters * numBins);
    double *D = (double *)malloc(sizeof(double) * numCoeffs * numFilters);
    double *mb = (double *)malloc(sizeof(double) * numFilters);
    mxo_mfcc_tables(numBins, numFilters, numCoeffs, minFreq, maxFreq, W, D);
    for (size_t f = 0; f < nframes; f++) {
        const float *powerSpectrum = mags + f * mag_stride;
        for (unsigned filter = 0; filter < numFilters; filter++) {
            mb[filter] = 0.0;
            for (unsigned bin = 0; bin < numBins; bin++) {
                int idx = filter + (bin * numFilters);
                mb[filter] += (W[idx] * powerSpectrum[bin]);
            }
        }
        for (unsigned filter = 0; filter < numFilters; filter++)
            mb[filter] = mb[filter] > 0.000001 ? log(mb[filter] * mb[filter]) : 0.0;
        if (melbands) memcpy(melbands + f * numFilters, mb, sizeof(double) * numFilters);
        double *c = mfcc + f * numCoeffs;
        for (unsigned i = 0; i < numCoeffs; i++) c[i] = 0.0;
        for (unsigned i = 0; i < numCoeffs; i++)
            for (unsigned j = 0; j < numFilters; j++) c[i] += (D[i + (j * numCoeffs)] * mb[j]);
        for (unsigned i = 0; i < numCoeffs; i++) c[i] /= numCoeffs;
    }
    free(W);
    free(D);
    free(mb);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiSample::load / read (C:605-692) and save (C:698-725): 16-bit PCM RIFF/WAVE.
 * read(): ChunkSize at byte 4; SubChunk1Size at 16, then format, channels, sampleRate, byteRate,
 * blockAlign, bitsPerSample read sequentially (bytes 20..35); chunks are walked from
 * 20 + SubChunk1Size until "data"; the data are read as shorts; for multi-channel files the
 * reference de-interleaves IN PLACE with `for (i = readChannel*2; i < myDataSize+6; i += myChannels*2)
 * shortAmps[position++] = shortAmps[i]` -- a SHORT index advanced by a BYTE stride and bounded by
 * a byte count: it picks every (2*channels)-th short, keeps the vector at its full length (the
 * tail keeps the interleaved data) and reads past the end of the vector for i >= size.  Those
 * out-of-range reads are undefined in the reference; here they read 0 (parity: compare the
 * defined prefix).  amplitudes[i] = short/32767.0; position = size.
 * hdr as in oracle/ref_harness.cpp.  Returns size, -1 cannot open, -2 malformed.
 * ------------------------------------------------------------------------------------ */
#include <stdio.h>
long mxo_wav_load(const char *path, int channel, double *out, size_t cap, int32_t *hdr, double *position) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long fsize = ftell(f);
    unsigned char *b = (unsigned char *)calloc((size_t)fsize + 16, 1);
    fseek(f, 0, SEEK_SET);
    if (fread(b, 1, (size_t)fsize, f) != (size_t)fsize) { fclose(f); free(b); return -2; }
    fclose(f);
    if (fsize < 44) { free(b); return -2; }
    int32_t myChunkSize, mySubChunk1Size, mySampleRate, myByteRate, myDataSize = 0;
    int16_t myFormat, myChannels, myBlockAlign, myBitsPerSample;
    memcpy(&myChunkSize, b + 4, 4);
    memcpy(&mySubChunk1Size, b + 16, 4);
    memcpy(&myFormat, b + 20, 2);
    memcpy(&myChannels, b + 22, 2);
    memcpy(&mySampleRate, b + 24, 4);
    memcpy(&myByteRate, b + 28, 4);
    memcpy(&myBlockAlign, b + 32, 2);
    memcpy(&myBitsPerSample, b + 34, 2);
    long filePos = 20 + (long)mySubChunk1Size;
    int datafound = 0;
    while (!datafound) {
        if (filePos < 0 || filePos + 8 > fsize) { free(b); return -2; } /* the reference runs into eof here */
        memcpy(&myDataSize, b + filePos + 4, 4);
        const int isdata = memcmp(b + filePos, "data", 4) == 0;
        filePos += 8;
        if (isdata) datafound = 1; else filePos += myDataSize;
    }
    if (myDataSize < 0) { free(b); return -2; }
    const size_t n = (size_t)myDataSize / 2;
    int16_t *sh = (int16_t *)calloc(n + 1, sizeof(int16_t));
    size_t avail = (size_t)(fsize - filePos);
    if (avail > (size_t)myDataSize) avail = (size_t)myDataSize;
    memcpy(sh, b + filePos, avail > 2 * n ? 2 * n : avail); /* a short file leaves zeros, as the resized vector does */
    if (myChannels > 1) {
        size_t pos = 0;
        for (long i = (long)channel * 2; i < (long)myDataSize + 6; i += ((long)myChannels * 2)) {
            const int16_t v = ((size_t)i < n) ? sh[i] : 0; /* i >= n: undefined in the reference */
            if (pos < n) sh[pos] = v;
            pos++;
        }
    }
    for (size_t i = 0; i < n && i < cap; i++) out[i] = sh[i] / 32767.0;
    if (hdr) {
        hdr[0] = myChunkSize; hdr[1] = mySubChunk1Size; hdr[2] = myFormat; hdr[3] = myChannels;
        hdr[4] = mySampleRate; hdr[5] = myByteRate; hdr[6] = myBlockAlign; hdr[7] = myBitsPerSample;
    }
    if (position) *position = (double)n;
    free(sh);
    free(b);
    return (long)n;
}

/* save(): shorts = static_cast<short>(round(amplitude*32767.0)); 44-byte header from the members. */
int mxo_wav_save(const char *path, const double *amp, size_t len, const int32_t *hdr) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int16_t *sh = (int16_t *)malloc(sizeof(int16_t) * (len ? len : 1));
    for (size_t i = 0; i < len; i++) sh[i] = (int16_t)(int32_t)round(amp[i] * 32767.0);
    const int32_t chunk = hdr[0], sub1 = hdr[1], rate = hdr[4], brate = hdr[5], dsize = (int32_t)(len * 2);
    const int16_t fmt = (int16_t)hdr[2], ch = (int16_t)hdr[3], align = (int16_t)hdr[6], bits = (int16_t)hdr[7];
    fwrite("RIFF", 1, 4, f); fwrite(&chunk, 4, 1, f); fwrite("WAVE", 1, 4, f); fwrite("fmt ", 1, 4, f);
    fwrite(&sub1, 4, 1, f); fwrite(&fmt, 2, 1, f); fwrite(&ch, 2, 1, f); fwrite(&rate, 4, 1, f);
    fwrite(&brate, 4, 1, f); fwrite(&align, 2, 1, f); fwrite(&bits, 2, 1, f); fwrite("data", 1, 4, f);
    fwrite(&dsize, 4, 1, f); fwrite(sh, 2, len, f);
    fclose(f);
    free(sh);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiIFFT::setup/process L/maxiFFT.cpp:140-192 (SPECTRUM mode); fft::polToCart L/fft.cpp:590-604,
 * calcIFFT :606-611, inversePowerSpectrum :622-625.
 * polToCart: in_real[i] = magnitude[i]*cos(phase[i]) -- fft.cpp includes <math.h>/<iostream>, so the
 * float argument selects the float overload (cosf/sinf), pinned against the compiled reference; the
 * upper half (negative frequencies) is zeroed; a FULL n-point complex inverse FFT follows, only its
 * real part is used: finalOut[i] += out_real[i]*window[i] into a zeroed ifftOut, then the hop buffer
 * is shifted by hopSize, its tail zeroed and ifftOut added.  window = Hann over windowSize
 * (windowSize ? windowSize : fftSize), zero beyond.
 * ------------------------------------------------------------------------------------ */
int mxo_ifft_stream(const float *mags, const float *phases, size_t nframes, int fftSize, int hopSize,
                    int windowSize, float *out, float *ifft_out, float *buffer_io) {
    if (fftSize < 4 || (fftSize & (fftSize - 1)) || hopSize <= 0 || hopSize > fftSize) return -1;
    if (windowSize > fftSize) return -2;
    const int n = fftSize, half = fftSize / 2;
    const int win = windowSize ? windowSize : fftSize; /* :143 */
    float *window = (float *)calloc(n, sizeof(float));
    for (int i = 0; i < win; i++) window[i] = 0.50 - 0.50 * cos(2 * M_PI * i / (win - 1));
    float *in_real = (float *)calloc(n, sizeof(float)), *in_img = (float *)calloc(n, sizeof(float));
    float *out_real = (float *)calloc(n, sizeof(float)), *out_img = (float *)calloc(n, sizeof(float));
    float *ifftOut = (float *)calloc(n, sizeof(float)), *buffer = (float *)calloc(n, sizeof(float));
    if (buffer_io) memcpy(buffer, buffer_io, sizeof(float) * n);
    for (size_t k = 0; k < nframes; k++) {
        const float *magnitude = mags + k * half, *phase = phases + k * half;
        for (int i = 0; i < n; i++) ifftOut[i] = 0;
        for (int i = 0; i < half; i++) {
            in_real[i] = magnitude[i] * cosf(phase[i]);
            in_img[i] = magnitude[i] * sinf(phase[i]);
        }
        memset(in_real + half, 0, sizeof(float) * half);
        memset(in_img + half, 0, sizeof(float) * half);
        fft_complex_dir(n, 1, in_real, in_img, out_real, out_img);
        for (int i = 0; i < n; i++) ifftOut[i] += out_real[i] * window[i];
        if (ifft_out) memcpy(ifft_out + k * n, ifftOut, sizeof(float) * n);
        memmove(buffer, buffer + hopSize, (n - hopSize) * sizeof(float));
        memset(buffer + (n - hopSize), 0, hopSize * sizeof(float));
        for (int i = 0; i < n; i++) buffer[i] += ifftOut[i];
        for (int i = 0; i < hopSize; i++) out[k * hopSize + i] = buffer[i];
    }
    if (buffer_io) memcpy(buffer_io, buffer, sizeof(float) * n);
    free(window); free(in_real); free(in_img); free(out_real); free(out_img); free(ifftOut); free(buffer);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiConvolve (L/maxiConvolve.cpp:13-107): partitioned convolution over a frequency delay line.
 * setup(impulseFile, fftsize, hopsize):
 *   - the impulse maxiSample is loaded (amplitudes = pcm/32767.0, position = size, C:679-681) and played with
 *     play() (C:740-747) getLength() times into maxiFFT::setup(fftsize, fftsize, hopsize) -- note the argument
 *     order: hop = fftsize, window = max(hopsize, fftsize): non-overlapping Hann-windowed frames (:35-40);
 *     the first play() reads amplitudes[size] (one past the end: 0 here, as the guard of every sample bank);
 *   - then getNumBins() - (len % getNumBins()) zeros are fed (:41-45): whether that completes a last frame
 *     depends on len (the count uses bins, not fftsize: a partial last frame is usually dropped);
 *   - every frame's real/imag spectra are divided by the largest positive real / imaginary value seen (:47-52).
 * play(w) (:76-107): inFFT (same setup) takes w; on a new frame the spectrum is pushed to the front of the delay
 * line and sumReal/sumImag = sum over k of impulse[k] (x) FDL[k] (bin 0: real*real and imag*imag only, :86-87),
 * accumulated in float in k order; then ifft.process(sumReal, sumImag, COMPLEX) every sample.
 * maxiIFFT in COMPLEX mode (L/fft.cpp:613-619) copies real/imag into out_real/out_img, which calcIFFT then
 * OVERWRITES with the inverse transform of in_real/in_img -- never written in this mode, zeros from setup() -- so
 * the reference convolver returns silence.  mode 0 reproduces that; mode 1 ("as intended") routes the sums to
 * the transform inputs (negative frequencies zero, like polToCart) and is otherwise the same maxiIFFT
 * (setup(fftsize, fftsize, hopsize): hop = fftsize, Hann over the first `hopsize` samples, zero beyond).
 * pcm: the 16-bit impulse.  Returns the number of impulse frames (<0 on error); imp_real/imp_imag (optional)
 * receive them [frames][bins], at most cap_frames.
 * ------------------------------------------------------------------------------------ */
long mxo_convolve(const int16_t *pcm, size_t len, int fftsize, int hopsize, const float *in, size_t n, float *out,
                  int mode, float *imp_real, float *imp_imag, size_t cap_frames) {
    if (fftsize < 4 || (fftsize & (fftsize - 1)) || hopsize <= 0 || hopsize > fftsize || len == 0) return -1;
    const int bins = fftsize / 2, F = fftsize;
    double *amp = (double *)calloc(len + 1, sizeof(double)); /* + the element one past the end */
    for (size_t i = 0; i < len; i++) amp[i] = pcm[i] / 32767.0;
    float *window = (float *)calloc(F, sizeof(float));
    for (int i = 0; i < F; i++) window[i] = 0.50 - 0.50 * cos(2 * M_PI * i / (F - 1)); /* windowSize = fftsize */
    float *buffer = (float *)calloc(F, sizeof(float)), *in_real = (float *)calloc(F, sizeof(float));
    float *o_re = (float *)calloc(F, sizeof(float)), *o_im = (float *)calloc(F, sizeof(float));
    /* ---- analyseImpulse ---- */
    size_t frames_cap = len / F + 2, nfr = 0;
    float *iR = (float *)calloc(frames_cap * bins, sizeof(float)), *iI = (float *)calloc(frames_cap * bins, sizeof(float));
    float maxReal = 0, maxImag = 0;
    double position = (double)len; /* after load()/read() */
    int pos = 0;                   /* windowSize - hopSize = 0 */
    size_t total = len + (size_t)(bins - (int)(len % (size_t)bins));
    for (size_t s = 0; s < total; s++) {
        float v = 0;
        if (s < len) { /* impulse.play() */
            v = (float)amp[(long)position];
            position++;
            if ((long)position >= (long)len) position = 0;
        }
        buffer[pos++] = v;
        if (pos == F) {
            for (int i = 0; i < F; i++) in_real[i] = buffer[i] * window[i];
            fft_real(F, in_real, o_re, o_im);
            for (int i = 0; i < bins; i++) {
                iR[nfr * bins + i] = o_re[i];
                if (o_re[i] > maxReal) maxReal = o_re[i];
                iI[nfr * bins + i] = o_im[i];
                if (o_im[i] > maxImag) maxImag = o_im[i];
            }
            nfr++;
            pos = 0;
        }
    }
    for (size_t i = 0; i < nfr * (size_t)bins; i++) {
        iR[i] /= maxReal;
        iI[i] /= maxImag;
    }
    if (imp_real) memcpy(imp_real, iR, sizeof(float) * bins * (nfr < cap_frames ? nfr : cap_frames));
    if (imp_imag) memcpy(imp_imag, iI, sizeof(float) * bins * (nfr < cap_frames ? nfr : cap_frames));
    /* ---- play ---- */
    float *fdlR = (float *)calloc((nfr ? nfr : 1) * bins, sizeof(float)), *fdlI = (float *)calloc((nfr ? nfr : 1) * bins, sizeof(float));
    float *sumR = (float *)calloc(bins, sizeof(float)), *sumI = (float *)calloc(bins, sizeof(float));
    float *iwin = (float *)calloc(F, sizeof(float));
    for (int i = 0; i < hopsize; i++) iwin[i] = 0.50 - 0.50 * cos(2 * M_PI * i / (hopsize - 1)); /* maxiIFFT window */
    float *t_re = (float *)calloc(F, sizeof(float)), *t_im = (float *)calloc(F, sizeof(float));
    float *r_re = (float *)calloc(F, sizeof(float)), *r_im = (float *)calloc(F, sizeof(float));
    float *obuf = (float *)calloc(F, sizeof(float));
    size_t head = 0; /* ring: logical FDL[k] = row (head + k) % nfr */
    int ipos = 0, opos = 0;
    memset(buffer, 0, sizeof(float) * F);
    for (size_t s = 0; s < n; s++) {
        buffer[ipos++] = in[s];
        if (ipos == F) {
            ipos = 0;
            for (int i = 0; i < F; i++) in_real[i] = buffer[i] * window[i];
            fft_real(F, in_real, o_re, o_im);
            if (nfr) { /* push_front + pop_back */
                head = (head + nfr - 1) % nfr;
                memcpy(fdlR + head * bins, o_re, sizeof(float) * bins);
                memcpy(fdlI + head * bins, o_im, sizeof(float) * bins);
            }
            for (int i = 0; i < bins; i++) sumR[i] = sumI[i] = 0;
            for (size_t k = 0; k < nfr; k++) {
                const float *ir = iR + k * bins, *ii = iI + k * bins;
                const float *fr = fdlR + ((head + k) % nfr) * bins, *fi = fdlI + ((head + k) % nfr) * bins;
                sumR[0] += (ir[0] * fr[0]);
                sumI[0] += (ii[0] * fi[0]);
                for (int i = 1; i < bins; i++) {
                    sumR[i] += (ir[i] * fr[i]) - (ii[i] * fi[i]);
                    sumI[i] += (ir[i] * fi[i]) + (ii[i] * fr[i]);
                }
            }
        }
        if (opos == 0) { /* maxiIFFT::process, pos == 0 */
            if (mode == 1) {
                for (int i = 0; i < bins; i++) { t_re[i] = sumR[i]; t_im[i] = sumI[i]; }
            } /* mode 0: the transform inputs stay as setup() left them: zeros */
            fft_complex_dir(F, 1, t_re, t_im, r_re, r_im);
            /* hop = fftsize: the shift moves nothing, the whole buffer is cleared, then += ifftOut (0 + out*window) */
            for (int i = 0; i < F; i++) obuf[i] = 0.0f + (0.0f + r_re[i] * iwin[i]);
        }
        out[s] = obuf[opos];
        if (++opos == F) opos = 0;
    }
    free(amp); free(window); free(buffer); free(in_real); free(o_re); free(o_im); free(iR); free(iI);
    free(fdlR); free(fdlI); free(sumR); free(sumI); free(iwin); free(t_re); free(t_im); free(r_re); free(r_im); free(obuf);
    return (long)nfr;
}

/* ------------------------------------------------------------------------------------
 * maxiDCBlocker::play H:1261-1266; maxiSVF::setParams H:1320-1332, play H:1303-1317;
 * maxiBiquad::set H:1376-1478, play H:1360-1367.  PI = H:55 (3.1415926535897932384626433832795).
 * Layouts as oracle/ref_harness.cpp (mxo_filter2).
 * ------------------------------------------------------------------------------------ */
#define MXO_PI 3.1415926535897932384626433832795
void mxo_svf_coeffs(double freq, double res, double *c5) {
    double g = tan(MXO_PI * freq / g_sampleRate);
    double damping = res == 0 ? 0 : 1.0 / res;
    double k = damping;
    double ginv = g / (1.0 + g * (g + k));
    c5[0] = ginv;                   /* g1 */
    c5[1] = 2.0 * (g + k) * ginv;   /* g2 */
    c5[2] = g * ginv;               /* g3 */
    c5[3] = 2.0 * ginv;             /* g4 */
    c5[4] = k;
}

int mxo_biquad_coeffs(int filtType, double cutoff, double Q, double peakGain, double *c5) {
    double a0 = 0, a1 = 0, a2 = 0, b1 = 0, b2 = 0; /* members keep 0 for an unknown type */
    const double SQRT2 = sqrt(2.0);
    double norm = 0;
    double V = pow(10.0, fabs(peakGain) / 20.0); /* `abs` resolves to the double overload (pinned vs _ref) */
    double K = tan(MXO_PI * cutoff / g_sampleRate);
    switch (filtType) {
        case 0: /* LOWPASS */
            norm = 1.0 / (1.0 + K / Q + K * K);
            a0 = K * K * norm; a1 = 2.0 * a0; a2 = a0;
            b1 = 2.0 * (K * K - 1.0) * norm; b2 = (1.0 - K / Q + K * K) * norm;
            break;
        case 1: /* HIGHPASS */
            norm = 1. / (1. + K / Q + K * K);
            a0 = 1 * norm; a1 = -2 * a0; a2 = a0;
            b1 = 2 * (K * K - 1) * norm; b2 = (1 - K / Q + K * K) * norm;
            break;
        case 2: /* BANDPASS */
            norm = 1. / (1. + K / Q + K * K);
            a0 = K / Q * norm; a1 = 0.; a2 = -a0;
            b1 = 2. * (K * K - 1.) * norm; b2 = (1. - K / Q + K * K) * norm;
            break;
        case 3: /* NOTCH */
            norm = 1. / (1. + K / Q + K * K);
            a0 = (1. + K * K) * norm; a1 = 2. * (K * K - 1.) * norm; a2 = a0;
            b1 = a1; b2 = (1. - K / Q + K * K) * norm;
            break;
        case 4: /* PEAK */
            if (peakGain >= 0.0) {
                norm = 1. / (1. + 1. / Q * K + K * K);
                a0 = (1. + V / Q * K + K * K) * norm; a1 = 2. * (K * K - 1.) * norm;
                a2 = (1. - V / Q * K + K * K) * norm; b1 = a1; b2 = (1. - 1. / Q * K + K * K) * norm;
            } else {
                norm = 1. / (1. + V / Q * K + K * K);
                a0 = (1. + 1 / Q * K + K * K) * norm; a1 = 2. * (K * K - 1) * norm;
                a2 = (1. - 1. / Q * K + K * K) * norm; b1 = a1; b2 = (1. - V / Q * K + K * K) * norm;
            }
            break;
        case 5: /* LOWSHELF */
            if (peakGain >= 0.) {
                norm = 1. / (1. + SQRT2 * K + K * K);
                a0 = (1. + sqrt(2. * V) * K + V * K * K) * norm; a1 = 2. * (V * K * K - 1.) * norm;
                a2 = (1. - sqrt(2. * V) * K + V * K * K) * norm; b1 = 2. * (K * K - 1.) * norm;
                b2 = (1. - SQRT2 * K + K * K) * norm;
            } else {
                norm = 1. / (1. + sqrt(2. * V) * K + V * K * K);
                a0 = (1. + SQRT2 * K + K * K) * norm; a1 = 2. * (K * K - 1.) * norm;
                a2 = (1. - SQRT2 * K + K * K) * norm; b1 = 2. * (V * K * K - 1.) * norm;
                b2 = (1. - sqrt(2. * V) * K + V * K * K) * norm;
            }
            break;
        case 6: /* HIGHSHELF */
            if (peakGain >= 0.) {
                norm = 1. / (1. + SQRT2 * K + K * K);
                a0 = (V + sqrt(2. * V) * K + K * K) * norm; a1 = 2. * (K * K - V) * norm;
                a2 = (V - sqrt(2. * V) * K + K * K) * norm; b1 = 2. * (K * K - 1) * norm;
                b2 = (1. - SQRT2 * K + K * K) * norm;
            } else {
                norm = 1. / (V + sqrt(2. * V) * K + K * K);
                a0 = (1. + SQRT2 * K + K * K) * norm; a1 = 2. * (K * K - 1.) * norm;
                a2 = (1. - SQRT2 * K + K * K) * norm; b1 = 2. * (K * K - V) * norm;
                b2 = (V - sqrt(2. * V) * K + K * K) * norm;
            }
            break;
        default: return -1;
    }
    c5[0] = a0; c5[1] = a1; c5[2] = a2; c5[3] = b1; c5[4] = b2;
    return 0;
}

int mxo_filter2(int kind, size_t V, size_t N, const double *in, const double *par, double *st, double *coef,
                double *out) {
    if (kind < 0 || kind > 2) return -1;
    for (size_t v = 0; v < V; v++) {
        double c[5] = {0, 0, 0, 0, 0};
        if (kind == 0) {
            double xm1 = st[v], ym1 = st[V + v];
            const double R = par[v];
            for (size_t n = 0; n < N; n++) {
                const double input = in[n * V + v];
                ym1 = input - xm1 + R * ym1;
                xm1 = input;
                out[n * V + v] = ym1;
            }
            st[v] = xm1; st[V + v] = ym1;
        } else if (kind == 1) {
            mxo_svf_coeffs(par[v], par[V + v], c);
            const double g1 = c[0], g2 = c[1], g3 = c[2], g4 = c[3], k = c[4];
            const double lpmix = par[2 * V + v], bpmix = par[3 * V + v], hpmix = par[4 * V + v], notchmix = par[5 * V + v];
            double v0z = st[v], v1 = st[V + v], v2 = st[2 * V + v];
            for (size_t n = 0; n < N; n++) {
                const double w = in[n * V + v];
                double low, band, high, notch;
                double v1z = v1;
                double v2z = v2;
                double v3 = w + v0z - 2.0 * v2z;
                v1 += g1 * v3 - g2 * v1z;
                v2 += g3 * v3 + g4 * v1z;
                v0z = w;
                low = v2;
                band = v1;
                high = w - k * v1 - v2;
                notch = w - k * v1;
                out[n * V + v] = (low * lpmix) + (band * bpmix) + (high * hpmix) + (notch * notchmix);
            }
            st[v] = v0z; st[V + v] = v1; st[2 * V + v] = v2;
        } else {
            if (mxo_biquad_coeffs((int)par[v], par[V + v], par[2 * V + v], par[3 * V + v], c)) return -2;
            const double a0 = c[0], a1 = c[1], a2 = c[2], b1 = c[3], b2 = c[4];
            double v0 = st[v], v1 = st[V + v], v2 = st[2 * V + v];
            for (size_t n = 0; n < N; n++) {
                const double input = in[n * V + v];
                v0 = input - (b1 * v1) - (b2 * v2);
                double y = (a0 * v0) + (a1 * v1) + (a2 * v2);
                v2 = v1;
                v1 = v0;
                out[n * V + v] = y;
            }
            st[v] = v0; st[V + v] = v1; st[2 * V + v] = v2;
        }
        if (coef && kind > 0)
            for (int r = 0; r < 5; r++) coef[r * V + v] = c[r];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiEnvGen H:2268-2547: setup H:2366-2399 + setupSegmentTime H:2531-2545 (stage table), play
 * H:2277-2354 (switch with fall-through WAITING -> TRIGGERED -> HOLDING), reset H:2402-2410,
 * resetAndArm H:2413-2416, maxiMap::linlin H:801-805, maxiTrigger::onZX H:569-579.
 * Only stages[phase] ever holds a non-zero counter/currentlevel (every exit from a stage zeroes
 * them), so the per-voice state is (counter, currentlevel) of the current stage.
 * Layouts as oracle/ref_harness.cpp (mxo_envgen).  HOLD = -46692 (H:2271).
 * ------------------------------------------------------------------------------------ */
#define MXO_ENVGEN_HOLD (-46692.0)
typedef struct {
    double startlevel, endlevel, gradient, curve;
    size_t length;
    int hold;
} envstage_t;

/* returns the number of stages, -2 if a second HOLD stage is found (setup returns 0) */
int mxo_envgen_stages(size_t nlevels, const double *levels, const double *times, const double *curves,
                      double *stages6) {
    double accumulatedTime = 0;
    int containsHold = 0;
    for (size_t i = 0; i + 1 < nlevels; i++) {
        double stageTime = times[i];
        double length, gradient, hold;
        if (stageTime == MXO_ENVGEN_HOLD) {
            if (containsHold) return -2;
            length = 0; hold = 1; gradient = 0; containsHold = 1;
        } else {
            double len = ((stageTime / 1000.0) * g_sampleRate) + accumulatedTime;
            size_t l = (size_t)(floor(len));
            accumulatedTime = len - l;
            gradient = 1.0 / l;
            length = (double)l; hold = 0;
        }
        stages6[i * 6 + 0] = levels[i]; stages6[i * 6 + 1] = levels[i + 1]; stages6[i * 6 + 2] = gradient;
        stages6[i * 6 + 3] = curves[i]; stages6[i * 6 + 4] = length; stages6[i * 6 + 5] = hold;
    }
    return (int)(nlevels - 1);
}

int mxo_envgen(size_t V, size_t N, const double *trig, int tpv, size_t nlevels, const double *levels,
               const double *times, const double *curves, int loop, int retrigger, double *dst, int64_t *ist,
               double *stages_out, double *out) {
    if (nlevels < 2) return -1;
    const size_t S = nlevels - 1;
    double *tab = (double *)malloc(sizeof(double) * 6 * S);
    if (mxo_envgen_stages(nlevels, levels, times, curves, tab) < 0) { free(tab); return -2; }
    if (stages_out) memcpy(stages_out, tab, sizeof(double) * 6 * S);
    enum { WAITING = 0, TRIGGERED = 1, HOLDING = 2 };
    for (size_t v = 0; v < V; v++) {
        double envval = dst[v], currentlevel = dst[V + v];
        size_t phase = (size_t)ist[v], counter = (size_t)ist[3 * V + v];
        int state = (int)ist[V + v], nxcHappened = ist[2 * V + v] != 0;
        zx_t trigDetector = {dst[2 * V + v], ist[4 * V + v] != 0};
        zx_t holdDetector = {dst[3 * V + v], ist[5 * V + v] != 0};
        zx_t retriggerDetector = {dst[4 * V + v], ist[6 * V + v] != 0};
        for (size_t n = 0; n < N; n++) {
            const double trigger = tpv ? trig[n * V + v] : trig[n];
            int stage_entry = state; /* emulate the fall-through with an explicit cursor */
            if (stage_entry == WAITING) {
                if (zx_onZX(&trigDetector, trigger) != 0.0) {
                    state = TRIGGERED; /* stages.size() > 0 always here */
                    nxcHappened = 0;
                    stage_entry = TRIGGERED;
                } else {
                    stage_entry = -1; /* break */
                }
            }
            if (stage_entry == TRIGGERED) {
                const double *cs = tab + 6 * phase;
                if (zx_onZX(&holdDetector, -trigger) != 0.0) nxcHappened = 1;
                if (cs[5] != 0) {
                    state = HOLDING;
                    stage_entry = HOLDING; /* falls through */
                } else {
                    double val = pow(currentlevel, cs[3]);
                    /* linlin(val, 0, 1, startlevel, endlevel) H:801-805 */
                    val = (1.0 < val) ? 1.0 : val; /* min(val, inMax) */
                    val = (val < 0.0) ? 0.0 : val; /* max(., inMin) */
                    envval = ((val - 0.0) / (1.0 - 0.0) * (cs[1] - cs[0])) + cs[0];
                    counter++;
                    if (counter == (size_t)cs[4]) {
                        counter = 0;
                        currentlevel = 0;
                        phase++;
                    } else {
                        currentlevel += cs[2];
                    }
                    if (retrigger) {
                        if (zx_onZX(&retriggerDetector, trigger) != 0.0) {
                            nxcHappened = 0;
                            counter = 0; currentlevel = 0; phase = 0; state = TRIGGERED; /* reset() */
                        }
                    }
                    stage_entry = -1;
                }
            }
            if (stage_entry == HOLDING) {
                if (zx_onZX(&holdDetector, -trigger) != 0.0) nxcHappened = 1;
                if (nxcHappened) {
                    state = TRIGGERED;
                    phase++;
                }
                if (retrigger) {
                    if (zx_onZX(&retriggerDetector, trigger) != 0.0) {
                        nxcHappened = 0;
                        counter = 0; currentlevel = 0; phase = 0; state = TRIGGERED;
                    }
                }
            }
            if (phase == S) {
                counter = 0; currentlevel = 0; /* stages[phase] does not exist: reset() only rewinds */
                phase = 0;
                state = loop ? TRIGGERED : WAITING; /* reset() / resetAndArm() */
            }
            out[n * V + v] = envval;
        }
        dst[v] = envval; dst[V + v] = phase < S ? currentlevel : 0.0;
        ist[v] = (int64_t)phase; ist[V + v] = state; ist[2 * V + v] = nxcHappened;
        ist[3 * V + v] = phase < S ? (int64_t)counter : 0;
        dst[2 * V + v] = trigDetector.previousValue; ist[4 * V + v] = trigDetector.firstTrigger;
        dst[3 * V + v] = holdDetector.previousValue; ist[5 * V + v] = holdDetector.firstTrigger;
        dst[4 * V + v] = retriggerDetector.previousValue; ist[6 * V + v] = retriggerDetector.firstTrigger;
    }
    free(tab);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * maxiSampler::play L/maxiSynths.cpp:289-312 (class L/maxiSynths.h:137-187): per voice
 *   envOut = envelopes[i].adsr(envOutGain[i], envelopes[i].trigger);
 *   if (envOut > 0) { outputs[i] = samples[i].play4(pitchRatios[(int)pitch[i]+originalPitch] *
 *       ((1./len)*sampleRate), 0, len) * envOut;  output += outputs[i]/voices;
 *       if (trigger == 1 && !sustain) trigger = 0; }
 * originalPitch = 67 (H:152), pitchRatios src/maximilian.h:112.  Layout as oracle/ref_harness.cpp.
 * ------------------------------------------------------------------------------------ */
double mxo_sampler_frequency(double pitch, size_t len) {
    return MAXI_PITCH_RATIOS[(int)pitch + 67] * ((1. / len) * g_sampleRate);
}

int mxo_sampler(size_t NS, int voices, size_t N, const double *amp, size_t len, int sustain, const double *pitch,
                const double *gain, const double *par, const int64_t *holdtime, double *position,
                int32_t *trigger, double *outhold, double *dst, int64_t *ist, double *mix, double *outputs) {
    if (voices < 1 || voices > 32) return -1;
    const size_t V = NS * (size_t)voices;
    for (size_t s = 0; s < NS; s++) {
        env_t e[32];
        smp_t sm[32];
        int trig[32];
        double outs[32];
        for (int i = 0; i < voices; i++) {
            const size_t v = s * voices + i;
            env_load(&e[i], V, v, par, holdtime, dst, ist);
            sm[i].amp = amp; sm[i].len = len; sm[i].mySampleRate = 44100; sm[i].position = position[v];
            trig[i] = trigger[v];
            outs[i] = outhold[v];
        }
        for (size_t n = 0; n < N; n++) {
            double output = 0;
            for (int i = 0; i < voices; i++) {
                const size_t v = s * voices + i;
                double envOut = env_adsr(&e[i], gain[v], trig[i]);
                if (envOut > 0.) {
                    outs[i] = smp_play4(&sm[i], mxo_sampler_frequency(pitch[v], len), 0, len) * envOut;
                    output += outs[i] / voices;
                    if (trig[i] == 1 && !sustain) trig[i] = 0;
                }
                if (outputs) outputs[n * V + v] = outs[i];
            }
            mix[n * NS + s] = output;
        }
        for (int i = 0; i < voices; i++) {
            const size_t v = s * voices + i;
            env_store(&e[i], V, v, dst, ist);
            position[v] = sm[i].position;
            trigger[v] = trig[i];
            outhold[v] = outs[i];
        }
    }
    return 0;
}
