// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" driver around the UNMODIFIED reference classes.  It is compiled
// together with the reference's own sources where they lie under /root/reference
// (see oracle/Makefile, target _ref/libmaxiref.so); no reference source is copied
// into this repository.  Built with -fno-access-control so the harness can seed
// and read back the classes' private state (phase, x, y, amplitude ...), which is
// how block-to-block state carry is compared against the HIP banks.
//
// Every entry point renders a *bank* the way a user's play() would: for each
// sample n, for each voice v, call the reference's per-sample method once
// (loop order of cpp/commandline/maximilian_examples/15.polysynth/main.cpp:54-70).
// Output layout is sample-major / voice-minor: out[n*V + v].
//
// The same symbol names (mxo_*) are exported by oracle/maxi_oracle.c (the plain-C
// restatement), so tests drive both through one ctypes signature table.
#include "maximilian.h"
#include "libs/maxiFFT.h"
#include "libs/maxiMFCC.h"
#include "libs/maxiSynths.h"
#include "libs/maxiConvolve.h"
// maxiTimeStretch/maxiStretch draw `rand() % 10` from the process-wide libc stream
// (maxiGrains.h:352, :524), which cannot be reproduced per stream in a bank.  The header-only
// grain code is compiled in THIS translation unit, so its rand() calls are routed to a
// per-stream queue supplied by the test (the reference source on disk is untouched).
#include <cstdlib>
static const int32_t *g_rnd_q = nullptr;
static size_t g_rnd_n = 0, g_rnd_i = 0;
static int g_rnd_underrun = 0;
static int mxo_ref_rand() {
    if (!g_rnd_q) return 0;
    if (g_rnd_i >= g_rnd_n) {
        g_rnd_underrun = 1;
        return 0;
    }
    return g_rnd_q[g_rnd_i++];
}
#define rand mxo_ref_rand
#include "libs/maxiGrains.h"
#undef rand

#include <cstdint>
#include <cstdio>
#include <unistd.h>
#include <cstring>
#include <iostream>
#include <memory>
#include <thread>
#include <vector>
#include <chrono>

extern double sineBuffer[514];
extern double transition[1001];

namespace {
inline double fq(const double *freq, int fps, size_t n, size_t v, size_t V) {
    return fps ? freq[n * V + v] : freq[v];
}
}  // namespace

extern "C" {

const char *mxo_kind(void) { return "reference"; }

void mxo_settings(size_t sr, size_t ch, size_t buf) { maxiSettings::setup(sr, ch, buf); }

// Raw tables of the reference (src/maximilian.cpp:63, :67-200), for oracle/gen_tables.py.
const double *mxo_sine_table(void) { return sineBuffer; }
const double *mxo_transition_table(void) { return transition; }
const double *mxo_pitch_ratios(void) { return pitchRatios; }  // src/maximilian.h:112
extern double mtofarray[129];                                  // src/maximilian.cpp:203
const double *mxo_mtof_table(void) { return mtofarray; }       // what maxiConvert::mtof indexes (maximilian.cpp:1498-1500)
// The two out-of-bounds neighbours the reference reads (sinebuf4 -> sineBuffer[-1],
// sawn -> transition[1001]); exported so the tests can assert what this build holds.
double mxo_sine_table_guard(void) { return (&sineBuffer[0])[-1]; }
double mxo_transition_guard(void) { return (&transition[0])[1001]; }

// ---- maxiOsc bank (src/maximilian.cpp:209-373) -------------------------------------
// wf: 0 sinewave 1 coswave 2 phasor 3 saw 4 triangle 5 square 6 pulse 7 impulse
//     8 sinebuf 9 sinebuf4 10 sawn 11 phasorBetween
int mxo_osc(int wf, size_t V, size_t N, const double *freq, int fps, const double *p1,
            const double *p2, double *phase, double *outhold, double *out) {
    std::vector<maxiOsc> bank(V);
    for (size_t v = 0; v < V; v++) {
        bank[v].phase = phase[v];
        bank[v].output = outhold[v];
    }
    for (size_t n = 0; n < N; n++) {
        double *row = out + n * V;
        for (size_t v = 0; v < V; v++) {
            maxiOsc &o = bank[v];
            double f = fq(freq, fps, n, v, V);
            double r;
            switch (wf) {
                case 0: r = o.sinewave(f); break;
                case 1: r = o.coswave(f); break;
                case 2: r = o.phasor(f); break;
                case 3: r = o.saw(f); break;
                case 4: r = o.triangle(f); break;
                case 5: r = o.square(f); break;
                case 6: r = o.pulse(f, p1[v]); break;
                case 7: r = o.impulse(f); break;
                case 8: r = o.sinebuf(f); break;
                case 9: r = o.sinebuf4(f); break;
                case 10: r = o.sawn(f); break;
                case 11: r = o.phasorBetween(f, p1[v], p2[v]); break;
                default: return -1;
            }
            row[v] = r;
        }
    }
    for (size_t v = 0; v < V; v++) {
        phase[v] = bank[v].phase;
        outhold[v] = bank[v].output;
    }
    return 0;
}

// ---- maxiFilter bank (src/maximilian.cpp:442-500) ----------------------------------
// kind: 0 lores 1 hires 2 bandpass 3 lopass 4 hipass.  st = [5][V]: x, y, outputs[0..2]
int mxo_filter(int kind, size_t V, size_t N, const double *in, const double *cutoff, int cps,
               const double *res, int rps, double *st, double *out) {
    std::vector<maxiFilter> bank(V);
    for (size_t v = 0; v < V; v++) {
        maxiFilter &f = bank[v];
        f.x = st[0 * V + v];
        f.y = st[1 * V + v];
        f.outputs[0] = st[2 * V + v];
        f.outputs[1] = st[3 * V + v];
        f.outputs[2] = st[4 * V + v];
    }
    for (size_t n = 0; n < N; n++) {
        for (size_t v = 0; v < V; v++) {
            maxiFilter &f = bank[v];
            double x = in[n * V + v];
            double c = cps ? cutoff[n * V + v] : cutoff[v];
            double r = res ? (rps ? res[n * V + v] : res[v]) : 0.0;
            double o;
            switch (kind) {
                case 0: o = f.lores(x, c, r); break;
                case 1: o = f.hires(x, c, r); break;
                case 2: o = f.bandpass(x, c, r); break;
                case 3: o = f.lopass(x, c); break;
                case 4: o = f.hipass(x, c); break;
                default: return -1;
            }
            out[n * V + v] = o;
        }
    }
    for (size_t v = 0; v < V; v++) {
        maxiFilter &f = bank[v];
        st[0 * V + v] = f.x;
        st[1 * V + v] = f.y;
        st[2 * V + v] = f.outputs[0];
        st[3 * V + v] = f.outputs[1];
        st[4 * V + v] = f.outputs[2];
    }
    return 0;
}

// The coefficients the hoisted (block-constant) lores/hires/bandpass path uploads: obtained by
// running the reference filter once on a scratch object and reading back its members
// (lores/hires: c and r re-derived exactly as C:459-461; bandpass: inputs[0..2], C:492-495).
// coef = [3][V].
void mxo_filter_coeffs(int kind, size_t V, const double *cutoff, const double *res, double *coef) {
    for (size_t v = 0; v < V; v++) {
        if (kind == 2) {
            maxiFilter f;
            f.outputs[1] = f.outputs[2] = 0;
            f.bandpass(0.0, cutoff[v], res[v]);
            coef[v] = f.inputs[0];
            coef[V + v] = f.inputs[1];
            coef[2 * V + v] = f.inputs[2];
        } else {
            double cut = cutoff[v], resonance = res[v];
            if (cut < 10) cut = 10;
            if (cut > (maxiSettings::sampleRate)) cut = (maxiSettings::sampleRate);
            if (resonance < 1.) resonance = 1.;
            double z = cos(TWOPI * cut / maxiSettings::sampleRate);
            coef[v] = 2 - 2 * z;
            coef[V + v] = (sqrt(2.0) * sqrt(-pow((z - 1.0), 3.0)) + resonance * (z - 1)) /
                          (resonance * (z - 1));
            coef[2 * V + v] = 0.0;
            // cross-check against the member the reference itself stores (c, H:303)
            maxiFilter f;
            f.lores(0.0, cutoff[v], res[v]);
            if (f.c != coef[v]) coef[v] = NAN;
        }
    }
}

// ---- maxiEnv bank (src/maximilian.cpp:1319-1494) -----------------------------------
// par = [4][V]: attack, decay, sustain, release (as the setters would have stored them)
// dst = [2][V]: amplitude, output.   ist = [6][V]: holdcount, attackphase, decayphase,
// sustainphase, holdphase, releasephase.  trig: [N] (tpv=0) or [N][V] (tpv=1).
// in == NULL -> constant 1.0 input (examples 14/15 drive adsr(1., trigger)).
// mode 0: adsr   mode 1: ar (decay/sustain ignored)
int mxo_env(int mode, size_t V, size_t N, const double *in, const int32_t *trig, int tpv,
            const double *par, const int64_t *holdtime, double *dst, int64_t *ist, double *out) {
    std::vector<maxiEnv> bank(V);
    for (size_t v = 0; v < V; v++) {
        maxiEnv &e = bank[v];
        e.attack = par[0 * V + v];
        e.decay = par[1 * V + v];
        e.sustain = par[2 * V + v];
        e.release = par[3 * V + v];
        e.holdtime = holdtime[v];
        e.amplitude = dst[0 * V + v];
        e.output = dst[1 * V + v];
        e.holdcount = ist[0 * V + v];
        e.attackphase = (int)ist[1 * V + v];
        e.decayphase = (int)ist[2 * V + v];
        e.sustainphase = (int)ist[3 * V + v];
        e.holdphase = (int)ist[4 * V + v];
        e.releasephase = (int)ist[5 * V + v];
    }
    for (size_t n = 0; n < N; n++) {
        for (size_t v = 0; v < V; v++) {
            maxiEnv &e = bank[v];
            double x = in ? in[n * V + v] : 1.0;
            int t = tpv ? trig[n * V + v] : trig[n];
            double o;
            if (mode == 0)
                o = e.adsr(x, t);
            else
                o = e.ar(x, e.attack, e.release, e.holdtime, t);
            out[n * V + v] = o;
        }
    }
    for (size_t v = 0; v < V; v++) {
        maxiEnv &e = bank[v];
        dst[0 * V + v] = e.amplitude;
        dst[1 * V + v] = e.output;
        ist[0 * V + v] = e.holdcount;
        ist[1 * V + v] = e.attackphase;
        ist[2 * V + v] = e.decayphase;
        ist[3 * V + v] = e.sustainphase;
        ist[4 * V + v] = e.holdphase;
        ist[5 * V + v] = e.releasephase;
    }
    return 0;
}

// maxiEnv setters (src/maximilian.cpp:1469-1494): ms -> stored coefficient.
// which: 0 setAttack 1 setDecay 2 setRelease 3 setAttackMS
double mxo_env_coeff(int which, double ms) {
    maxiEnv e{};
    switch (which) {
        case 0: e.setAttack(ms); return e.attack;
        case 1: e.setDecay(ms); return e.decay;
        case 2: e.setRelease(ms); return e.release;
        case 3: e.setAttackMS(ms); return e.attack;
    }
    return 0;
}

// ---- fused subtractive voice (config 3; shape of examples 14/15) --------------------
// mode 0 (A): out = adsr( lores( saw(freq), cutoff, res ), trig )
// mode 1 (B, 14.monosynth/main.cpp:50-55): e = adsr(1., trig);
//            out = lores( saw(freq), e*cutoff, res ) * e          (cutoff acts as the scale)
// ost=[2][V] osc phase/output, fst=[5][V] filter, env state as mxo_env.
int mxo_voice(int mode, size_t V, size_t N, const double *freq, const double *cutoff,
              const double *res, const int32_t *trig, int tpv, const double *par,
              const int64_t *holdtime, double *ost, double *fst, double *dst, int64_t *ist,
              double *out) {
    std::vector<maxiOsc> osc(V);
    std::vector<maxiFilter> flt(V);
    std::vector<maxiEnv> env(V);
    for (size_t v = 0; v < V; v++) {
        osc[v].phase = ost[v];
        osc[v].output = ost[V + v];
        maxiFilter &f = flt[v];
        f.x = fst[0 * V + v];
        f.y = fst[1 * V + v];
        f.outputs[0] = fst[2 * V + v];
        f.outputs[1] = fst[3 * V + v];
        f.outputs[2] = fst[4 * V + v];
        maxiEnv &e = env[v];
        e.attack = par[0 * V + v];
        e.decay = par[1 * V + v];
        e.sustain = par[2 * V + v];
        e.release = par[3 * V + v];
        e.holdtime = holdtime[v];
        e.amplitude = dst[0 * V + v];
        e.output = dst[1 * V + v];
        e.holdcount = ist[0 * V + v];
        e.attackphase = (int)ist[1 * V + v];
        e.decayphase = (int)ist[2 * V + v];
        e.sustainphase = (int)ist[3 * V + v];
        e.holdphase = (int)ist[4 * V + v];
        e.releasephase = (int)ist[5 * V + v];
    }
    for (size_t n = 0; n < N; n++) {
        for (size_t v = 0; v < V; v++) {
            int t = tpv ? trig[n * V + v] : trig[n];
            double o;
            if (mode == 0) {
                double s = osc[v].saw(freq[v]);
                double f = flt[v].lores(s, cutoff[v], res[v]);
                o = env[v].adsr(f, t);
            } else {
                double e = env[v].adsr(1.0, t);
                double s = osc[v].saw(freq[v]);
                double f = flt[v].lores(s, e * cutoff[v], res[v]);
                o = f * e;
            }
            out[n * V + v] = o;
        }
    }
    for (size_t v = 0; v < V; v++) {
        ost[v] = osc[v].phase;
        ost[V + v] = osc[v].output;
        maxiFilter &f = flt[v];
        fst[0 * V + v] = f.x;
        fst[1 * V + v] = f.y;
        fst[2 * V + v] = f.outputs[0];
        fst[3 * V + v] = f.outputs[1];
        fst[4 * V + v] = f.outputs[2];
        maxiEnv &e = env[v];
        dst[0 * V + v] = e.amplitude;
        dst[1 * V + v] = e.output;
        ist[0 * V + v] = e.holdcount;
        ist[1 * V + v] = e.attackphase;
        ist[2 * V + v] = e.decayphase;
        ist[3 * V + v] = e.sustainphase;
        ist[4 * V + v] = e.holdphase;
        ist[5 * V + v] = e.releasephase;
    }
    return 0;
}

// ---- maxiMix::stereo + user-side sum over voices (src/maximilian.cpp:503-509) --------
// mix[n][0..1] = sum_v (in[n][v]*sqrt(1-x_v), in[n][v]*sqrt(x_v)), summed in voice
// order exactly as a play() loop would (15.polysynth/main.cpp:67 pattern).
int mxo_mix_stereo(size_t V, size_t N, const double *in, const double *pan, double *mix) {
    maxiMix m;
    std::vector<double> two(2);
    for (size_t n = 0; n < N; n++) {
        double l = 0, r = 0;
        for (size_t v = 0; v < V; v++) {
            m.stereo(in[n * V + v], two, pan[v]);
            l += two[0];
            r += two[1];
        }
        mix[2 * n] = l;
        mix[2 * n + 1] = r;
    }
    return 0;
}

// ---- maxiDelayline bank (src/maximilian.cpp:415-439) ------------------------------------------
// The reference object embeds a 5.6 MB memory[] array (H:273); the harness keeps one object on
// the heap per voice and mirrors the first `cap` slots to/from mem[slot*V + v].
int mxo_delay(int mode, size_t V, size_t N, const double *in, const int32_t *size,
              const double *feedback, const int32_t *position, double *mem, size_t cap,
              int32_t *phase, double *out) {
    if (mode < 0 || mode > 1) return -1;
    for (size_t v = 0; v < V; v++) {
        if ((size_t)size[v] > cap) return -2;
        std::unique_ptr<maxiDelayline> d(new maxiDelayline());
        d->phase = phase[v];
        for (size_t k = 0; k < cap; k++) d->memory[k] = mem[k * V + v];
        for (size_t n = 0; n < N; n++) {
            out[n * V + v] = mode == 0 ? d->dl(in[n * V + v], size[v], feedback[v])
                                       : d->dlFromPosition(in[n * V + v], size[v], feedback[v], position[v]);
        }
        for (size_t k = 0; k < cap; k++) mem[k * V + v] = d->memory[k];
        phase[v] = d->phase;
    }
    return 0;
}

// ---- maxiSample play family (src/maximilian.cpp:740-1075) ----------------------------------------
// `amp` points at element 0 of a buffer valid on [-1, len+1] (guards 0.0).  Each voice gets
// its own maxiSample holding a copy of the data (setSample, H:670-678); the two elements past
// the end are made deterministic by reserving len+2 and zeroing them in place.
int mxo_sample(int mode, size_t V, size_t N, const double *amp, size_t len, int mySampleRate,
               const double *a, int aps, const double *start, const double *end, double *position,
               double *out) {
    if (mode < 0 || mode > 8) return -1;
    std::vector<double> data(amp, amp + len);
    for (size_t v = 0; v < V; v++) {
        maxiSample s;
        s.amplitudes.reserve(len + 2);
        s.setSample(data);
        s.amplitudes.data()[len] = 0.0;
        s.amplitudes.data()[len + 1] = 0.0;
        s.mySampleRate = mySampleRate;
        s.position = position[v];
        double st = start ? start[v] : 0.0, en = end ? end[v] : 1.0;
        for (size_t n = 0; n < N; n++) {
            double x = a ? (aps ? a[n * V + v] : a[v]) : 1.0;
            double o = 0;
            switch (mode) {
                case 0: o = s.play(); break;
                case 1: o = s.playOnce(); break;
                case 2: o = s.playLoop(st, en); break;
                case 3: o = s.playUntil(en); break;
                case 4: o = s.playAtSpeed(x); break;
                case 5: o = s.playOnceAtSpeed(x); break;
                case 6: o = s.playUntilAtSpeed(en, x); break;
                case 7: o = s.play4(x, st, en); break;
                case 8: o = s.playAtSpeedBetweenPoints(x, st, en); break;
            }
            out[n * V + v] = o;
        }
        position[v] = s.position;
    }
    return 0;
}

// ---- maxiFFT streamed over a signal (src/libs/maxiFFT.cpp:45-91) ---------------------------------
long mxo_fft_stream(const float *signal, size_t nsamples, int fftSize, int hopSize, int windowSize,
                    size_t max_frames, float *real, float *imag, float *mags, float *phases) {
    if (fftSize < 4 || (fftSize & (fftSize - 1))) return -1;
    int win = windowSize > fftSize ? windowSize : fftSize;
    if (win > fftSize || hopSize <= 0 || hopSize > win) return -2;  // would overrun the vectors
    maxiFFT f;
    f.setup(fftSize, hopSize, windowSize);
    const int bins = f.getNumBins();
    size_t frames = 0;
    for (size_t s = 0; s < nsamples; s++) {
        if (frames >= max_frames) break;
        if (f.process(signal[s], maxiFFT::WITH_POLAR_CONVERSION)) {
            std::vector<float> &m = f.getMagnitudes();
            std::vector<float> &p = f.getPhases();
            for (int i = 0; i < bins; i++) {
                if (mags) mags[frames * bins + i] = m[i];
                if (phases) phases[frames * bins + i] = p[i];
                if (real) real[frames * bins + i] = f.getReal()[i];
                if (imag) imag[frames * bins + i] = f.getImag()[i];
            }
            frames++;
        }
    }
    return (long)frames;
}

// fft::convToDB (src/libs/fft.cpp:526-534)
void mxo_fft_to_db(const float *in, float *out, size_t n) {
    fft f;
    f.setup((int)(2 * n));
    f.convToDB(const_cast<float *>(in), out);
}

// maxiFFT::spectralFlatness / spectralCentroid (src/libs/maxiFFT.cpp:113-132) over frames of magnitudes
int mxo_fft_features(const float *mags, size_t nframes, int fftSize, float *flatness, float *centroid) {
    if (fftSize < 4 || (fftSize & (fftSize - 1))) return -1;
    maxiFFT f;
    f.setup(fftSize, fftSize, fftSize);
    const int bins = f.getNumBins();
    for (size_t k = 0; k < nframes; k++) {
        for (int i = 0; i < bins; i++) f.magnitudes[i] = mags[k * bins + i];
        if (flatness) flatness[k] = f.spectralFlatness();
        if (centroid) centroid[k] = f.spectralCentroid();
    }
    return 0;
}

// ---- maxiMFCC (src/libs/maxiMFCC.h, maxiMFCC.cpp) ---------------------------------------------------
// The reference never writes column 0 of melFilters (loop starts at filter 1, maxiMFCC.h:149);
// the harness zeroes that column after setup() so the oracle is deterministic.
static void mfcc_setup(maxiMFCC &m, unsigned numBins, unsigned numFilters, unsigned numCoeffs,
                       double minFreq, double maxFreq) {
    m.setup(numBins, numFilters, numCoeffs, minFreq, maxFreq);
    for (unsigned bin = 0; bin < numBins; bin++) m.melFilters[0 + bin * numFilters] = 0.0;
}

int mxo_mfcc_tables(unsigned numBins, unsigned numFilters, unsigned numCoeffs, double minFreq,
                    double maxFreq, double *melFilters, double *dct) {
    maxiMFCC m;
    mfcc_setup(m, numBins, numFilters, numCoeffs, minFreq, maxFreq);
    memcpy(melFilters, m.melFilters, sizeof(double) * numFilters * numBins);
    memcpy(dct, m.dctMatrix, sizeof(double) * numCoeffs * numFilters);
    return 0;
}

int mxo_mfcc(unsigned numBins, unsigned numFilters, unsigned numCoeffs, double minFreq,
             double maxFreq, const float *mags, size_t mag_stride, size_t nframes, double *melbands,
             double *mfcc) {
    maxiMFCC m;
    mfcc_setup(m, numBins, numFilters, numCoeffs, minFreq, maxFreq);
    std::vector<float> spec(numBins);
    for (size_t f = 0; f < nframes; f++) {
        memcpy(spec.data(), mags + f * mag_stride, sizeof(float) * numBins);
        std::vector<double> &c = m.mfcc(spec);
        if (melbands) memcpy(melbands + f * numFilters, m.melBands, sizeof(double) * numFilters);
        memcpy(mfcc + f * numCoeffs, c.data(), sizeof(double) * numCoeffs);
    }
    return 0;
}

// ---- maxiGrains: window tables, maxiTimeStretch / maxiStretch banks (src/libs/maxiGrains.h) -------
extern "C++" {
template <typename F>
static int window_of(unsigned length, double *out) {
    maxiGrainWindowCache<F> cache;
    if (length >= cache.cacheSize) return -2;
    double *w = cache.getWindow(length);
    memcpy(out, w, sizeof(double) * length);
    return 0;
}
}
int mxo_grain_window(int kind, unsigned length, double *out) {
    switch (kind) {
        case 0: return window_of<hannWinFunctor>(length, out);
        case 1: return window_of<hammingWinFunctor>(length, out);
        case 2: return window_of<cosineWinFunctor>(length, out);
        case 3: return window_of<rectWinFunctor>(length, out);
        case 4: return window_of<triangleWinFunctor>(length, out);
        case 5: return window_of<triangleNZWinFunctor>(length, out);
        case 6: return window_of<blackmanHarrisWinFunctor>(length, out);
        case 7: return window_of<blackmanNutallWinFunctor>(length, out);
        case 8: return window_of<gaussianWinFunctor>(length, out);
    }
    return -1;
}

// State layout as in oracle/maxi_oracle.c: st = [4][S] position, looper, randomOffset, rand cursor;
// gst = [4][8][S] per live grain (creation order) pos, inc, sampleIdx, sampleDur.
extern "C++" {
template <typename F>
static int granular_bank(int mode, size_t S, size_t T, maxiSample &smp, double grainLength, int overlaps,
                         const double *a, const double *b, const double *posMod, const int32_t *rnd, size_t R,
                         double *st, double *gst, double *out) {
    int rc = 0;
    for (size_t s = 0; s < S && rc == 0; s++) {
        // mode 0 maxiTimeStretch::play, 1 maxiStretch::play, 2 maxiTimeStretch::playAtPosition
        // (a = per-sample pos [T][S]), 3 maxiPitchShift::play (st[1] = cycles)
        maxiTimeStretch<F> ts(&smp);
        maxiStretch<F> stx(&smp);
        maxiPitchShift<F> ps(&smp);
        maxiGrainPlayer *gp = mode == 3 ? ps.grainPlayer : (mode == 1 ? stx.grainPlayer : ts.grainPlayer);
        maxiGrainWindowCache<F> *wc = mode == 3 ? &ps.windowCache : (mode == 1 ? &stx.windowCache : &ts.windowCache);
        if (mode == 0 || mode == 2) {
            ts.position = st[0 * S + s];
            ts.looper = st[1 * S + s];
            ts.randomOffset = st[2 * S + s];
        } else if (mode == 1) {
            stx.position = st[0 * S + s];
            stx.looper = st[1 * S + s];
            stx.randomOffset = st[2 * S + s];
        } else {
            ps.position = st[0 * S + s];
            ps.cycles = (long)st[1 * S + s];
            ps.randomOffset = st[2 * S + s];
        }
        for (int k = 0; k < 8; k++) {  // re-create the carried-over grains, in creation order
            unsigned long dur = (unsigned long)gst[(3 * 8 + k) * S + s];
            if (!dur) continue;
            maxiGrain<F> *g = new maxiGrain<F>(&smp, 0.0, grainLength, 1, wc);
            g->pos = gst[(0 * 8 + k) * S + s];
            g->inc = gst[(1 * 8 + k) * S + s];
            g->sampleIdx = (unsigned long)gst[(2 * 8 + k) * S + s];
            g->sampleDur = dur;
            g->window = wc->getWindow(dur);
            gp->addGrain(g);
        }
        g_rnd_q = rnd ? rnd + s * R : nullptr;
        g_rnd_n = R;
        g_rnd_i = (size_t)st[3 * S + s];
        g_rnd_underrun = 0;
        for (size_t n = 0; n < T; n++) {
            const double pm = posMod ? posMod[s] : 0.0;
            switch (mode) {
                case 0: out[n * S + s] = ts.play(a[s], grainLength, overlaps, pm); break;
                case 1: out[n * S + s] = stx.play(a[s], b[s], grainLength, overlaps, pm); break;
                case 2: out[n * S + s] = ts.playAtPosition(a[n * S + s], grainLength, overlaps); break;
                default: out[n * S + s] = ps.play(a[s], grainLength, overlaps, pm); break;
            }
            if (gp->grains.size() > 8) {
                rc = -3;
                break;
            }
        }
        if (g_rnd_underrun) rc = -4;
        if (mode == 0 || mode == 2) {
            st[0 * S + s] = ts.position;
            st[1 * S + s] = ts.looper;
            st[2 * S + s] = ts.randomOffset;
        } else if (mode == 1) {
            st[0 * S + s] = stx.position;
            st[1 * S + s] = stx.looper;
            st[2 * S + s] = stx.randomOffset;
        } else {
            st[0 * S + s] = ps.position;
            st[1 * S + s] = (double)ps.cycles;
            st[2 * S + s] = ps.randomOffset;
        }
        st[3 * S + s] = (double)g_rnd_i;
        int k = 0;
        for (maxiGrainBase *gb : gp->grains) {
            if (k >= 8) break;
            maxiGrain<F> *g = static_cast<maxiGrain<F> *>(gb);
            gst[(0 * 8 + k) * S + s] = g->pos;
            gst[(1 * 8 + k) * S + s] = g->inc;
            gst[(2 * 8 + k) * S + s] = (double)g->sampleIdx;
            gst[(3 * 8 + k) * S + s] = (double)g->sampleDur;
            k++;
        }
        for (; k < 8; k++)
            for (int q = 0; q < 4; q++) gst[(q * 8 + k) * S + s] = 0.0;
        for (maxiGrainBase *gb : gp->grains) delete gb;  // the reference's players leak live grains
        gp->grains.clear();
        g_rnd_q = nullptr;
    }
    return rc;
}
}  // extern "C++"

int mxo_granular(int mode, int window_kind, size_t S, size_t T, const double *amp, size_t len,
                 int mySampleRate, double grainLength, int overlaps, const double *a, const double *b,
                 const double *posMod, const int32_t *rnd, size_t R, double *st, double *gst, double *out) {
    if (mode < 0 || mode > 3 || overlaps <= 0) return -1;
    unsigned long sampleDur = grainLength * (double)mySampleRate;
    if (sampleDur == 0 || sampleDur >= (unsigned long)(maxiSettings::sampleRate / 2.0)) return -2;
    maxiSample smp;
    std::vector<double> data(amp, amp + len);
    smp.amplitudes.reserve(len + 2);
    smp.setSample(data);
    smp.amplitudes.data()[len] = amp[len];  // the guard element (see mxo_sample)
    smp.mySampleRate = mySampleRate;
#define MXO_G(F) return granular_bank<F>(mode, S, T, smp, grainLength, overlaps, a, b, posMod, rnd, R, st, gst, out)
    switch (window_kind) {
        case 0: MXO_G(hannWinFunctor);
        case 1: MXO_G(hammingWinFunctor);
        case 2: MXO_G(cosineWinFunctor);
        case 3: MXO_G(rectWinFunctor);
        case 4: MXO_G(triangleWinFunctor);
        case 5: MXO_G(triangleNZWinFunctor);
        case 6: MXO_G(blackmanHarrisWinFunctor);
        case 7: MXO_G(blackmanNutallWinFunctor);
        case 8: MXO_G(gaussianWinFunctor);
    }
#undef MXO_G
    return -1;
}

// ---- CPU baseline timer: sinebuf bank sharded over host threads ----------------------
// Renders N samples x V voices with maxiOsc::sinebuf, voices split in contiguous
// ranges over `threads` std::threads; returns seconds (steady clock, render loop only).
double mxo_time_osc(int wf, size_t V, size_t N, const double *freq, int threads, double *sink) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    std::vector<double> acc(threads, 0.0);
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            size_t v0 = V * t / threads, v1 = V * (t + 1) / threads, nv = v1 - v0;
            std::vector<maxiOsc> bank(nv);
            std::vector<double> row(nv);
            double a = 0;
            for (size_t n = 0; n < N; n++) {
                for (size_t v = 0; v < nv; v++) {
                    double r;
                    switch (wf) {
                        case 9: r = bank[v].sinebuf4(freq[v0 + v]); break;
                        case 3: r = bank[v].saw(freq[v0 + v]); break;
                        case 0: r = bank[v].sinewave(freq[v0 + v]); break;
                        default: r = bank[v].sinebuf(freq[v0 + v]); break;
                    }
                    row[v] = r;
                }
                a += row[n % nv];
            }
            acc[t] = a;
        });
    }
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    double s = 0;
    for (double a : acc) s += a;
    if (sink) *sink = s;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- CPU baselines for the other BASELINE configs (SURVEY 8d "CPU baseline"): the reference's own per-sample
// loops, units sharded contiguously over `threads` std::threads (no shared mutable state), render loop timed only.
// config 3: per voice saw -> lores -> adsr in the voice-inner order of 15.polysynth/main.cpp:54-70, setAttack(10)
// setDecay(100) setSustain(0.5) setRelease(500), trig(n) = (n mod 44100) < 22050.  mode 1 = the 14.monosynth:50-55 order
// (cutoff = adsr * cutoff[v], per-sample coefficients).
double mxo_time_voice(int mode, size_t V, size_t N, const double *freq, const double *cutoff, const double *res,
                      int threads, double *sink) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    std::vector<double> acc(threads, 0.0);
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            const size_t v0 = V * t / threads, v1 = V * (t + 1) / threads, nv = v1 - v0;
            std::vector<maxiOsc> osc(nv);
            std::vector<maxiFilter> flt(nv);
            std::vector<maxiEnv> env(nv);
            for (size_t v = 0; v < nv; v++) {
                memset(&env[v], 0, sizeof(maxiEnv));  // static-storage semantics (maxiEnv has no ctor)
                env[v].holdtime = 1;
                env[v].setAttack(10);
                env[v].setDecay(100);
                env[v].setSustain(0.5);
                env[v].setRelease(500);
                for (int k = 0; k < 3; k++) flt[v].outputs[k] = 0.0;
            }
            double a = 0;
            for (size_t n = 0; n < N; n++) {
                const int trig = (n % 44100) < 22050 ? 1 : 0;
                double last = 0;
                for (size_t v = 0; v < nv; v++) {
                    if (mode == 0) {
                        last = env[v].adsr(flt[v].lores(osc[v].saw(freq[v0 + v]), cutoff[v0 + v], res[v0 + v]), trig);
                    } else {
                        const double e = env[v].adsr(1.0, trig);
                        last = flt[v].lores(osc[v].saw(freq[v0 + v]), e * cutoff[v0 + v], res[v0 + v]) * e;
                    }
                }
                a += last;
            }
            acc[t] = a;
        });
    }
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    double s = 0;
    for (double a : acc) s += a;
    if (sink) *sink = s;
    return std::chrono::duration<double>(t1 - t0).count();
}

// config 4: the streaming loop of cpp/commandline/tests/mfcctest/mfcctest.cpp:21-32 -- one sample at a time into
// maxiFFT::process (setup(1024,1024,1024), polar conversion), maxiMFCC::mfcc(512,42,13,20,20000) on every new frame --
// frames sharded over threads, each thread with its own maxiFFT / maxiMFCC objects.
double mxo_time_spectral(size_t nframes, const float *signal, int threads, double *sink) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    std::vector<double> acc(threads, 0.0);
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            const size_t f0 = nframes * t / threads, f1 = nframes * (t + 1) / threads;
            maxiFFT f;
            f.setup(1024, 1024, 1024);
            maxiMFCC m;
            mfcc_setup(m, 512, 42, 13, 20.0, 20000.0);
            double a = 0;
            for (size_t s = f0 * 1024; s < f1 * 1024; s++) {
                if (f.process(signal[s], maxiFFT::WITH_POLAR_CONVERSION)) {
                    std::vector<double> &c = m.mfcc(f.getMagnitudes());
                    a += c[1];
                }
            }
            acc[t] = a;
        });
    }
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    double s = 0;
    for (double a : acc) s += a;
    if (sink) *sink = s;
    return std::chrono::duration<double>(t1 - t0).count();
}

// config 5: S maxiTimeStretch<hannWinFunctor> streams over one shared maxiSample, play(speed, 0.05, 4, 0)
// (src/libs/maxiGrains.h:341-355) for T samples, stream-inner loop, streams sharded over threads.  rand() % 10 is
// routed to the (empty) per-stream queue => 0, as in the parity runs.
double mxo_time_grains(size_t S, size_t T, const double *amp, size_t len, const double *speed, const double *pos01,
                       int threads, double *sink) {
    if (threads < 1) threads = 1;
    maxiSample smp;
    std::vector<double> data(amp, amp + len);
    smp.amplitudes.reserve(len + 2);
    smp.setSample(data);
    smp.amplitudes.data()[len] = 0.0;
    std::vector<std::thread> pool;
    std::vector<double> acc(threads, 0.0);
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            const size_t s0 = S * t / threads, s1 = S * (t + 1) / threads, ns = s1 - s0;
            std::vector<std::unique_ptr<maxiTimeStretch<hannWinFunctor>>> ts;
            for (size_t s = 0; s < ns; s++) {
                ts.emplace_back(new maxiTimeStretch<hannWinFunctor>(&smp));
                ts.back()->setPosition(pos01[s0 + s]);
            }
            double a = 0;
            for (size_t n = 0; n < T; n++) {
                double last = 0;
                for (size_t s = 0; s < ns; s++) last = ts[s]->play(speed[s0 + s], 0.05, 4, 0.0);
                a += last;
            }
            acc[t] = a;
        });
    }
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    double s = 0;
    for (double a : acc) s += a;
    if (sink) *sink = s;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- maxiConvolve (src/libs/maxiConvolve.cpp:13-107) ------------------------------------------------
// The impulse goes through the reference's own file path: the 16-bit samples are written as a mono WAV and
// maxiConvolve::setup(file, fftsize, hopsize) loads and analyses it.  mode 0: play(w) verbatim (silence: COMPLEX mode of
// maxiIFFT transforms its never-written inputs, fft.cpp:613-619).  mode 1 ("as intended"): play()'s own body, with the
// last line replaced by maxiIFFT::process's body in which the sums are routed to the transform INPUTS.
long mxo_convolve(const int16_t *pcm, size_t len, int fftsize, int hopsize, const float *in, size_t n, float *out,
                  int mode, float *imp_real, float *imp_imag, size_t cap_frames) {
    char path[64];
    snprintf(path, sizeof(path), "/tmp/mxo_convolve_%d_%p.wav", (int)getpid(), (void *)pcm);
    {
        FILE *f = fopen(path, "wb");
        if (!f) return -2;
        const int32_t dsize = (int32_t)(len * 2), chunk = 36 + dsize, sub1 = 16, rate = 44100, brate = 88200;
        const int16_t fmt = 1, ch = 1, align = 2, bits = 16;
        fwrite("RIFF", 1, 4, f); fwrite(&chunk, 4, 1, f); fwrite("WAVE", 1, 4, f); fwrite("fmt ", 1, 4, f);
        fwrite(&sub1, 4, 1, f); fwrite(&fmt, 2, 1, f); fwrite(&ch, 2, 1, f); fwrite(&rate, 4, 1, f);
        fwrite(&brate, 4, 1, f); fwrite(&align, 2, 1, f); fwrite(&bits, 2, 1, f); fwrite("data", 1, 4, f);
        fwrite(&dsize, 4, 1, f); fwrite(pcm, 2, len, f);
        fclose(f);
    }
    maxiConvolve c;
    {
        // maxiConvolve::setup (maxiConvolve.cpp:13-70) driven step by step instead of called: its impulse analysis starts by
        // reading amplitudes[len] -- maxiSample::load leaves position = size (maximilian.cpp:681) and play() reads before it
        // wraps (:740-744) -- i.e. one element past its vector, which is whatever the heap holds there (usually 0, now and then
        // not: the analysis then differs from run to run).  The oracle's convention for that element is 0 (as for every
        // player), so the sample is given a zeroed spare element before the reference's own objects do the analysis.
        std::streambuf *old = std::cout.rdbuf(nullptr);  // load() prints the channel count
        maxiSample impulse;
        impulse.load(path);
        std::cout.rdbuf(old);
        remove(path);
        impulse.amplitudes.push_back(0.0);
        impulse.amplitudes.pop_back();  // capacity stays: [len] is readable and 0
        maxiFFT fft;
        fft.setup(fftsize, fftsize, hopsize);
        float maxReal = 0, maxImag = 0;
        const int nb = fft.getNumBins();
        auto keep = [&]() {
            c.impulseReal.emplace_back(fft.getReal(), fft.getReal() + nb);
            c.impulseImag.emplace_back(fft.getImag(), fft.getImag() + nb);
            for (float r : c.impulseReal.back()) if (r > maxReal) maxReal = r;
            for (float q : c.impulseImag.back()) if (q > maxImag) maxImag = q;
        };
        const long L = (long)impulse.getLength();
        for (long i = 0; i < L; i++)
            if (fft.process(impulse.play(), maxiFFT::NO_POLAR_CONVERSION)) keep();
        for (int i = 0; i < nb - (int)(L % nb); i++)
            if (fft.process(0, maxiFFT::NO_POLAR_CONVERSION)) keep();
        for (size_t k = 0; k < c.impulseReal.size(); k++)
            for (int j = 0; j < nb; j++) {
                c.impulseReal[k][j] /= maxReal;
                c.impulseImag[k][j] /= maxImag;
            }
        c.inFFT.setup(fftsize, fftsize, hopsize);
        c.ifft.setup(fftsize, fftsize, hopsize);
        for (size_t k = 0; k < c.impulseReal.size(); k++) {
            c.FDLReal.push_front(std::vector<float>(nb, 0.0f));
            c.FDLImag.push_front(std::vector<float>(nb, 0.0f));
        }
        c.sumReal.assign(nb, 0.0f);
        c.sumImag.assign(nb, 0.0f);
    }
    const size_t nfr = c.impulseReal.size();
    const int bins = c.inFFT.getNumBins();
    for (size_t k = 0; k < nfr && k < cap_frames; k++) {
        if (imp_real) memcpy(imp_real + k * bins, c.impulseReal[k].data(), sizeof(float) * bins);
        if (imp_imag) memcpy(imp_imag + k * bins, c.impulseImag[k].data(), sizeof(float) * bins);
    }
    for (size_t s = 0; s < n; s++) {
        if (mode == 0) {
            out[s] = c.play(in[s]);
            continue;
        }
        const float w = in[s];
        if (c.inFFT.process(w, maxiFFT::NO_POLAR_CONVERSION)) {  // :77-99, verbatim
            vector<float> realFrame;
            realFrame.assign(c.inFFT.getReal(), c.inFFT.getReal() + c.inFFT.getNumBins());
            c.FDLReal.push_front(realFrame);
            c.FDLReal.pop_back();
            vector<float> imagFrame;
            imagFrame.assign(c.inFFT.getImag(), c.inFFT.getImag() + c.inFFT.getNumBins());
            c.FDLImag.push_front(imagFrame);
            c.FDLImag.pop_back();
            std::fill(c.sumReal.begin(), c.sumReal.end(), 0);
            std::fill(c.sumImag.begin(), c.sumImag.end(), 0);
            auto impRealIt = c.impulseReal.begin();
            auto impImagIt = c.impulseImag.begin();
            auto fdlRealIt = c.FDLReal.begin();
            for (auto fdlImagIt = c.FDLImag.begin(); fdlImagIt != c.FDLImag.end(); ++fdlRealIt, ++fdlImagIt, ++impRealIt, ++impImagIt) {
                c.sumReal[0] += ((*impRealIt)[0] * (*fdlRealIt)[0]);
                c.sumImag[0] += ((*impImagIt)[0] * (*fdlImagIt)[0]);
                for (int i = 1; i < (int)c.sumReal.size(); i++) {
                    c.sumReal[i] += ((*impRealIt)[i] * (*fdlRealIt)[i]) - ((*impImagIt)[i] * (*fdlImagIt)[i]);
                    c.sumImag[i] += ((*impRealIt)[i] * (*fdlImagIt)[i]) + ((*impImagIt)[i] * (*fdlRealIt)[i]);
                }
            }
        }
        maxiIFFT &f = c.ifft;  // maxiIFFT::process (maxiFFT.cpp:155-192) with the inputs routed to in_real / in_img
        if (0 == f.pos) {
            std::fill(f.ifftOut.begin(), f.ifftOut.end(), 0);
            for (int i = 0; i < f._fft.half; i++) {
                f._fft.in_real[i] = c.sumReal[i];
                f._fft.in_img[i] = c.sumImag[i];
            }
            f._fft.calcIFFT(0, &f.ifftOut[0], &f.window[0]);
            memcpy(&f.buffer[0], &f.buffer[0] + f.hopSize, (f.fftSize - f.hopSize) * sizeof(float));
            memset(&f.buffer[0] + (f.fftSize - f.hopSize), 0, f.hopSize * sizeof(float));
            for (int i = 0; i < f.fftSize; i++) f.buffer[i] += f.ifftOut[i];
        }
        f.nextValue = f.buffer[f.pos];
        if (f.hopSize == ++f.pos) f.pos = 0;
        out[s] = f.nextValue;
    }
    return (long)nfr;
}

// ---- maxiMix::stereo/quad/ambisonic over a bank (src/maximilian.cpp:503-541) ---------------------
// bus[n][c][v] = what the reference leaves in two/four/eight[c]; mix[n][c] = voice-order sum.
int mxo_mix_bus(int C, size_t V, size_t N, const double *in, const double *x, const double *y,
                const double *z, double *bus, double *mix) {
    if (C != 2 && C != 4 && C != 8) return -1;
    maxiMix m;
    std::vector<double> o(C);
    for (size_t n = 0; n < N; n++) {
        std::vector<double> acc(C, 0.0);
        for (size_t v = 0; v < V; v++) {
            const double i = in[n * V + v];
            if (C == 2) m.stereo(i, o, x[v]);
            else if (C == 4) m.quad(i, o, x[v], y[v]);
            else m.ambisonic(i, o, x[v], y[v], z[v]);
            for (int c = 0; c < C; c++) {
                if (bus) bus[(n * C + c) * V + v] = o[c];
                acc[c] += o[c];
            }
        }
        for (int c = 0; c < C; c++) mix[n * C + c] = acc[c];
    }
    return 0;
}

// ---- maxiOsc::noise (src/maximilian.cpp:214-220) ---------------------------------------------------
// The reference draws from libc rand().  srand(seed), record count draws, srand(seed) again and
// let the reference consume the same draws in voice-inner order: out[n*V+v] pairs with rnd[n*V+v].
int mxo_noise(unsigned seed, size_t V, size_t N, int32_t *rnd, double *out) {
    srand(seed);
    for (size_t i = 0; i < V * N; i++) rnd[i] = rand();
    srand(seed);
    std::vector<maxiOsc> bank(V);
    for (size_t n = 0; n < N; n++)
        for (size_t v = 0; v < V; v++) out[n * V + v] = bank[v].noise();
    return 0;
}

// ---- maxiSample trigger-driven players (src/maximilian.cpp:1006-1042) and playWithPhasor (:753-816)
// mode 0 playOnZX(trig); 1 playOnZXAtSpeed(trig, a); 2 playOnZXAtSpeedFromOffset(trig, a, p0);
// 3 playOnZXAtSpeedBetweenPoints(trig, a, p0, p1); 4 loopSetPosOnZX(trig, p0).
// zx_prev/zx_first mirror maxiTrigger::previousValue/firstTrigger (H:593-594; fresh = 1, 1).
int mxo_sample_zx(int mode, size_t V, size_t N, const double *amp, size_t len, int mySampleRate,
                  const double *trig, const double *a, int aps, const double *p0, const double *p1,
                  double *position, double *zx_prev, int32_t *zx_first, double *out) {
    if (mode < 0 || mode > 4) return -1;
    std::vector<double> data(amp, amp + len);
    for (size_t v = 0; v < V; v++) {
        maxiSample s;
        s.amplitudes.reserve(len + 2);
        s.setSample(data);
        s.amplitudes.data()[len] = 0.0;
        s.amplitudes.data()[len + 1] = 0.0;
        s.mySampleRate = mySampleRate;
        s.position = position[v];
        s.zxTrig.previousValue = zx_prev[v];
        s.zxTrig.firstTrigger = zx_first[v] != 0;
        for (size_t n = 0; n < N; n++) {
            const double t = trig[n * V + v];
            const double x = a ? (aps ? a[n * V + v] : a[v]) : 1.0;
            double o = 0;
            switch (mode) {
                case 0: o = s.playOnZX(t); break;
                case 1: o = s.playOnZXAtSpeed(t, x); break;
                case 2: o = s.playOnZXAtSpeedFromOffset(t, x, p0[v]); break;
                case 3: o = s.playOnZXAtSpeedBetweenPoints(t, x, p0[v], p1[v]); break;
                case 4: o = s.loopSetPosOnZX(t, p0[v]); break;
            }
            out[n * V + v] = o;
        }
        position[v] = s.position;
        zx_prev[v] = s.zxTrig.previousValue;
        zx_first[v] = s.zxTrig.firstTrigger;
    }
    return 0;
}

int mxo_sample_phasor(size_t V, size_t N, const double *amp, size_t len, const double *pha,
                      double *phasor_prev, int32_t *phasor_first, double *out) {
    std::vector<double> data(amp, amp + len);
    for (size_t v = 0; v < V; v++) {
        maxiSample s;
        s.setSample(data);
        s.phasorPrev = phasor_prev[v];
        s.phasorFirst = phasor_first[v] != 0;
        for (size_t n = 0; n < N; n++) out[n * V + v] = s.playWithPhasor(pha[n * V + v]);
        phasor_prev[v] = s.phasorPrev;
        phasor_first[v] = s.phasorFirst;
    }
    return 0;
}

// ---- maxiSample::load / read / save (src/maximilian.cpp:605-725) ----------------------------------
// hdr = {ChunkSize, SubChunk1Size, Format, Channels, SampleRate, ByteRate, BlockAlign, BitsPerSample}
// Returns amplitudes.size() (or -1 if the file cannot be opened); copies min(size, cap) values.
// The reference prints "Loading: ..." to stdout; stdout is parked on /dev/null for the call.
long mxo_wav_load(const char *path, int channel, double *out, size_t cap, int32_t *hdr, double *position) {
    maxiSample s;
    fflush(stdout);
    std::streambuf *old = std::cout.rdbuf(nullptr);
    bool ok = s.load(path, channel);
    std::cout.rdbuf(old);
    if (!ok) return -1;
    const size_t n = s.amplitudes.size();
    for (size_t i = 0; i < n && i < cap; i++) out[i] = s.amplitudes[i];
    if (hdr) {
        hdr[0] = s.myChunkSize; hdr[1] = s.mySubChunk1Size; hdr[2] = s.myFormat; hdr[3] = s.myChannels;
        hdr[4] = s.mySampleRate; hdr[5] = s.myByteRate; hdr[6] = s.myBlockAlign; hdr[7] = s.myBitsPerSample;
    }
    if (position) *position = s.position;
    return (long)n;
}

int mxo_wav_save(const char *path, const double *amp, size_t len, const int32_t *hdr) {
    maxiSample s;
    s.amplitudes.assign(amp, amp + len);
    s.myChunkSize = hdr[0]; s.mySubChunk1Size = hdr[1]; s.myFormat = (short)hdr[2]; s.myChannels = (short)hdr[3];
    s.mySampleRate = hdr[4]; s.myByteRate = hdr[5]; s.myBlockAlign = (short)hdr[6]; s.myBitsPerSample = (short)hdr[7];
    return s.save(path) ? 0 : -1;
}

// ---- maxiIFFT (src/libs/maxiFFT.cpp:140-192; fft::polToCart/calcIFFT src/libs/fft.cpp:590-624) ----------
// nframes spectra (mags/phases [nframes][bins]) -> the nframes*hopSize samples process() returns when it
// is called hopSize times per spectrum (SPECTRUM mode).  ifft_out (optional, [nframes][fftSize]) = the
// windowed inverse transform of each frame before overlap-add; buffer (fftSize, in/out) = the member
// `buffer` (overlap-add state), so consecutive calls continue one stream.
int mxo_ifft_stream(const float *mags, const float *phases, size_t nframes, int fftSize, int hopSize,
                    int windowSize, float *out, float *ifft_out, float *buffer) {
    if (fftSize < 4 || (fftSize & (fftSize - 1)) || hopSize <= 0 || hopSize > fftSize) return -1;
    if (windowSize > fftSize) return -2;  // genWindow would overrun `window`
    maxiIFFT f;
    f.setup(fftSize, hopSize, windowSize);
    const int bins = f.getNumBins();
    if (buffer) std::copy(buffer, buffer + fftSize, f.buffer.begin());
    std::vector<float> m(bins), p(bins);
    for (size_t k = 0; k < nframes; k++) {
        std::copy(mags + k * bins, mags + (k + 1) * bins, m.begin());
        std::copy(phases + k * bins, phases + (k + 1) * bins, p.begin());
        for (int i = 0; i < hopSize; i++) {
            out[k * hopSize + i] = f.process(m, p, maxiIFFT::SPECTRUM);
            if (i == 0 && ifft_out) std::copy(f.ifftOut.begin(), f.ifftOut.end(), ifft_out + k * fftSize);
        }
    }
    if (buffer) std::copy(f.buffer.begin(), f.buffer.end(), buffer);
    return 0;
}

// ---- maxiDCBlocker H:1255-1267, maxiSVF H:1281-1338, maxiBiquad H:1343-1486 ------------------------------
// kind 0 DC blocker: par = [1][V] R.                      st = [3][V] xm1, ym1, -
// kind 1 SVF: par = [6][V] cutoff, res, lpmix, bpmix, hpmix, notchmix.  st = [3][V] v0z, v1, v2
// kind 2 biquad: par = [4][V] type, cutoff, Q, peakGain.  st = [3][V] v[0], v[1], v[2]
// coef (optional, [5][V]): SVF g1,g2,g3,g4,k ; biquad a0,a1,a2,b1,b2 -- what setParams()/set() computed.
int mxo_filter2(int kind, size_t V, size_t N, const double *in, const double *par, double *st, double *coef,
                double *out) {
    if (kind < 0 || kind > 2) return -1;
    for (size_t v = 0; v < V; v++) {
        if (kind == 0) {
            maxiDCBlocker d;
            d.xm1 = st[v];
            d.ym1 = st[V + v];
            for (size_t n = 0; n < N; n++) out[n * V + v] = d.play(in[n * V + v], par[v]);
            st[v] = d.xm1;
            st[V + v] = d.ym1;
        } else if (kind == 1) {
            maxiSVF f;
            f.setCutoff(par[v]);
            f.setResonance(par[V + v]);
            f.v0z = st[v]; f.v1 = st[V + v]; f.v2 = st[2 * V + v];
            if (coef) { coef[v] = f.g1; coef[V + v] = f.g2; coef[2 * V + v] = f.g3; coef[3 * V + v] = f.g4; coef[4 * V + v] = f.k; }
            for (size_t n = 0; n < N; n++)
                out[n * V + v] = f.play(in[n * V + v], par[2 * V + v], par[3 * V + v], par[4 * V + v], par[5 * V + v]);
            st[v] = f.v0z; st[V + v] = f.v1; st[2 * V + v] = f.v2;
        } else {
            maxiBiquad b;
            b.set((maxiBiquad::filterTypes)(int)par[v], par[V + v], par[2 * V + v], par[3 * V + v]);
            b.v[0] = st[v]; b.v[1] = st[V + v]; b.v[2] = st[2 * V + v];
            if (coef) { coef[v] = b.a0; coef[V + v] = b.a1; coef[2 * V + v] = b.a2; coef[3 * V + v] = b.b1; coef[4 * V + v] = b.b2; }
            for (size_t n = 0; n < N; n++) out[n * V + v] = b.play(in[n * V + v]);
            st[v] = b.v[0]; st[V + v] = b.v[1]; st[2 * V + v] = b.v[2];
        }
    }
    return 0;
}

// ---- maxiEnvGen (src/maximilian.h:2268-2547) -----------------------------------------------------------------
// One envelope shape (levels/times/curves, loop, retrigger) for the bank, one trigger signal per voice
// (tpv) or shared.  dst = [5][V]: envval, stages[phase].currentlevel, previousValue of trigDetector,
// holdDetector, retriggerDetector.  ist = [7][V] int64: phase, state (0 WAITING 1 TRIGGERED 2 HOLDING),
// nxcHappened, stages[phase].counter, firstTrigger of the three detectors.
// stages_out (optional, [nstages][6]): startlevel, endlevel, gradient, curve, length, hold as setup() left them.
int mxo_envgen(size_t V, size_t N, const double *trig, int tpv, size_t nlevels, const double *levels,
               const double *times, const double *curves, int loop, int retrigger, double *dst, int64_t *ist,
               double *stages_out, double *out) {
    if (nlevels < 2) return -1;
    std::vector<double> lv(levels, levels + nlevels), tm(times, times + nlevels - 1), cv(curves, curves + nlevels - 1);
    for (size_t v = 0; v < V; v++) {
        maxiEnvGen e;
        fflush(stdout);
        std::streambuf *old = std::cout.rdbuf(nullptr);  // setup() prints every stage
        bool ok = e.setup(lv, tm, cv, loop != 0, retrigger != 0);
        std::cout.rdbuf(old);
        if (!ok) return -2;
        if (v == 0 && stages_out)
            for (size_t i = 0; i < e.stages.size(); i++) {
                stages_out[i * 6 + 0] = e.stages[i].startlevel; stages_out[i * 6 + 1] = e.stages[i].endlevel;
                stages_out[i * 6 + 2] = e.stages[i].gradient;   stages_out[i * 6 + 3] = e.stages[i].curve;
                stages_out[i * 6 + 4] = (double)e.stages[i].length; stages_out[i * 6 + 5] = e.stages[i].hold;
            }
        e.envval = dst[v];
        e.phase = (size_t)ist[v];
        e.state = (decltype(e.state))ist[V + v];
        e.nxcHappened = ist[2 * V + v] != 0;
        if (e.phase < e.stages.size()) {
            e.stages[e.phase].currentlevel = dst[V + v];
            e.stages[e.phase].counter = (size_t)ist[3 * V + v];
        }
        e.trigDetector.previousValue = dst[2 * V + v];      e.trigDetector.firstTrigger = ist[4 * V + v] != 0;
        e.holdDetector.previousValue = dst[3 * V + v];      e.holdDetector.firstTrigger = ist[5 * V + v] != 0;
        e.retriggerDetector.previousValue = dst[4 * V + v]; e.retriggerDetector.firstTrigger = ist[6 * V + v] != 0;
        for (size_t n = 0; n < N; n++) out[n * V + v] = e.play(tpv ? trig[n * V + v] : trig[n]);
        dst[v] = e.envval;
        ist[v] = (int64_t)e.phase;
        ist[V + v] = (int64_t)e.state;
        ist[2 * V + v] = e.nxcHappened;
        const bool in = e.phase < e.stages.size();
        dst[V + v] = in ? e.stages[e.phase].currentlevel : 0.0;
        ist[3 * V + v] = in ? (int64_t)e.stages[e.phase].counter : 0;
        dst[2 * V + v] = e.trigDetector.previousValue;      ist[4 * V + v] = e.trigDetector.firstTrigger;
        dst[3 * V + v] = e.holdDetector.previousValue;      ist[5 * V + v] = e.holdDetector.firstTrigger;
        dst[4 * V + v] = e.retriggerDetector.previousValue; ist[6 * V + v] = e.retriggerDetector.firstTrigger;
    }
    return 0;
}

// ---- maxiSampler (src/libs/maxiSynths.h:137-187, maxiSynths.cpp:262-300, 484-491) ---------------------------
// NS samplers of `voices` (<= 32) voices each, V = NS*voices lanes, voice v = sampler v/voices, slot v%voices.
// Every voice plays the same sample data.  Per-voice in/out state: position, trigger (envelopes[i].trigger),
// outhold (outputs[i]), envelope dst [2][V] / ist [6][V] as mxo_env.  In: pitch [V], gain (envOutGain) [V],
// env par [4][V] + holdtime [V].  Out: mix [N][NS] = play(); outputs (optional) [N][V] = outputs[i] after
// each play().
int mxo_sampler(size_t NS, int voices, size_t N, const double *amp, size_t len, int sustain, const double *pitch,
                const double *gain, const double *par, const int64_t *holdtime, double *position,
                int32_t *trigger, double *outhold, double *dst, int64_t *ist, double *mix, double *outputs) {
    if (voices < 1 || voices > 32) return -1;
    const size_t V = NS * (size_t)voices;
    std::vector<double> data(amp, amp + len);
    std::unique_ptr<maxiSampler> sp(new maxiSampler());
    for (size_t s = 0; s < NS; s++) {
        maxiSampler &m = *sp;
        m.setNumVoices(voices);
        m.sustain = sustain != 0;
        for (int i = 0; i < voices; i++) {
            const size_t v = s * voices + i;
            maxiSample &smp = m.samples[i];
            smp.amplitudes.reserve(len + 2);
            smp.setSample(data);
            smp.amplitudes.data()[len] = 0.0;
            smp.amplitudes.data()[len + 1] = 0.0;
            smp.position = position[v];
            m.pitch[i] = pitch[v];
            m.envOutGain[i] = gain[v];
            m.outputs[i] = outhold[v];
            maxiEnv &e = m.envelopes[i];
            e.attack = par[v]; e.decay = par[V + v]; e.sustain = par[2 * V + v]; e.release = par[3 * V + v];
            e.holdtime = (long)holdtime[v];
            e.trigger = trigger[v];
            e.amplitude = dst[v]; e.output = dst[V + v];
            e.holdcount = (long)ist[v]; e.attackphase = (int)ist[V + v]; e.decayphase = (int)ist[2 * V + v];
            e.sustainphase = (int)ist[3 * V + v]; e.holdphase = (int)ist[4 * V + v]; e.releasephase = (int)ist[5 * V + v];
        }
        for (size_t n = 0; n < N; n++) {
            mix[n * NS + s] = m.play();
            if (outputs)
                for (int i = 0; i < voices; i++) outputs[n * V + s * voices + i] = m.outputs[i];
        }
        for (int i = 0; i < voices; i++) {
            const size_t v = s * voices + i;
            position[v] = m.samples[i].position;
            trigger[v] = m.envelopes[i].trigger;
            outhold[v] = m.outputs[i];
            maxiEnv &e = m.envelopes[i];
            dst[v] = e.amplitude; dst[V + v] = e.output;
            ist[v] = e.holdcount; ist[V + v] = e.attackphase; ist[2 * V + v] = e.decayphase;
            ist[3 * V + v] = e.sustainphase; ist[4 * V + v] = e.holdphase; ist[5 * V + v] = e.releasephase;
        }
    }
    return 0;
}

}  // extern "C"
