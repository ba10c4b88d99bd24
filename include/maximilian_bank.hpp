// include/maximilian_bank.hpp -- C++ host facade over the C-ABI (include/maxigpu.h).
//
// Host code stays ordinary C++ (g++, no HIP headers): these classes keep the reference's class
// and method names (src/maximilian.h, src/libs/*.h) but stand for a BANK of V instances whose
// state lives in HBM.  Two ways to use a bank:
//
//   block API      bank.sinebuf(N)            -> device block [N][V] (one launch), chain it into
//                                                the next bank (filter, mix ...) without leaving HBM
//   per-sample API bank.frame(v) / bank.tick() -> what the reference's per-sample call returns for
//                                                voice v in the current audio frame; blocks of
//                                                `blockSize` frames are rendered behind the scenes
//                                                when the frame counter wraps, so an existing
//                                                `void play(double *output)` keeps its shape
//                                                (see host/polysynth_host.cpp).
//
// Parameters handed to a bank (frequencies, cutoffs, pans ...) are block-rate: they take effect
// at the next block boundary, the same way the reference's players read control values once
// per callback buffer.  Every method that can fail throws std::runtime_error carrying
// mxg_last_error(); nothing here falls back to the CPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "maxigpu.h"

namespace maxigpu {

inline void check(int status, const char *what) {
    if (status < 0) throw std::runtime_error(std::string(what) + ": " + mxg_last_error());
}

// RAII device array of T (hipMalloc behind mxg_malloc)
template <typename T>
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t n, bool zero = true) { resize(n, zero); }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    DeviceArray(DeviceArray &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceArray &operator=(DeviceArray &&o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceArray() { release(); }
    void resize(size_t n, bool zero = true) {
        release();
        check(mxg_init(-1), "mxg_init");
        p_ = static_cast<T *>(mxg_malloc(n * sizeof(T)));
        if (!p_) throw std::runtime_error(std::string("mxg_malloc: ") + mxg_last_error());
        n_ = n;
        if (zero && n) {  // complete before any launch on any stream can touch the array
            check(mxg_memset(p_, 0, n * sizeof(T), nullptr), "mxg_memset");
            check(mxg_stream_sync(nullptr), "mxg_stream_sync");
        }
    }
    void upload(const T *h, size_t n, size_t offset = 0) {
        check(mxg_memcpy_h2d(p_ + offset, h, n * sizeof(T), nullptr), "mxg_memcpy_h2d");
    }
    void upload(const std::vector<T> &h) {
        if (h.size() > n_) resize(h.size(), false);  // an array that was never sized grows to fit
        upload(h.data(), h.size());
    }
    void download(T *h, size_t n, size_t offset = 0) const {
        check(mxg_memcpy_d2h(h, p_ + offset, n * sizeof(T), nullptr), "mxg_memcpy_d2h");
    }
    std::vector<T> download() const {
        std::vector<T> h(n_);
        if (n_) download(h.data(), n_);
        return h;
    }
    T *get() const { return p_; }
    size_t size() const { return n_; }

private:
    void release() {
        if (p_) mxg_free(p_);
        p_ = nullptr;
        n_ = 0;
    }
    T *p_ = nullptr;
    size_t n_ = 0;
};

}  // namespace maxigpu

// maxiSettings (H:117-163)
class maxiSettings {
public:
    static size_t sampleRate, channels, bufferSize;
    static void setup(size_t initSampleRate, size_t initChannels, size_t initBufferSize) {
        maxigpu::check(mxg_settings(initSampleRate, initChannels, initBufferSize), "mxg_settings");
        sampleRate = initSampleRate;
        channels = initChannels;
        bufferSize = initBufferSize;
    }
    static size_t getSampleRate() { return sampleRate; }
};
inline size_t maxiSettings::sampleRate = 44100;
inline size_t maxiSettings::channels = 2;
inline size_t maxiSettings::bufferSize = 1024;

namespace maxigpu {

// Common machinery of a bank that produces one [N][V] block per launch and can serve it back
// frame by frame on the host.
class BlockServer {
public:
    BlockServer(size_t voices, size_t blockSize) : V(voices), B(blockSize), out_(voices * blockSize, false) {}
    size_t voices() const { return V; }
    size_t blockSize() const { return B; }
    double *deviceBlock() const { return out_.get(); }

protected:
    // per-sample facade: value of voice v in the current frame; fetches the block on first use
    template <typename RenderFn>
    double frame(size_t v, RenderFn render) {
        if (cursor_ == B || host_.empty()) {
            render(B, out_.get());
            host_.resize(V * B);
            out_.download(host_.data(), V * B);
            cursor_ = 0;
        }
        return host_[cursor_ * V + v];
    }
    void advance() { if (!host_.empty()) ++cursor_; }
    size_t V, B;
    DeviceArray<double> out_;
    std::vector<double> host_;
    size_t cursor_ = 0;
};

}  // namespace maxigpu

// ---- maxiOsc (H:169-215) -------------------------------------------------------------------------
class maxiOscBank : public maxigpu::BlockServer {
public:
    explicit maxiOscBank(size_t voices, size_t blockSize = 512)
        : BlockServer(voices, blockSize), freq_(voices), p1_(voices), p2_(voices), phase_(voices), hold_(voices) {}
    void setFrequencies(const std::vector<double> &f) { freq_.upload(f); }
    void setDuty(const std::vector<double> &d) { p1_.upload(d); }
    void setPhasorRange(const std::vector<double> &start, const std::vector<double> &end) { p1_.upload(start); p2_.upload(end); }
    void phaseReset(const std::vector<double> &phaseIn) { phase_.upload(phaseIn); }  // C:222-226
    std::vector<double> phases() const { return phase_.download(); }

    // block API: N samples of every voice into d_out ([N][V], device), state carried
    void render(int waveform, size_t N, double *d_out, const double *d_freq_per_sample = nullptr, void *stream = nullptr) {
        maxigpu::check(mxg_osc_render(waveform, V, N, d_freq_per_sample ? d_freq_per_sample : freq_.get(),
                                      d_freq_per_sample ? 1 : 0, p1_.get(), p2_.get(), phase_.get(), hold_.get(),
                                      d_out, stream), "mxg_osc_render");
    }
    void sinewave(size_t N, double *d_out) { render(MXG_OSC_SINEWAVE, N, d_out); }
    void coswave(size_t N, double *d_out) { render(MXG_OSC_COSWAVE, N, d_out); }
    void phasor(size_t N, double *d_out) { render(MXG_OSC_PHASOR, N, d_out); }
    void saw(size_t N, double *d_out) { render(MXG_OSC_SAW, N, d_out); }
    void triangle(size_t N, double *d_out) { render(MXG_OSC_TRIANGLE, N, d_out); }
    void square(size_t N, double *d_out) { render(MXG_OSC_SQUARE, N, d_out); }
    void pulse(size_t N, double *d_out) { render(MXG_OSC_PULSE, N, d_out); }
    void impulse(size_t N, double *d_out) { render(MXG_OSC_IMPULSE, N, d_out); }
    void sinebuf(size_t N, double *d_out) { render(MXG_OSC_SINEBUF, N, d_out); }
    void sinebuf4(size_t N, double *d_out) { render(MXG_OSC_SINEBUF4, N, d_out); }
    void sawn(size_t N, double *d_out) { render(MXG_OSC_SAWN, N, d_out); }
    void phasorBetween(size_t N, double *d_out) { render(MXG_OSC_PHASORBETWEEN, N, d_out); }
    // noise() (C:214-220): d_rand = the rand() draws, int32 [N][V] (draw n*V+v for a voice-inner loop)
    void noise(size_t N, const int32_t *d_rand, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_osc_noise(V, N, d_rand, hold_.get(), d_out, stream), "mxg_osc_noise");
    }

    // per-sample API: `waveform` is fixed for the facade stream; call tick() once per audio frame
    void setWaveform(int waveform) { waveform_ = waveform; }
    double frame(size_t v) {
        return BlockServer::frame(v, [this](size_t N, double *d) { render(waveform_, N, d); });
    }
    void tick() { advance(); }

private:
    maxigpu::DeviceArray<double> freq_, p1_, p2_, phase_, hold_;
    int waveform_ = MXG_OSC_SINEBUF;
};

// ---- maxiFilter (H:289-366) ------------------------------------------------------------------------
class maxiFilterBank {
public:
    explicit maxiFilterBank(size_t voices) : V(voices), state_(5 * voices), cutoff_(voices), res_(voices), coef_(3 * voices) {}
    // block-constant parameters: coefficients on the host libm (bit-exact recurrence)
    void setParams(int kind, const std::vector<double> &cutoff, const std::vector<double> &resonance) {
        cutoff_.upload(cutoff);
        if (kind <= MXG_FLT_BANDPASS) {
            res_.upload(resonance);
            std::vector<double> coef(3 * V);
            maxigpu::check(mxg_filter_coeffs_host(kind, V, cutoff.data(), resonance.data(), coef.data()), "mxg_filter_coeffs_host");
            coef_.upload(coef);
        }
        kind_ = kind;
    }
    void render(size_t N, const double *d_in, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_filter_render(kind_, V, N, d_in, cutoff_.get(), 0, res_.get(), 0, coef_.get(), state_.get(),
                                         d_out, stream), "mxg_filter_render");
    }
    // audio-rate modulated cutoff ([N][V] on device), device coefficients, stated tolerance
    void renderModulated(int kind, size_t N, const double *d_in, const double *d_cutoff, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_filter_render(kind, V, N, d_in, d_cutoff, 1, res_.get(), 0, nullptr, state_.get(), d_out, stream),
                       "mxg_filter_render");
    }
    std::vector<double> state() const { return state_.download(); }

private:
    size_t V;
    int kind_ = MXG_FLT_LORES;
    maxigpu::DeviceArray<double> state_, cutoff_, res_, coef_;
};

// ---- maxiDCBlocker / maxiSVF / maxiBiquad (H:1255-1486) --------------------------------------------------
class maxiDCBlockerBank {
public:
    explicit maxiDCBlockerBank(size_t voices) : V(voices), state_(3 * voices), R_(voices) {}
    void setR(const std::vector<double> &R) { R_.upload(R); }
    void play(size_t N, const double *d_in, double *d_out, void *stream = nullptr) {  // play(input, R) H:1261-1266
        maxigpu::check(mxg_filter2_render(0, V, N, d_in, R_.get(), state_.get(), d_out, stream), "mxg_filter2_render");
    }

private:
    size_t V;
    maxigpu::DeviceArray<double> state_, R_;
};

class maxiSVFBank {
public:
    explicit maxiSVFBank(size_t voices) : V(voices), freq_(voices, 1000.0), res_(voices, 1.0), mix_(4 * voices, 0.0), state_(3 * voices), coef_(9 * voices) {}
    void setCutoff(const std::vector<double> &cutoff) { freq_ = cutoff; dirty_ = true; }   // H:1287-1290
    void setResonance(const std::vector<double> &q) { res_ = q; dirty_ = true; }          // H:1293-1296
    void setMix(double lpmix, double bpmix, double hpmix, double notchmix) {              // the play() arguments
        for (size_t v = 0; v < V; v++) { mix_[v] = lpmix; mix_[V + v] = bpmix; mix_[2 * V + v] = hpmix; mix_[3 * V + v] = notchmix; }
        dirty_ = true;
    }
    void play(size_t N, const double *d_in, double *d_out, void *stream = nullptr) {      // H:1303-1317
        if (dirty_) {
            std::vector<double> c(9 * V);
            maxigpu::check(mxg_svf_coeffs_host(V, freq_.data(), res_.data(), c.data()), "mxg_svf_coeffs_host");
            std::copy(mix_.begin(), mix_.end(), c.begin() + 5 * V);
            coef_.upload(c);
            dirty_ = false;
        }
        maxigpu::check(mxg_filter2_render(1, V, N, d_in, coef_.get(), state_.get(), d_out, stream), "mxg_filter2_render");
    }

private:
    size_t V;
    std::vector<double> freq_, res_, mix_;
    maxigpu::DeviceArray<double> state_, coef_;
    bool dirty_ = true;
};

class maxiBiquadBank {
public:
    enum filterTypes { LOWPASS, HIGHPASS, BANDPASS, NOTCH, PEAK, LOWSHELF, HIGHSHELF };
    explicit maxiBiquadBank(size_t voices) : V(voices), state_(3 * voices), coef_(5 * voices) {}
    void set(filterTypes filtType, const std::vector<double> &cutoff, const std::vector<double> &Q, const std::vector<double> &peakGain) {  // H:1376-1478
        std::vector<int32_t> t(V, (int32_t)filtType);
        std::vector<double> c(5 * V);
        maxigpu::check(mxg_biquad_coeffs_host(V, t.data(), cutoff.data(), Q.data(), peakGain.data(), c.data()), "mxg_biquad_coeffs_host");
        coef_.upload(c);
    }
    void play(size_t N, const double *d_in, double *d_out, void *stream = nullptr) {  // H:1360-1367
        maxigpu::check(mxg_filter2_render(2, V, N, d_in, coef_.get(), state_.get(), d_out, stream), "mxg_filter2_render");
    }

private:
    size_t V;
    maxigpu::DeviceArray<double> state_, coef_;
};

// ---- maxiEnv (H:888-932) ------------------------------------------------------------------------------
class maxiEnvBank {
public:
    explicit maxiEnvBank(size_t voices)
        : V(voices), par_h_(4 * voices, 0.0), hold_h_(voices, 1), par_(4 * voices), hold_(voices), dst_(2 * voices), ist_(6 * voices) {}
    void setAttack(double ms) { fill(0, mxg_env_coeff_host(0, ms)); }      // C:1480-1482
    void setAttackMS(double ms) { fill(0, mxg_env_coeff_host(3, ms)); }    // C:1486-1488
    void setDecay(double ms) { fill(1, mxg_env_coeff_host(1, ms)); }       // C:1475-1477
    void setSustain(double level) { fill(2, level); }                     // C:1492-1494
    void setRelease(double ms) { fill(3, mxg_env_coeff_host(2, ms)); }     // C:1470-1472
    void setHoldtime(long holdtime) { for (auto &h : hold_h_) h = holdtime; dirty_ = true; }
    // trigger: int32 [N] on device (shared gate) or [N][V] (per voice)
    void adsr(size_t N, const double *d_in, const int32_t *d_trig, bool per_voice, double *d_out, void *stream = nullptr) {
        sync();
        maxigpu::check(mxg_env_render(0, V, N, d_in, d_trig, per_voice ? 1 : 0, par_.get(), hold_.get(), dst_.get(), ist_.get(),
                                      d_out, stream), "mxg_env_render");
    }
    void ar(size_t N, const double *d_in, const int32_t *d_trig, bool per_voice, double *d_out, void *stream = nullptr) {
        sync();
        maxigpu::check(mxg_env_render(1, V, N, d_in, d_trig, per_voice ? 1 : 0, par_.get(), hold_.get(), dst_.get(), ist_.get(),
                                      d_out, stream), "mxg_env_render");
    }
    const double *params() { sync(); return par_.get(); }
    const int64_t *holdtimes() { sync(); return hold_.get(); }
    double *dstate() { return dst_.get(); }
    int64_t *istate() { return ist_.get(); }

private:
    void fill(int row, double v) { for (size_t i = 0; i < V; i++) par_h_[row * V + i] = v; dirty_ = true; }
    void sync() { if (dirty_) { par_.upload(par_h_); hold_.upload(hold_h_); dirty_ = false; } }
    size_t V;
    std::vector<double> par_h_;
    std::vector<int64_t> hold_h_;
    maxigpu::DeviceArray<double> par_;
    maxigpu::DeviceArray<int64_t> hold_;
    maxigpu::DeviceArray<double> dst_;
    maxigpu::DeviceArray<int64_t> ist_;
    bool dirty_ = true;
};

// ---- maxiEnvGen (H:2268-2547): one envelope shape per bank, per-voice or shared trigger signals --------------
class maxiEnvGenBank {
public:
    static constexpr double HOLD = -46692;  // maxiEnvGen::HOLD
    explicit maxiEnvGenBank(size_t voices) : V(voices), dst_(5 * voices), ist_(7 * voices) { arm(); }
    bool setup(const std::vector<double> &levels, const std::vector<double> &times, const std::vector<double> &curves,
               bool looping, bool allowRetrigger = false) {  // H:2366-2399
        if (!(levels.size() == times.size() + 1 && levels.size() == curves.size() + 1)) return false;
        std::vector<double> st(6 * times.size());
        const int n = mxg_envgen_stages_host(levels.size(), levels.data(), times.data(), curves.data(), st.data());
        if (n < 0) return false;
        stages_.resize(st.size(), false);
        stages_.upload(st);
        nstages_ = n; loop_ = looping; retrigger_ = allowRetrigger;
        arm();
        return true;
    }
    void setupAR(double attack, double release) { setup({0, 1, 0}, {attack, release}, {1, 1}, false, false); }
    void setupASR(double attack, double release) { setup({0, 1, 1, 0}, {attack, HOLD, release}, {1, 1, 1}, false, false); }
    void setupADSR(double attack, double decay, double sustain, double release) {
        setup({0, 1, sustain, sustain, 0}, {attack, decay, HOLD, release}, {1, 1, 1, 1}, false, false);
    }
    void setRetrigger(bool v) { retrigger_ = v; }
    void setLoop(bool v) { loop_ = v; }
    // d_trig: [N][V] (per_voice) or [N] (shared gate), doubles on the device
    void play(size_t N, const double *d_trig, bool per_voice, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_envgen_render(V, N, d_trig, per_voice ? 1 : 0, stages_.get(), nstages_, loop_ ? 1 : 0, retrigger_ ? 1 : 0,
                                         dst_.get(), ist_.get(), d_out, stream), "mxg_envgen_render");
    }

private:
    void arm() {  // resetAndArm() of fresh objects: WAITING, detectors previousValue = 1 / firstTrigger = 1
        std::vector<double> d(5 * V, 0.0);
        std::vector<int64_t> i(7 * V, 0);
        for (size_t v = 0; v < V; v++) { d[2 * V + v] = d[3 * V + v] = d[4 * V + v] = 1.0; i[4 * V + v] = i[5 * V + v] = i[6 * V + v] = 1; }
        dst_.upload(d); ist_.upload(i);
    }
    size_t V;
    int nstages_ = 0;
    bool loop_ = false, retrigger_ = false;
    maxigpu::DeviceArray<double> stages_, dst_;
    maxigpu::DeviceArray<int64_t> ist_;
};

// ---- fused subtractive voice: maxiOsc::saw -> maxiFilter::lores -> maxiEnv::adsr -----------------------
class maxiVoiceBank : public maxigpu::BlockServer {
public:
    explicit maxiVoiceBank(size_t voices, size_t blockSize = 512)
        : BlockServer(voices, blockSize), env(voices), freq_(voices), cutoff_(voices), res_(voices), coef_(3 * voices),
          ost_(2 * voices), fst_(5 * voices), trig_(blockSize) {}
    maxiEnvBank env;
    void setVoices(const std::vector<double> &freq, const std::vector<double> &cutoff, const std::vector<double> &resonance) {
        freq_.upload(freq);
        cutoff_.upload(cutoff);
        res_.upload(resonance);
        std::vector<double> coef(3 * V);
        maxigpu::check(mxg_filter_coeffs_host(MXG_FLT_LORES, V, cutoff.data(), resonance.data(), coef.data()), "mxg_filter_coeffs_host");
        coef_.upload(coef);
    }
    // mode 0: coefficients hoisted (bit-exact); mode 1: cutoff = envelope*cutoff per sample (14.monosynth)
    void render(int mode, size_t N, const int32_t *d_trig, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_voice_render(mode, V, N, freq_.get(), cutoff_.get(), res_.get(), coef_.get(), d_trig, 0, env.params(),
                                        env.holdtimes(), ost_.get(), fst_.get(), env.dstate(), env.istate(), d_out, stream),
                       "mxg_voice_render");
    }
    // The same block with the maxiMix::stereo mixdown of the bank fused into the render (round 6: mxg_voice_render_mix) -- what the user
    // code of 15.polysynth/main.cpp:54-70 computes voice after voice (`mymix.stereo(out, outputs, pan)` + the sums), here as d_mix [N][2]
    // from ONE kernel that never re-reads the block.  d_out may be nullptr (mix only).  setPan() first.
    void setPan(const std::vector<double> &x) { pan_.upload(x); }
    void renderMix(int mode, size_t N, const int32_t *d_trig, double *d_out, double *d_mix, void *stream = nullptr) {
        maxigpu::check(mxg_voice_render_mix(mode, V, N, freq_.get(), cutoff_.get(), res_.get(), coef_.get(), d_trig, 0, env.params(),
                                            env.holdtimes(), ost_.get(), fst_.get(), env.dstate(), env.istate(), d_out, pan_.get(), d_mix,
                                            stream), "mxg_voice_render_mix");
    }
    // per-sample API: the gate for the NEXT block is whatever setGate() holds when the block is rendered
    void setGate(const std::vector<int32_t> &gateForBlock) { trig_.upload(gateForBlock); }
    double frame(size_t v) {
        return BlockServer::frame(v, [this](size_t N, double *d) { render(0, N, trig_.get(), d); });
    }
    // per-sample API of the mixed bank: channel ch (0 / 1) of the current frame's stereo mix, formed on the device
    double mixFrame(int ch) {
        if (mcursor_ == B || mix_host_.empty()) {
            if (mix_.size() < 2 * B) mix_.resize(2 * B, false);
            renderMix(0, B, trig_.get(), nullptr, mix_.get());
            mix_host_.resize(2 * B);
            mix_.download(mix_host_.data(), 2 * B);
            mcursor_ = 0;
        }
        return mix_host_[mcursor_ * 2 + ch];
    }
    void tick() {
        advance();
        if (!mix_host_.empty()) ++mcursor_;
    }

private:
    maxigpu::DeviceArray<double> freq_, cutoff_, res_, coef_, ost_, fst_;
    maxigpu::DeviceArray<int32_t> trig_;
    maxigpu::DeviceArray<double> pan_, mix_;
    std::vector<double> mix_host_;
    size_t mcursor_ = 0;
};

// ---- maxiMix::stereo over a bank + mixdown over voices (C:503-509) ---------------------------------------
class maxiMixBank {
public:
    explicit maxiMixBank(size_t voices) : V(voices), pan_(voices), y_(voices), z_(voices) {}
    void setPan(const std::vector<double> &x) { pan_.upload(x); }
    void setPan(const std::vector<double> &x, const std::vector<double> &y) { pan_.upload(x); y_.upload(y); }
    void setPan(const std::vector<double> &x, const std::vector<double> &y, const std::vector<double> &z) {
        pan_.upload(x); y_.upload(y); z_.upload(z);
    }
    // d_mix: [N][2] on device
    void stereo(size_t N, const double *d_in, double *d_mix, void *stream = nullptr) {
        maxigpu::check(mxg_mix_stereo(V, N, d_in, pan_.get(), d_mix, stream), "mxg_mix_stereo");
    }
    // quad (C:512-522): d_mix [N][4]; ambisonic (C:525-541): d_mix [N][8].  d_bus (optional, [N][C][V]) receives
    // the per-voice four/eight signals.
    void quad(size_t N, const double *d_in, double *d_mix, double *d_bus = nullptr, void *stream = nullptr) {
        maxigpu::check(mxg_mix_bus(4, V, N, d_in, pan_.get(), y_.get(), nullptr, d_bus, d_mix, stream), "mxg_mix_bus");
    }
    void ambisonic(size_t N, const double *d_in, double *d_mix, double *d_bus = nullptr, void *stream = nullptr) {
        maxigpu::check(mxg_mix_bus(8, V, N, d_in, pan_.get(), y_.get(), z_.get(), d_bus, d_mix, stream), "mxg_mix_bus");
    }

private:
    size_t V;
    maxigpu::DeviceArray<double> pan_, y_, z_;
};

// ---- maxiDelayline (H:266-284) ---------------------------------------------------------------------------
class maxiDelaylineBank {
public:
    maxiDelaylineBank(size_t voices, size_t capacity)
        : V(voices), cap(capacity), mem_(voices * capacity), phase_(voices), size_(voices), fb_(voices), pos_(voices) {}
    void setParams(const std::vector<int32_t> &size, const std::vector<double> &feedback) { size_.upload(size); fb_.upload(feedback); }
    void setPositions(const std::vector<int32_t> &position) { pos_.upload(position); }
    void dl(size_t N, const double *d_in, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_delay_render(0, V, N, d_in, size_.get(), fb_.get(), nullptr, mem_.get(), cap, phase_.get(), d_out, stream),
                       "mxg_delay_render");
    }
    void dlFromPosition(size_t N, const double *d_in, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_delay_render(1, V, N, d_in, size_.get(), fb_.get(), pos_.get(), mem_.get(), cap, phase_.get(), d_out, stream),
                       "mxg_delay_render");
    }

private:
    size_t V, cap;
    maxigpu::DeviceArray<double> mem_;
    maxigpu::DeviceArray<int32_t> phase_, size_;
    maxigpu::DeviceArray<double> fb_;
    maxigpu::DeviceArray<int32_t> pos_;
};

// ---- maxiSample play family (H:602-783): V play heads over one sample ---------------------------------------
class maxiSampleBank {
public:
    explicit maxiSampleBank(size_t voices)
        : V(voices), position_(voices), a_(voices), start_(voices), end_(voices), zxPrev_(voices), phasorPrev_(voices),
          zxFirst_(voices), phasorFirst_(voices) {
        zxPrev_.upload(std::vector<double>(V, 1.0));     // maxiTrigger::previousValue = 1, firstTrigger = 1 (H:593-594)
        zxFirst_.upload(std::vector<int32_t>(V, 1));
        phasorFirst_.upload(std::vector<int32_t>(V, 1));  // phasorFirst = 1, phasorPrev = 0 (H:731-732)
    }
    ~maxiSampleBank() { clear(); }
    maxiSampleBank(const maxiSampleBank &) = delete;
    maxiSampleBank &operator=(const maxiSampleBank &) = delete;
    void setSample(const std::vector<double> &sampleData) {  // H:670-678
        clear();
        d_samples_ = mxg_sample_upload(sampleData.data(), sampleData.size());
        if (!d_samples_) throw std::runtime_error(std::string("mxg_sample_upload: ") + mxg_last_error());
        length_ = sampleData.size();
        mySampleRate = 44100;
        position_.upload(std::vector<double>(V, (double)length_ - 1));
    }
    void setSampleAndRate(const std::vector<double> &sampleData, int sampleRate) { setSample(sampleData); mySampleRate = sampleRate; }
    // load(fileName, channel) C:605-692: 16-bit PCM WAV, de-interleaved and normalised on the device; position = size (C:681)
    bool load(const std::string &fileName, int channel = 0) {
        size_t n = 0;
        double *p = mxg_sample_load_wav(fileName.c_str(), channel, &n, wavHeader_);
        if (!p) return false;  // the reference's `result`
        clear();
        d_samples_ = p;
        length_ = n;
        mySampleRate = wavHeader_[4];
        position_.upload(std::vector<double>(V, (double)n));
        return true;
    }
    bool save(const std::string &fileName) {  // C:698-725
        return mxg_sample_save_wav(fileName.c_str(), d_samples_, length_, wavHeader_, nullptr) == MXG_OK;
    }
    void trigger() { position_.upload(std::vector<double>(V, 0.0)); }  // C:597-600
    void setPositions(const std::vector<double> &p) { position_.upload(p); }
    void setSpeeds(const std::vector<double> &a) { a_.upload(a); }
    void setStartEnd(const std::vector<double> &s, const std::vector<double> &e) { start_.upload(s); end_.upload(e); }
    size_t getLength() const { return length_; }
    bool isReady() const { return length_ > 1; }
    const double *deviceSamples() const { return d_samples_; }
    std::vector<double> download() const {  // the buffer as the reference's public `amplitudes` would hold it
        std::vector<double> h(length_);
        if (length_) maxigpu::check(mxg_memcpy_d2h(h.data(), d_samples_, sizeof(double) * length_, nullptr), "mxg_memcpy_d2h");
        return h;
    }
    void clear() { if (d_samples_) mxg_sample_free(d_samples_); d_samples_ = nullptr; length_ = 0; }
    void render(int mode, size_t N, double *d_out, void *stream = nullptr) {
        maxigpu::check(mxg_sample_render(mode, V, N, d_samples_, length_, mySampleRate, a_.get(), 0, start_.get(), end_.get(),
                                         position_.get(), d_out, stream), "mxg_sample_render");
    }
    void play(size_t N, double *d_out) { render(MXG_SMP_PLAY, N, d_out); }
    void playOnce(size_t N, double *d_out) { render(MXG_SMP_PLAYONCE, N, d_out); }
    void playAtSpeed(size_t N, double *d_out) { render(MXG_SMP_PLAYATSPEED, N, d_out); }
    // trigger-driven players (C:1006-1042): d_trig [N][V]; speed = setSpeeds(), offset/length (or pos) = setStartEnd()
    void renderTrig(int mode, size_t N, const double *d_trig, double *d_out, void *stream = nullptr) {
        const bool ph = mode == MXG_SMP_PLAYWITHPHASOR;
        maxigpu::check(mxg_sample_render_trig(mode, V, N, d_samples_, length_, mySampleRate, d_trig, a_.get(), 0, start_.get(),
                                              end_.get(), position_.get(), ph ? phasorPrev_.get() : zxPrev_.get(),
                                              ph ? phasorFirst_.get() : zxFirst_.get(), d_out, stream), "mxg_sample_render_trig");
    }
    void playOnZX(size_t N, const double *d_trig, double *d_out) { renderTrig(MXG_SMP_PLAYONZX, N, d_trig, d_out); }
    void playOnZXAtSpeed(size_t N, const double *d_trig, double *d_out) { renderTrig(MXG_SMP_PLAYONZXATSPEED, N, d_trig, d_out); }
    void playOnZXAtSpeedFromOffset(size_t N, const double *d_trig, double *d_out) { renderTrig(MXG_SMP_PLAYONZXATSPEEDFROMOFFSET, N, d_trig, d_out); }
    void playOnZXAtSpeedBetweenPoints(size_t N, const double *d_trig, double *d_out) { renderTrig(MXG_SMP_PLAYONZXATSPEEDBETWEENPOINTS, N, d_trig, d_out); }
    void loopSetPosOnZX(size_t N, const double *d_trig, double *d_out) { renderTrig(MXG_SMP_LOOPSETPOSONZX, N, d_trig, d_out); }
    void playWithPhasor(size_t N, const double *d_pha, double *d_out) { renderTrig(MXG_SMP_PLAYWITHPHASOR, N, d_pha, d_out); }  // C:753-816
    int mySampleRate = 44100;

private:
    size_t V;
    double *d_samples_ = nullptr;
    size_t length_ = 0;
    int32_t wavHeader_[8] = {36, 16, 1, 1, 44100, 88200, 2, 16};  // ChunkSize .. BitsPerSample of the last load()
    maxigpu::DeviceArray<double> position_, a_, start_, end_, zxPrev_, phasorPrev_;
    maxigpu::DeviceArray<int32_t> zxFirst_, phasorFirst_;
};

// ---- maxiFFT / maxiMFCC batches (L/maxiFFT.h, L/maxiMFCC.h) ----------------------------------------------------
class maxiFFTBatch {
public:
    enum fftModes { NO_POLAR_CONVERSION = 0, WITH_POLAR_CONVERSION = 1 };
    ~maxiFFTBatch() { if (plan_) mxg_fft_plan_destroy(plan_); }
    void setup(int fftSize = 1024, int hopSize = 512, int windowSize = 0) {  // L/maxiFFT.cpp:45-60
        if (plan_) mxg_fft_plan_destroy(plan_);
        maxigpu::check(mxg_init(-1), "mxg_init");
        plan_ = mxg_fft_plan_create(fftSize, hopSize, windowSize);
        if (!plan_) throw std::runtime_error(std::string("mxg_fft_plan_create: ") + mxg_last_error());
        fftSize_ = fftSize; hopSize_ = hopSize; bins_ = fftSize / 2;
    }
    int getNumBins() const { return bins_; }
    int getFFTSize() const { return fftSize_; }
    int getHopSize() const { return hopSize_; }
    // frames every `frame_stride` samples of a device signal -> [nframes][bins] outputs (any may be null)
    void process(const float *d_signal, size_t frame_stride, size_t nframes, float *d_mags, float *d_phases,
                 float *d_real = nullptr, float *d_imag = nullptr, void *stream = nullptr) {
        maxigpu::check(mxg_fft_batch(plan_, d_signal, frame_stride, nframes, d_real, d_imag, d_mags, d_phases, stream), "mxg_fft_batch");
    }
    // magsToDB / spectralFlatness / spectralCentroid (L/maxiFFT.cpp:101-132) for every frame; any output may be null
    void features(const float *d_mags, size_t nframes, float *d_db, float *d_flatness, float *d_centroid, void *stream = nullptr) {
        maxigpu::check(mxg_fft_features(plan_, d_mags, nframes, d_db, d_flatness, d_centroid, stream), "mxg_fft_features");
    }

private:
    mxg_fft_plan *plan_ = nullptr;
    int fftSize_ = 0, hopSize_ = 0, bins_ = 0;
};

// maxiIFFT (L/maxiFFT.h:117-156), SPECTRUM mode, over batches of spectra; the overlap-add buffer is carried
class maxiIFFTBatch {
public:
    ~maxiIFFTBatch() { if (plan_) mxg_ifft_plan_destroy(plan_); }
    void setup(int fftSize = 1024, int hopSize = 512, int windowSize = 0) {  // L/maxiFFT.cpp:140-153
        if (plan_) mxg_ifft_plan_destroy(plan_);
        maxigpu::check(mxg_init(-1), "mxg_init");
        plan_ = mxg_ifft_plan_create(fftSize, hopSize, windowSize);
        if (!plan_) throw std::runtime_error(std::string("mxg_ifft_plan_create: ") + mxg_last_error());
        fftSize_ = fftSize; hopSize_ = hopSize;
        buffer_.resize((size_t)fftSize);  // zero-filled, as setup() leaves `buffer`
    }
    int getNumBins() const { return fftSize_ / 2; }
    // d_mags/d_phases [nframes][bins] -> d_signal [nframes*hopSize]: what process() returns, hopSize calls per spectrum
    void process(const float *d_mags, const float *d_phases, size_t nframes, float *d_signal, void *stream = nullptr) {
        maxigpu::check(mxg_ifft_batch(plan_, d_mags, d_phases, nframes, buffer_.get(), d_signal, nullptr, stream), "mxg_ifft_batch");
    }

private:
    mxg_ifft_plan *plan_ = nullptr;
    int fftSize_ = 0, hopSize_ = 0;
    maxigpu::DeviceArray<float> buffer_;
};

class maxiMFCCBatch {
public:
    ~maxiMFCCBatch() { if (plan_) mxg_mfcc_plan_destroy(plan_); }
    void setup(unsigned numBins, unsigned numFilters, unsigned numCoeffs, double minFreq, double maxFreq) {  // L/maxiMFCC.h:56-75
        if (plan_) mxg_mfcc_plan_destroy(plan_);
        plan_ = mxg_mfcc_plan_create(numBins, numFilters, numCoeffs, minFreq, maxFreq);
        if (!plan_) throw std::runtime_error(std::string("mxg_mfcc_plan_create: ") + mxg_last_error());
        numBins_ = numBins;
    }
    void mfcc(const float *d_mags, size_t nframes, double *d_mfcc, int method = 0, void *stream = nullptr) {  // L/maxiMFCC.h:77-81
        maxigpu::check(mxg_mfcc_batch(plan_, d_mags, numBins_, nframes, nullptr, nullptr, d_mfcc, method, stream), "mxg_mfcc_batch");
    }

private:
    mxg_mfcc_plan *plan_ = nullptr;
    unsigned numBins_ = 0;
};

// ---- maxiTimeStretch / maxiStretch banks (L/maxiGrains.h) ---------------------------------------------------------
class maxiTimeStretchBank {
public:
    maxiTimeStretchBank(size_t streams, maxiSampleBank *sample, int window_kind = 0)
        : S(streams), sample_(sample), window_(window_kind), st_(4 * streams), gst_(32 * streams), speed_(streams) {}
    ~maxiTimeStretchBank() { if (plan_) mxg_grain_plan_destroy(plan_); }
    void setPosition(const std::vector<double> &pos01) {  // L/maxiGrains.h:335-338
        std::vector<double> st = st_.download();
        const double len = (double)sample_->getLength();
        for (size_t s = 0; s < S; s++) {
            double p = pos01[s] * len;
            st[s] = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        }
        st_.upload(st);
    }
    void setSpeeds(const std::vector<double> &speed) { speed_.upload(speed); }
    // play(speed, grainLength, overlaps) for N samples of every stream -> d_out [N][S]
    void play(double grainLength, int overlaps, size_t N, double *d_out, void *stream = nullptr) {
        plan(grainLength);
        maxigpu::check(mxg_granular_render(plan_, mode_, S, N, sample_->deviceSamples(), sample_->getLength(), overlaps, speed_.get(),
                                           nullptr, nullptr, nullptr, 0, st_.get(), gst_.get(), d_out, stream), "mxg_granular_render");
    }
    // playAtPosition(pos, grainLength, overlaps) (L/maxiGrains.h:359-367): d_pos = the per-sample [N][S] position signal
    void playAtPosition(const double *d_pos, double grainLength, int overlaps, size_t N, double *d_out, void *stream = nullptr) {
        plan(grainLength);
        maxigpu::check(mxg_granular_render(plan_, 2, S, N, sample_->deviceSamples(), sample_->getLength(), overlaps, d_pos, nullptr,
                                           nullptr, nullptr, 0, st_.get(), gst_.get(), d_out, stream), "mxg_granular_render");
    }

protected:
    int mode_ = 0;
    mxg_grain_plan *plan_handle() const { return plan_; }
    const double *speeds() const { return speed_.get(); }
    double *state() { return st_.get(); }
    double *grains() { return gst_.get(); }
    void plan(double grainLength) {
        if (!plan_ || grainLength != grainLength_) {
            if (plan_) mxg_grain_plan_destroy(plan_);
            plan_ = mxg_grain_plan_create(window_, grainLength, sample_->mySampleRate);
            if (!plan_) throw std::runtime_error(std::string("mxg_grain_plan_create: ") + mxg_last_error());
            grainLength_ = grainLength;
        }
    }

private:
    size_t S;
    maxiSampleBank *sample_;
    int window_;
    mxg_grain_plan *plan_ = nullptr;
    double grainLength_ = 0;
    maxigpu::DeviceArray<double> st_, gst_, speed_;
};

// ---- maxiPitchShift bank (L/maxiGrains.h:374-432): same grains, speed uncoupled from position ----------------------
class maxiPitchShiftBank : public maxiTimeStretchBank {
public:
    maxiPitchShiftBank(size_t streams, maxiSampleBank *sample, int window_kind = 0)
        : maxiTimeStretchBank(streams, sample, window_kind) { mode_ = 3; }
};

// ---- maxiStretch bank (L/maxiGrains.h:437-542): pitch (grain speed) and time stretch uncoupled, the default loop ----------
class maxiStretchBank : public maxiTimeStretchBank {
public:
    maxiStretchBank(size_t streams, maxiSampleBank *sample, int window_kind = 0)
        : maxiTimeStretchBank(streams, sample, window_kind), nS_(streams), smp_(sample), rate_(streams) { mode_ = 1; }
    void setPitch(const std::vector<double> &pitchstretch) { setSpeeds(pitchstretch); }  // the first argument of play()
    void setRate(const std::vector<double> &timestretch) { rate_.upload(timestretch); }   // the second
    // play(pitchstretch, timestretch, grainLength, overlaps) for N samples of every stream -> d_out [N][S]; d_rnd (optional):
    // int32 [S][R], the values rand() % 10 returns at the spawns of each stream (NULL: 0, no jitter)
    void play(double grainLength, int overlaps, size_t N, double *d_out, const int32_t *d_rnd = nullptr, size_t R = 0,
              void *stream = nullptr) {
        plan(grainLength);
        maxigpu::check(mxg_granular_render(plan_handle(), 1, nS_, N, smp_->deviceSamples(), smp_->getLength(), overlaps, speeds(),
                                           rate_.get(), nullptr, d_rnd, R, state(), grains(), d_out, stream), "mxg_granular_render");
    }

private:
    size_t nS_;
    maxiSampleBank *smp_;
    maxigpu::DeviceArray<double> rate_;
};

// ---- maxiConvolve (L/maxiConvolve.h:19-34, maxiConvolve.cpp:13-107), block form --------------------------------------------
// setup() analyses the impulse (a maxiSampleBank's buffer, its play head where load() left it); play() takes nblocks * fftsize
// input samples at once.  as_intended = false reproduces what the reference computes (its COMPLEX-mode inverse never receives the
// sums: silence after the window), true what it was written to do; both bit-exact (include/maxigpu.h, mxg_convolve_play).
class maxiConvolveBlock {
public:
    ~maxiConvolveBlock() { if (c_) mxg_convolve_destroy(c_); }
    void setup(const std::vector<double> &impulse, double position0, int fftsize = 1024, int hopsize = 256) {
        if (c_) mxg_convolve_destroy(c_);
        c_ = mxg_convolve_create(impulse.data(), impulse.size(), position0, fftsize, hopsize);
        if (!c_) throw std::runtime_error(std::string("mxg_convolve_create: ") + mxg_last_error());
        fftsize_ = fftsize;
    }
    int frames() const { return c_ ? mxg_convolve_frames(c_) : 0; }
    int fftSize() const { return fftsize_; }
    void play(const float *d_in, size_t nblocks, float *d_out, bool as_intended = false, void *stream = nullptr) {
        maxigpu::check(mxg_convolve_play(c_, d_in, nblocks, d_out, as_intended ? 1 : 0, stream), "mxg_convolve_play");
    }
    void reset() { maxigpu::check(mxg_convolve_reset(c_), "mxg_convolve_reset"); }

private:
    mxg_convolve *c_ = nullptr;
    int fftsize_ = 0;
};

// ---- maxiSampler banks (L/maxiSynths.h:137-187, maxiSynths.cpp:262-491) ------------------------------------------------------
// NS samplers of `voices` slots each over one sample; the control methods edit host copies of the slot state between renders,
// exactly as the reference's methods edit its members; play(N) renders N calls of maxiSampler::play() for every sampler.
class maxiSamplerBank {
public:
    maxiSamplerBank(size_t samplers, int voices, maxiSampleBank *sample)
        : NS(samplers), voices_(voices), V(samplers * (size_t)voices), sample_(sample), pitch_(V, 0.0), gain_(V, 0.0), par_(4 * V),
          hold_(V, 1), position_(V, 0.0), trig_(V, 0), outhold_(V, 0.0), dst_(2 * V, 0.0), ist_(6 * V, 0), currentVoice_(samplers, 0),
          d_freq_(V), d_gain_(V), d_par_(4 * V), d_hold_(V), d_pos_(V), d_trig_(V), d_outhold_(V), d_dst_(2 * V), d_ist_(6 * V) {
        for (size_t v = 0; v < V; v++) {  // ctor, maxiSynths.cpp:262-283
            par_[0 * V + v] = mxg_env_coeff_host(0, 0);
            par_[1 * V + v] = mxg_env_coeff_host(1, 1);
            par_[2 * V + v] = 1.;
            par_[3 * V + v] = mxg_env_coeff_host(2, 2000);
            position_[v] = (double)sample->getLength();  // after load(); setSample leaves len - 1: use resetPositions()
        }
    }
    bool sustain = true;
    void resetPositions(double p) { pull(); std::fill(position_.begin(), position_.end(), p); dirty_ = true; }
    void setPitch(size_t sampler, double pitchIn, bool setall = false) { for (size_t v : slots(sampler, setall)) pitch_[v] = pitchIn; }
    void midiNoteOn(size_t sampler, double pitchIn, double velocity, bool setall = false) {  // :341-358
        for (size_t v : slots(sampler, setall)) {
            pitch_[v] = pitchIn;
            if (!setall) gain_[v] = velocity / 128;
        }
    }
    void midiNoteOff(size_t sampler, double pitchIn) {  // :360-372
        pull();
        for (int i = 0; i < voices_; i++)
            if (pitch_[sampler * voices_ + i] == pitchIn) trig_[sampler * voices_ + i] = 0;
        dirty_ = true;
    }
    void setEnvelope(size_t sampler, int which /*0 attack 1 decay 2 sustain 3 release*/, double value, bool setall = true) {
        const double c = which == 2 ? value : mxg_env_coeff_host(which == 3 ? 2 : which, value);
        for (size_t v : slots(sampler, setall)) par_[(size_t)which * V + v] = c;
    }
    void trigger(size_t sampler) {  // :484-491
        pull();
        const size_t v = sampler * voices_ + (size_t)currentVoice_[sampler];
        trig_[v] = 1;
        position_[v] = 0;
        currentVoice_[sampler] = (currentVoice_[sampler] + 1) % voices_;
        dirty_ = true;
    }
    // N calls of play() for every sampler -> d_mix [N][NS]
    void play(size_t N, double *d_mix, void *stream = nullptr) {
        if (dirty_) {
            d_pos_.upload(position_); d_trig_.upload(trig_); d_outhold_.upload(outhold_); d_dst_.upload(dst_); d_ist_.upload(ist_);
            dirty_ = false;
        }
        std::vector<double> freq(V);
        maxigpu::check(mxg_sampler_freq_host(V, pitch_.data(), sample_->getLength(), freq.data()), "mxg_sampler_freq_host");
        d_freq_.upload(freq); d_gain_.upload(gain_); d_par_.upload(par_); d_hold_.upload(hold_);
        maxigpu::check(mxg_sampler_render(V, N, voices_, sustain ? 1 : 0, sample_->deviceSamples(), sample_->getLength(), d_freq_.get(),
                                          d_gain_.get(), d_par_.get(), d_hold_.get(), d_pos_.get(), d_trig_.get(), d_outhold_.get(),
                                          d_dst_.get(), d_ist_.get(), d_mix, nullptr, stream), "mxg_sampler_render");
        fresh_ = false;
    }

private:
    size_t NS;
    int voices_;
    size_t V;
    maxiSampleBank *sample_;
    std::vector<double> pitch_, gain_, par_;
    std::vector<int64_t> hold_;
    std::vector<double> position_;
    std::vector<int32_t> trig_;
    std::vector<double> outhold_, dst_;
    std::vector<int64_t> ist_;
    std::vector<int> currentVoice_;
    maxigpu::DeviceArray<double> d_freq_, d_gain_, d_par_;
    maxigpu::DeviceArray<int64_t> d_hold_;
    maxigpu::DeviceArray<double> d_pos_;
    maxigpu::DeviceArray<int32_t> d_trig_;
    maxigpu::DeviceArray<double> d_outhold_, d_dst_;
    maxigpu::DeviceArray<int64_t> d_ist_;
    bool dirty_ = true, fresh_ = true;
    std::vector<size_t> slots(size_t sampler, bool setall) const {
        std::vector<size_t> r;
        if (setall) for (int i = 0; i < voices_; i++) r.push_back(sampler * voices_ + (size_t)i);
        else r.push_back(sampler * voices_ + (size_t)currentVoice_[sampler]);
        return r;
    }
    void pull() {  // the device state back to the host copies (after a render)
        if (dirty_ || fresh_) return;
        position_ = d_pos_.download(); trig_ = d_trig_.download(); outhold_ = d_outhold_.download();
        dst_ = d_dst_.download(); ist_ = d_ist_.download();
    }
};
