// include/maxiSynths.h -- DROP-IN for the reference's polyphonic sampler maxiSampler (src/libs/maxiSynths.h:137-187,
// maxiSynths.cpp:262-491): up to 32 slots of maxiEnv::adsr x maxiSample::play4 over one sample, summed in slot order, one
// play() per sample, executed through the C-ABI's mxg_sampler_render (sampler.hip; bit-exact incl. the in-order sum).
//
// play() is served from blocks: as long as no control method is called (trigger, midiNoteOn / Off, setPitch, the envelope
// setters ...) the slot parameters cannot change, so a launch renders the next 1, 2, 4 ... 512 calls at once and play() reads them
// back; a control call first rewinds the sampler to the state it had at the current sample (the block is re-rendered from its
// start state up to there: same kernel, same bits), then edits that state on the host exactly as the reference's method does.
// Limits against the reference: every slot plays the SAME sample (load / setSample with setall, the reference's default; a
// per-slot load is refused with a printed error), and the unused members (LFO1-4, filters, distortion: never touched by maxiSampler::play) are absent.
#pragma once
#include "maximilian.h"

class maxiSampler {
    static constexpr int kMax = 32;
    struct State {
        double position[kMax], outhold[kMax], dst[2 * kMax];
        int32_t trig[kMax];
        int64_t ist[6 * kMax];
    };
    State cur_ = {}, end_ = {};
    double par_[4 * kMax], gain_[kMax];
    int64_t hold_[kMax];
    std::vector<double> blk_;
    size_t pos_ = 0, nextLen_ = 1;
    double *d_samples_ = nullptr;
    size_t len_ = 0;
    char *d_ = nullptr;  // device scratch, laid out in layout()
    bool dirty_ = true;  // parameters / state edited on the host since the last launch

    struct Lay { double *freq, *gain, *par, *pos, *outhold, *dst, *mix; int64_t *hold, *ist; int32_t *trig; };
    Lay layout() const {
        Lay l;
        double *p = reinterpret_cast<double *>(d_);
        l.freq = p; p += kMax;
        l.gain = p; p += kMax;
        l.par = p; p += 4 * kMax;
        l.pos = p; p += kMax;
        l.outhold = p; p += kMax;
        l.dst = p; p += 2 * kMax;
        l.mix = p; p += maxigpu::ps::kMaxBlock;
        l.hold = reinterpret_cast<int64_t *>(p); p += kMax;
        l.ist = reinterpret_cast<int64_t *>(p); p += 6 * kMax;
        l.trig = reinterpret_cast<int32_t *>(p);
        return l;
    }
    static constexpr size_t kBytes = sizeof(double) * (kMax * 10 + maxigpu::ps::kMaxBlock + kMax + 6 * kMax) + sizeof(int32_t) * kMax + 64;

    // state rows are [row][voices] on the device (V = voices): repack from the fixed-size host arrays
    void upload(const State &s) {
        MAXIGPU_TRY {
        using maxigpu::ps::check;
        const size_t V = (size_t)voices;
        if (!d_) {
            check(mxg_init(-1), "mxg_init");
            d_ = static_cast<char *>(mxg_malloc(kBytes));
            if (!d_) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        }
        const Lay l = layout();
        std::vector<double> dst(2 * V);
        std::vector<int64_t> ist(6 * V);
        for (size_t v = 0; v < V; v++) {
            for (int r = 0; r < 2; r++) dst[(size_t)r * V + v] = s.dst[r * kMax + (int)v];
            for (int r = 0; r < 6; r++) ist[(size_t)r * V + v] = s.ist[r * kMax + (int)v];
        }
        check(mxg_memcpy_h2d(l.pos, s.position, sizeof(double) * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.outhold, s.outhold, sizeof(double) * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.trig, s.trig, sizeof(int32_t) * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.dst, dst.data(), sizeof(double) * 2 * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.ist, ist.data(), sizeof(int64_t) * 6 * V, nullptr), "h2d");
        // parameters
        std::vector<double> par(4 * V), freq(V);
        for (size_t v = 0; v < V; v++)
            for (int r = 0; r < 4; r++) par[(size_t)r * V + v] = par_[r * kMax + (int)v];
        check(mxg_sampler_freq_host(V, pitch, len_, freq.data()), "mxg_sampler_freq_host");
        check(mxg_memcpy_h2d(l.freq, freq.data(), sizeof(double) * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.gain, gain_, sizeof(double) * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.par, par.data(), sizeof(double) * 4 * V, nullptr), "h2d");
        check(mxg_memcpy_h2d(l.hold, hold_, sizeof(int64_t) * V, nullptr), "h2d");
        }
        MAXIGPU_CATCH(return)
    }
    void download(State &s) {
        using maxigpu::ps::check;
        const size_t V = (size_t)voices;
        const Lay l = layout();
        std::vector<double> dst(2 * V);
        std::vector<int64_t> ist(6 * V);
        check(mxg_memcpy_d2h(s.position, l.pos, sizeof(double) * V, nullptr), "d2h");
        check(mxg_memcpy_d2h(s.outhold, l.outhold, sizeof(double) * V, nullptr), "d2h");
        check(mxg_memcpy_d2h(s.trig, l.trig, sizeof(int32_t) * V, nullptr), "d2h");
        check(mxg_memcpy_d2h(dst.data(), l.dst, sizeof(double) * 2 * V, nullptr), "d2h");
        check(mxg_memcpy_d2h(ist.data(), l.ist, sizeof(int64_t) * 6 * V, nullptr), "d2h");
        for (size_t v = 0; v < V; v++) {
            for (int r = 0; r < 2; r++) s.dst[r * kMax + (int)v] = dst[(size_t)r * V + v];
            for (int r = 0; r < 6; r++) s.ist[r * kMax + (int)v] = ist[(size_t)r * V + v];
        }
    }
    // render L calls of play() from cur_; leaves the state after them in end_ and the outputs in blk_
    void render(size_t L) {
        using maxigpu::ps::check;
        upload(cur_);
        const Lay l = layout();
        check(mxg_sampler_render((size_t)voices, L, voices, sustain ? 1 : 0, d_samples_, len_, l.freq, l.gain, l.par, l.hold, l.pos, l.trig,
                                 l.outhold, l.dst, l.ist, l.mix, nullptr, nullptr), "mxg_sampler_render");
        blk_.resize(L);
        check(mxg_memcpy_d2h(blk_.data(), l.mix, sizeof(double) * L, nullptr), "d2h sampler block");
        end_ = cur_;
        download(end_);
        pos_ = 0;
        launches++;
    }
    // the state at the current sample, cached block dropped: before any control method edits it
    void settle() {
        if (maxigpu::ps::dead()) {
            blk_.clear();
            pos_ = 0;
            return;
        }
        if (!blk_.empty()) {
            if (pos_ == blk_.size()) {
                cur_ = end_;
            } else if (pos_ > 0) {
                render(pos_);
                cur_ = end_;
            }
        }
        blk_.clear();
        pos_ = 0;
        nextLen_ = 1;
    }
    bool valid() const { return voices >= 1 && voices <= kMax; }  // (the reference's arrays hold 32 slots: beyond that it reads out of bounds)

public:
    size_t launches = 0;
    // the reference's public members that its own methods use
    double pitch[32];
    int originalPitch = 67;
    double output = 0;
    double gain = 1;
    int voices = 32;
    int currentVoice = 0;
    bool sustain = true;
    maxiSampler() {  // maxiSynths.cpp:262-283
        for (int i = 0; i < kMax; i++) {
            par_[0 * kMax + i] = mxg_env_coeff_host(0, 0);     // setAttack(0)
            par_[1 * kMax + i] = mxg_env_coeff_host(1, 1);     // setDecay(1)
            par_[2 * kMax + i] = 1.;                            // setSustain(1.)
            par_[3 * kMax + i] = mxg_env_coeff_host(2, 2000);  // setRelease(2000)
            hold_[i] = 1;
            gain_[i] = 0;  // envOutGain: uninitialised in the reference (static storage: 0)
            pitch[i] = 0;
        }
    }
    ~maxiSampler() {
        if (d_samples_) mxg_sample_free(d_samples_);
        if (d_) mxg_free(d_);
    }
    maxiSampler(const maxiSampler &) = delete;
    maxiSampler &operator=(const maxiSampler &) = delete;
    void setNumVoices(int numVoices) {
        settle();
        voices = numVoices;  // maxiSynths.cpp:284-289: any count; the slot arrays hold 32
        if (!valid()) maxigpu::ps::complain("maxiSampler::setNumVoices: 1 .. 32 voices (the reference's slot arrays hold 32) -- other counts play silence");
    }
    void load(string inFile, bool setall = true) {  // maxiSynths.cpp:303-321: samples[i].load(inFile) for every slot
        if (!setall) {  // (the reference loads the file into slot `currentVoice` only)
            maxigpu::ps::complain("maxiSampler::load(file, false): one sample per sampler on this backend (see include/maxiSynths.h) -- not loaded");
            return;
        }
        settle();
        if (d_samples_) mxg_sample_free(d_samples_);
        int32_t hdr[8];
        d_samples_ = mxg_sample_load_wav(inFile.c_str(), 0, &len_, hdr);
        if (!d_samples_) {
            printf("ERROR: Could not load sample.");
            len_ = 0;
            return;
        }
        for (int i = 0; i < kMax; i++) cur_.position[i] = (double)len_;  // maxiSample::read leaves position = size, C:681
    }
    void setSample(vector<double> &sampleData) {  // every samples[i].setSample(sampleData) (H:670-678)
        MAXIGPU_TRY {
        settle();
        if (d_samples_) mxg_sample_free(d_samples_);
        d_samples_ = mxg_sample_upload(sampleData.data(), sampleData.size());
        if (!d_samples_) maxigpu::ps::fatal(std::string("mxg_sample_upload: ") + mxg_last_error());
        len_ = sampleData.size();
        for (int i = 0; i < kMax; i++) cur_.position[i] = (double)len_ - 1;
        }
        MAXIGPU_CATCH(return)
    }
    double play() {  // maxiSynths.cpp:291-311
        if (!d_samples_ || !len_ || !valid() || maxigpu::ps::dead()) return output = 0;  // (a dead device path: silence, nothing touches the device)
        if (pos_ >= blk_.size()) {
            if (!blk_.empty()) cur_ = end_;
            const size_t L = blk_.empty() ? nextLen_ : std::min(2 * blk_.size(), maxigpu::ps::kMaxBlock);
            render(L);
            nextLen_ = L;
        }
        return output = blk_[pos_++];
    }
    void setPitch(double pitchIn, bool setall = false) {  // :323-339
        settle();
        if (setall) for (int i = 0; i < voices; i++) pitch[i] = pitchIn;
        else pitch[currentVoice] = pitchIn;
    }
    void midiNoteOn(double pitchIn, double velocity, bool setall = false) {  // :341-358
        settle();
        if (setall) {
            for (int i = 0; i < voices; i++) pitch[i] = pitchIn;
        } else {
            pitch[currentVoice] = pitchIn;
            gain_[currentVoice] = velocity / 128;
        }
    }
    void midiNoteOff(double pitchIn, double velocity, bool setall = false) {  // :360-372
        (void)velocity; (void)setall;
        settle();
        for (int i = 0; i < voices; i++)
            if (pitch[i] == pitchIn) cur_.trig[i] = 0;
    }
    void setAttack(double attackD, bool setall = true) { set_par(0, mxg_env_coeff_host(0, attackD), setall); }
    void setDecay(double decayD, bool setall = true) { set_par(1, mxg_env_coeff_host(1, decayD), setall); }
    void setSustain(double sustainD, bool setall = true) { set_par(2, sustainD, setall); }
    void setRelease(double releaseD, bool setall = true) { set_par(3, mxg_env_coeff_host(2, releaseD), setall); }
    void setPosition(double positionD, bool setall = true) {  // samples[i].setPosition (C:749-751)
        settle();
        const double p = (positionD < 0.0 ? 0.0 : (positionD > 1.0 ? 1.0 : positionD)) * (double)len_;
        if (setall) for (int i = 0; i < voices; i++) cur_.position[i] = p;
        else cur_.position[currentVoice] = p;
    }
    void trigger() {  // :484-491
        settle();
        cur_.trig[currentVoice] = 1;
        cur_.position[currentVoice] = 0;  // maxiSample::trigger, C:597-600
        currentVoice++;
        currentVoice = currentVoice % voices;
    }

private:
    void set_par(int row, double value, bool setall) {
        settle();
        if (setall) for (int i = 0; i < voices; i++) par_[row * kMax + i] = value;
        else par_[row * kMax + currentVoice] = value;
    }
};
