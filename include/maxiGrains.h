// include/maxiGrains.h -- DROP-IN for the reference's granular classes (src/libs/maxiGrains.h): maxiTimeStretch<F>,
// maxiPitchShift<F>, maxiStretch<F> with the reference's window functor names as template parameters, one call per sample,
// executed through the C-ABI's granular renderer (mxg_granular_render, include/maxigpu.h; kernels K8 .. K8d).
//
// How a per-sample `ts->play(speed, grainLength, overlaps, posMod)` is served.  A stream (one object) is a scheduler -- position,
// looper / cycles, randomOffset (L/maxiGrains.h:341-355, 412-430, 512-530) -- plus at most eight live grains; both live on the host
// between launches (4 + 32 doubles) and a launch renders a BLOCK of samples for the arguments of the current call, which later
// calls with the same arguments read back.  The reference draws `randomOffset = rand() % 10` from the process-wide rand() stream
// at every spawn (:352, :525), in call order with every other user of rand() in the patch (maxiOsc::noise ...).  To keep that order
// exact, a block never runs past a spawn: the host counts, with the scheduler's own `looper++` additions, how many samples it
// takes until `looper > cycleLength + randomOffset`, renders up to and including that sample (the grain is born with the kernel's
// placeholder draw, which only matters from the NEXT sample on) and draws the real rand() when the call of that sample is served.
// maxiPitchShift draws nothing and renders 512-sample blocks.  A call with other arguments rewinds the stream to the state it had
// at that sample (the block is re-rendered from its start state up to there: same kernel, same bits).
// Grain arithmetic is bit-exact (tests/test_gpu_dropin.py compiles tests/patches/granular_patch.cpp both ways).
#pragma once
#include "maximilian.h"

// window functors (L/maxiGrains.h:18-90): the names select the plan's window table (built with host libm by the library)
struct hannWinFunctor { static constexpr int kind = 0; };
struct hammingWinFunctor { static constexpr int kind = 1; };
struct cosineWinFunctor { static constexpr int kind = 2; };
struct rectWinFunctor { static constexpr int kind = 3; };
struct triangleWinFunctor { static constexpr int kind = 4; };
struct triangleNZWinFunctor { static constexpr int kind = 5; };
struct blackmanHarrisWinFunctor { static constexpr int kind = 6; };
struct blackmanNutallWinFunctor { static constexpr int kind = 7; };
struct gaussianWinFunctor { static constexpr int kind = 8; };

namespace maxigpu {
namespace ps {

// one granular stream behind the C-ABI: MODE 0 maxiTimeStretch::play, 1 maxiStretch::play, 2 playAtPosition, 3 maxiPitchShift::play
class GrainStream {
public:
    GrainStream(int window_kind) : kind_(window_kind) {}
    ~GrainStream() {
        for (auto &p : plans_) mxg_grain_plan_destroy(p.second);
        if (d_) mxg_free(d_);
    }
    GrainStream(const GrainStream &) = delete;
    GrainStream &operator=(const GrainStream &) = delete;
    void setSample(maxiSample *s) {  // a new grain player: no live grains (L/maxiGrains.h:313-323)
        sample_ = s;
        std::fill(gst_, gst_ + 32, 0.0);
        drop_block();
    }
    maxiSample *sample() const { return sample_; }
    // scheduler members (host-authoritative between launches)
    double &position() { settle(); return st_[0]; }
    double &looper() { settle(); return st_[1]; }
    double &randomOffset() { settle(); return st_[2]; }
    size_t launches = 0;

    double play(int mode, double a, double b, double grainLength, int overlaps, double posMod) {
        if (dead() || !sample_ || !sample_->deviceSamples() || sample_->getLength() == 0) return 0.0;
        Call c;
        c.method = mode;
        c.a[0] = a; c.a[1] = b; c.a[2] = grainLength; c.a[3] = (double)overlaps; c.a[4] = posMod;
        if (pos_ < blk_.size() && sig_.same(c)) return serve();
        // miss: the state at the current sample, then a fresh block under the new arguments
        settle();
        sig_ = c;
        render(next_len(c));
        // (a refused call -- e.g. more overlaps than the renderer's eight live grains -- or a device failure leaves no block: silence)
        if (blk_.empty() || pos_ >= blk_.size() || dead()) return 0.0;
        return serve();
    }

private:
    int kind_;
    maxiSample *sample_ = nullptr;
    std::vector<std::pair<std::pair<double, int>, mxg_grain_plan *>> plans_;
    double st_[4] = {0, 0, 0, 0}, gst_[32] = {0};        // state at the block's first sample
    double est_[4] = {0, 0, 0, 0}, egst_[32] = {0};      // ... after its last
    std::vector<double> blk_;
    size_t pos_ = 0;
    Call sig_;
    bool spawn_at_end_ = false;  // the block's last sample is a spawn of mode 0 / 1: its rand() is drawn when that call is served
    double *d_ = nullptr;        // device scratch: [st 4][gst 32][par 3][out kMaxBlock] + int32 rnd
    static constexpr size_t kDoubles = 4 + 32 + 3 + kMaxBlock + 2;

    mxg_grain_plan *plan(double grainLength) {
        MAXIGPU_TRY {
        const std::pair<double, int> key(grainLength, sample_->mySampleRate);
        for (auto &p : plans_)
            if (p.first == key) return p.second;
        mxg_grain_plan *pl = mxg_grain_plan_create(kind_, grainLength, sample_->mySampleRate);
        if (!pl) maxigpu::ps::fatal(std::string("mxg_grain_plan_create: ") + mxg_last_error());
        plans_.push_back(std::make_pair(key, pl));
        return pl;
        }
        MAXIGPU_CATCH(return nullptr)
    }
    void drop_block() {
        blk_.clear();
        pos_ = 0;
        spawn_at_end_ = false;
        sig_.method = -1;
    }
    // samples until (and including) the next spawn of modes 0 / 1, counted with the scheduler's own additions (:343, :515: looper++)
    size_t next_len(const Call &c) const {
        if (c.method != 0 && c.method != 1) return c.method == 2 ? 1 : kMaxBlock;  // playAtPosition: its position argument is a signal
        const double cycleLength = c.a[2] * maxiSettings::sampleRate / (int)c.a[3];  // :346 / :518
        const double thr = cycleLength + st_[2];
        double l = st_[1];
        size_t k = 0;
        do {
            l = l + 1.0;
            k++;
        } while (!(l > thr) && k < kMaxBlock);
        return k;
    }
    void render(size_t L) {
        MAXIGPU_TRY {
        check(mxg_init(-1), "mxg_init");
        if (!d_) {
            d_ = static_cast<double *>(mxg_malloc(sizeof(double) * kDoubles));
            if (!d_) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        }
        double *d_st = d_, *d_gst = d_ + 4, *d_par = d_ + 36, *d_out = d_ + 39;
        int32_t *d_rnd = reinterpret_cast<int32_t *>(d_ + 39 + kMaxBlock);
        double h[39];
        for (int i = 0; i < 4; i++) h[i] = st_[i];
        h[3] = 0.0;  // the rand cursor: one placeholder draw per launch
        for (int i = 0; i < 32; i++) h[4 + i] = gst_[i];
        h[36] = sig_.a[0]; h[37] = sig_.a[1]; h[38] = sig_.a[4];
        const int32_t zero = 0;
        bool ok = check(mxg_memcpy_h2d(d_, h, sizeof(h), nullptr), "h2d grain state");
        ok = check(mxg_memcpy_h2d(d_rnd, &zero, sizeof(zero), nullptr), "h2d grain draw") && ok;
        const int mode = sig_.method;
        ok = check(mxg_granular_render(plan(sig_.a[2]), mode, 1, L, sample_->deviceSamples(), sample_->getLength(), (int)sig_.a[3], d_par,
                                       mode == 1 ? d_par + 1 : nullptr, mode == 2 ? nullptr : d_par + 2, (mode == 0 || mode == 1) ? d_rnd : nullptr,
                                       1, d_st, d_gst, d_out, nullptr), "mxg_granular_render") && ok;
        double back[36] = {0};
        blk_.assign(L, 0.0);
        ok = check(mxg_memcpy_d2h(back, d_, sizeof(back), nullptr), "d2h grain state") && ok;  // (synchronises; reports a deferred render error)
        ok = check(mxg_memcpy_d2h(blk_.data(), d_out, sizeof(double) * L, nullptr), "d2h grain block") && ok;
        if (!ok) {  // nothing of this launch is used: the stream keeps the state it had, the call returns silence
            abandon();
            return;
        }
        for (int i = 0; i < 4; i++) est_[i] = back[i];
        for (int i = 0; i < 32; i++) egst_[i] = back[4 + i];
        spawn_at_end_ = (mode == 0 || mode == 1) && back[3] != 0.0;  // the kernel consumed the placeholder: a grain was born in the block
        pos_ = 0;
        launches++;
        }
        MAXIGPU_CATCH(abandon(); return)
    }
    void abandon() {  // a launch that failed or was refused: no block, the end state = the start state
        for (int i = 0; i < 4; i++) est_[i] = st_[i];
        for (int i = 0; i < 32; i++) egst_[i] = gst_[i];
        drop_block();
    }
    double serve() {
        const double out = blk_[pos_++];
        if (pos_ == blk_.size()) {  // the block is used up: its end state is the stream's state
            for (int i = 0; i < 4; i++) st_[i] = est_[i];
            for (int i = 0; i < 32; i++) gst_[i] = egst_[i];
            if (spawn_at_end_) st_[2] = (double)(rand() % 10);  // :352 / :525, drawn in call order
            spawn_at_end_ = false;
            blk_.clear();
            pos_ = 0;
        }
        return out;
    }
    // the state AT the current sample (a block partly served is re-rendered from its start state up to here)
    void settle() {
        if (blk_.empty()) return;
        if (pos_ > 0) {
            const Call keep = sig_;
            render(pos_);  // same start state, same arguments, pos_ samples: cannot contain the spawn (that is the block's last sample)
            for (int i = 0; i < 4; i++) st_[i] = est_[i];
            for (int i = 0; i < 32; i++) gst_[i] = egst_[i];
            sig_ = keep;
        }
        drop_block();
    }
};

}  // namespace ps
}  // namespace maxigpu

// ---- maxiTimeStretch<F> (L/maxiGrains.h:287-368) -------------------------------------------------------------------------
template <typename F>
class maxiTimeStretch {
protected:
    maxigpu::ps::GrainStream g_{F::kind};

public:
    maxiSample *sample = nullptr;
    maxiTimeStretch() {}
    maxiTimeStretch(maxiSample *sample_) : sample(sample_) { g_.setSample(sample_); }
    void setSample(maxiSample *sampleIn) {
        sample = sampleIn;
        g_.setSample(sampleIn);
    }
    double getNormalisedPosition() { return g_.position() / (double)sample->getLength(); }
    double getPosition() { return g_.position(); }
    void setPosition(double pos) {  // :334-337
        double p = pos * sample->getLength();
        g_.position() = maxiMap::clamp(p, 0, sample->getLength() - 1);
    }
    inline double play(double speed = 1, double grainLength = 0.05, int overlaps = 2, double posMod = 0) {
        return g_.play(0, speed, 0.0, grainLength, overlaps, posMod);
    }
    inline double playAtPosition(double pos, double grainLength, int overlaps) { return g_.play(2, pos, 0.0, grainLength, overlaps, 0.0); }
    size_t launches() const { return g_.launches; }
};

// ---- maxiPitchShift<F> (L/maxiGrains.h:374-432) --------------------------------------------------------------------------
template <typename F>
class maxiPitchShift {
    maxigpu::ps::GrainStream g_{F::kind};

public:
    maxiSample *sample = nullptr;
    maxiPitchShift() {}
    maxiPitchShift(maxiSample *sample_) : sample(sample_) { g_.setSample(sample_); }
    void setSample(maxiSample *sampleIn) {
        sample = sampleIn;
        g_.setSample(sampleIn);
    }
    double play(double speed, double grainLength, int overlaps, double posMod = 0.0) { return g_.play(3, speed, 0.0, grainLength, overlaps, posMod); }
    size_t launches() const { return g_.launches; }
};

// ---- maxiStretch<F> (L/maxiGrains.h:437-542), the default loop (the whole sample) --------------------------------------------
template <typename F>
class maxiStretch {
    maxigpu::ps::GrainStream g_{F::kind};

public:
    maxiSample *sample = nullptr;
    maxiStretch() {}
    maxiStretch(maxiSample *sample_) : sample(sample_) { g_.setSample(sample_); }
    void setSample(maxiSample *newSample) {
        sample = newSample;
        g_.setSample(newSample);
        g_.position() = 0;
        g_.looper() = 0;
    }
    double getNormalisedPosition() { return g_.position() / (double)sample->getLength(); }
    double getPosition() { return g_.position(); }
    void setPosition(double pos) {  // :488-491
        double p = pos * sample->getLength();
        g_.position() = maxiMap::clamp(p, 0, sample->getLength() - 1);
    }
    unsigned long getLoopEnd() { return sample ? (unsigned long)sample->getLength() : 0; }
    inline double play(double pitchstretch = 1, double timestretch = 1, double grainLength = 0.05, int overlaps = 2, double posMod = 0.0) {
        if (sample == nullptr) return 0;  // :513
        return g_.play(1, pitchstretch, timestretch, grainLength, overlaps, posMod);
    }
    size_t launches() const { return g_.launches; }
};
