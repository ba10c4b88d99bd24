// include/maxiConvolve.h -- DROP-IN for the reference's partitioned convolver (src/libs/maxiConvolve.h:19-34,
// maxiConvolve.cpp:13-107): setup(impulseFile, fftsize, hopsize) and float play(float w), one call per sample, over the C-ABI's
// mxg_convolve (impulse analysis, frequency delay line, complex multiply-accumulate and maxiIFFT on the device).
//
// One call per sample against a block renderer: the output of the fftsize samples of block b depends only on the sums the
// input frame b-1 left behind (maxiIFFT consumes a frame's sums from the sample after it on), so at the first sample of a block
// the whole output block is fetched (mxg_convolve_output), the block's inputs are collected as they come, and the last call of
// the block hands them to the delay line (mxg_convolve_input) -- two launches per fftsize samples, nothing computed on the CPU.
//
// maxiConvolve::asIntended (default false).  In COMPLEX mode the reference's maxiIFFT never receives the sums (fft::
// inverseFFTComplex writes them into the OUTPUT arrays that calcIFFT then overwrites, L/fft.cpp:613-619): its play() returns
// silence after the first window, and that is what this class returns bit for bit by default; asIntended = true routes the sums
// to the inverse transform (the convolution the class was written for; bit-exact against the reference's own transform fed
// that way, tests/test_gpu_convolve.py).
#pragma once
#include "maximilian.h"

class maxiConvolve {
    mxg_convolve *c_ = nullptr;
    int F_ = 0;
    std::vector<float> in_, out_;
    size_t pos_ = 0;
    float *d_io_ = nullptr;  // [2][fftsize]

public:
    static inline bool asIntended = false;
    maxiConvolve() = default;
    ~maxiConvolve() {
        if (c_) mxg_convolve_destroy(c_);
        if (d_io_) mxg_free(d_io_);
    }
    maxiConvolve(const maxiConvolve &) = delete;
    maxiConvolve &operator=(const maxiConvolve &) = delete;
    void setup(std::string impulseFile, int fftsize = 1024, int hopsize = 256) {  // L/maxiConvolve.cpp:13-74
        maxiSample impulse;
        impulse.load(impulseFile);
        setup(impulse, fftsize, hopsize);
    }
    // the same from a sample already in memory (its play head where load() / setSample() left it)
    void setup(maxiSample &impulse, int fftsize = 1024, int hopsize = 256, bool loaded = true) {
        MAXIGPU_TRY {
        if (c_) mxg_convolve_destroy(c_);
        const vector<double> &amps = impulse.getAmplitudes();
        const double position0 = loaded ? (double)amps.size() : (double)amps.size() - 1;  // C:681 / H:677
        c_ = mxg_convolve_create(amps.data(), amps.size(), position0, fftsize, hopsize);
        if (!c_) maxigpu::ps::fatal(std::string("mxg_convolve_create: ") + mxg_last_error());
        F_ = fftsize;
        in_.assign((size_t)F_, 0.0f);
        out_.assign((size_t)F_, 0.0f);
        pos_ = 0;
        if (d_io_) mxg_free(d_io_);
        d_io_ = static_cast<float *>(mxg_malloc(sizeof(float) * 2 * (size_t)F_));
        if (!d_io_) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        }
        MAXIGPU_CATCH(return)
    }
    float play(float w) {  // L/maxiConvolve.cpp:76-107
        if (!c_ || maxigpu::ps::dead()) return 0.0f;  // (a dead device path: silence, nothing touches the device)
        if (pos_ == 0) {
            maxigpu::ps::check(mxg_convolve_output(c_, d_io_ + F_, asIntended ? 1 : 0, nullptr), "mxg_convolve_output");
            maxigpu::ps::check(mxg_memcpy_d2h(out_.data(), d_io_ + F_, sizeof(float) * (size_t)F_, nullptr), "d2h convolve block");
        }
        const float o = out_[pos_];
        in_[pos_] = w;
        if (++pos_ == (size_t)F_) {
            maxigpu::ps::check(mxg_memcpy_h2d(d_io_, in_.data(), sizeof(float) * (size_t)F_, nullptr), "h2d convolve block");
            maxigpu::ps::check(mxg_convolve_input(c_, d_io_, nullptr), "mxg_convolve_input");
            pos_ = 0;
        }
        return o;
    }
    int frames() const { return c_ ? mxg_convolve_frames(c_) : 0; }  // impulseReal.size()
};
