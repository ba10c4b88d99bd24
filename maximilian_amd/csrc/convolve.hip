// convolve.hip -- maxiConvolve (L/maxiConvolve.cpp:13-107): partitioned convolution over a frequency delay line (FDL).
//
// Reference: setup() analyses the impulse into frames of real/imaginary spectra (maxiFFT::setup(fftsize, fftsize,
// hopsize): hop = fftsize, so the frames do not overlap), normalises them by their largest positive value, and play(w)
// (a) transforms every fftsize input samples, (b) pushes the spectrum onto the FDL, (c) forms
//     sum[i] = sum_k impulse[k][i] (x) FDL[k][i]   (complex product; bin 0: real*real, imag*imag), float, k ascending,
// (d) hands the sums to maxiIFFT in COMPLEX mode once per fftsize samples.
// Here a call covers nblocks input frames at once:
//   frames  -> mxg_fft_batch (K6a/K6b: real/imag bit-exact) straight into the history buffer behind the carried FDL tail
//   sums    -> conv_mac_kernel: one lane per (new frame m, bin i), k walked in order (the reference's float op sequence,
//              no contraction); impulse row k is read coalesced over i and is shared by every m (L2-resident)
//   inverse -> mxg_ifft_batch_complex on [carried sums, sums of frames 0 .. nblocks-2]: block b of the output belongs to
//              the sums formed at the end of block b-1 (maxiIFFT consumes its inputs when its pos wraps to 0, i.e. one
//              sample AFTER the frame that produced them completed)
// mode 0 reproduces the reference's COMPLEX-mode defect (see maxigpu.h): silence, state still advancing.
#include <string.h>

#include <vector>

#include "mxg_spectral.h"

struct mxg_convolve {
    int fftsize, hopsize, bins, frames;
    mxg_fft_plan *fplan;
    mxg_ifft_plan *iplan;
    std::vector<float> h_impR, h_impI;
    float *d_impR, *d_impI;    // [frames][bins]
    float *d_tailR, *d_tailI;  // the frames-1 most recent input spectra, oldest first
    float *d_sumR, *d_sumI;    // sums formed by the last input frame (consumed by the next block's inverse transform)
    float *d_obuf;             // maxiIFFT::buffer
};

namespace mxg {
namespace {

// S[m+1][i] = sum_k imp[k][i] (x) H[(frames-1+m) - k][i], k = 0 .. frames-1 (H = carried tail followed by the new frames)
__global__ __launch_bounds__(256) void conv_mac_kernel(int frames, int bins, size_t nblocks, const float *__restrict__ impR,
                                                       const float *__restrict__ impI, const float *__restrict__ HR,
                                                       const float *__restrict__ HI, float *__restrict__ SR,
                                                       float *__restrict__ SI) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * (size_t)bins) return;
    const size_t m = t / bins;
    const int i = (int)(t - m * bins);
    float sr = 0.0f, si = 0.0f;  // std::fill(sumReal ...), L/maxiConvolve.cpp:89-90
    const size_t newest = (size_t)(frames - 1) + m;
    for (int k = 0; k < frames; k++) {
        const float ir = impR[(size_t)k * bins + i], ii = impI[(size_t)k * bins + i];
        const float fr = HR[(newest - k) * bins + i], fi = HI[(newest - k) * bins + i];
        if (i == 0) {
            sr += (ir * fr);  // :95
            si += (ii * fi);  // :96
        } else {
            sr += (ir * fr) - (ii * fi);  // :98
            si += (ir * fi) + (ii * fr);  // :99
        }
    }
    SR[(m + 1) * bins + i] = sr;
    SI[(m + 1) * bins + i] = si;
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

int mxg_convolve_destroy(mxg_convolve *c) {
    if (!c) return MXG_OK;
    if (c->fplan) mxg_fft_plan_destroy(c->fplan);
    if (c->iplan) mxg_ifft_plan_destroy(c->iplan);
    float *ptrs[] = {c->d_impR, c->d_impI, c->d_tailR, c->d_tailI, c->d_sumR, c->d_sumI, c->d_obuf};
    for (float *p : ptrs)
        if (p) (void)hipFree(p);
    delete c;
    return MXG_OK;
}

mxg_convolve *mxg_convolve_create(const double *h_amp, size_t len, double position0, int fftsize, int hopsize) {
    if (ensure_init_only()) return nullptr;
    if (!h_amp || len == 0 || fftsize < 8 || fftsize > 8192 || (fftsize & (fftsize - 1)) || hopsize <= 0 || hopsize > fftsize) {
        fail(MXG_ERR_INVALID, "mxg_convolve_create: bad impulse / fftsize %d / hopsize %d", fftsize, hopsize);
        return nullptr;
    }
    mxg_convolve *c = new mxg_convolve();
    c->fftsize = fftsize; c->hopsize = hopsize; c->bins = fftsize / 2; c->frames = 0;
    c->fplan = nullptr; c->iplan = nullptr;
    c->d_impR = c->d_impI = c->d_tailR = c->d_tailI = c->d_sumR = c->d_sumI = c->d_obuf = nullptr;
    const int F = fftsize, bins = c->bins;
    // the float stream setup() feeds its maxiFFT: impulse.play() x len (C:740-747), then the zero padding (:41-45)
    const size_t total = len + (size_t)(bins - (int)(len % (size_t)bins));
    std::vector<float> seq(total, 0.0f);
    double position = position0;
    for (size_t s = 0; s < len; s++) {
        const long idx = (long)position;
        seq[s] = (idx >= 0 && (size_t)idx < len) ? (float)h_amp[idx] : 0.0f;  // amplitudes[size] and beyond: the guard zero
        position++;
        if ((long)position >= (long)len) position = 0;
    }
    const size_t nfr = total / (size_t)F;
    c->frames = (int)nfr;
    c->fplan = mxg_fft_plan_create(F, F, hopsize);   // maxiFFT::setup(fftsize, fftsize, hopsize): hop = window = fftsize
    c->iplan = mxg_ifft_plan_create(F, F, hopsize);  // maxiIFFT::setup(fftsize, fftsize, hopsize): Hann over `hopsize`
    bool ok = c->fplan && c->iplan;
    const size_t specBytes = sizeof(float) * (nfr ? nfr : 1) * bins;
    float *d_seq = nullptr;
    ok = ok && hipMalloc(&d_seq, sizeof(float) * (total ? total : 1)) == hipSuccess;
    ok = ok && hipMalloc(&c->d_impR, specBytes) == hipSuccess && hipMalloc(&c->d_impI, specBytes) == hipSuccess;
    const size_t tailBytes = sizeof(float) * (nfr > 1 ? nfr - 1 : 1) * bins;
    ok = ok && hipMalloc(&c->d_tailR, tailBytes) == hipSuccess && hipMalloc(&c->d_tailI, tailBytes) == hipSuccess;
    ok = ok && hipMalloc(&c->d_sumR, sizeof(float) * bins) == hipSuccess && hipMalloc(&c->d_sumI, sizeof(float) * bins) == hipSuccess;
    ok = ok && hipMalloc(&c->d_obuf, sizeof(float) * F) == hipSuccess;
    if (ok && nfr) {
        ok = hipMemcpy(d_seq, seq.data(), sizeof(float) * total, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && mxg_fft_batch(c->fplan, d_seq, (size_t)F, nfr, c->d_impR, c->d_impI, nullptr, nullptr, nullptr) == MXG_OK;
        ok = ok && mxg_stream_sync(nullptr) == MXG_OK;
        c->h_impR.resize(nfr * bins);
        c->h_impI.resize(nfr * bins);
        ok = ok && hipMemcpy(c->h_impR.data(), c->d_impR, specBytes, hipMemcpyDeviceToHost) == hipSuccess;
        ok = ok && hipMemcpy(c->h_impI.data(), c->d_impI, specBytes, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) {
            float maxReal = 0, maxImag = 0;  // :17-18, :25-32
            for (float v : c->h_impR)
                if (v > maxReal) maxReal = v;
            for (float v : c->h_impI)
                if (v > maxImag) maxImag = v;
            for (float &v : c->h_impR) v /= maxReal;  // :47-52 (a zero maximum divides by zero there too)
            for (float &v : c->h_impI) v /= maxImag;
            ok = hipMemcpy(c->d_impR, c->h_impR.data(), specBytes, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(c->d_impI, c->h_impI.data(), specBytes, hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    if (d_seq) (void)hipFree(d_seq);
    if (!ok || mxg_convolve_reset(c) != MXG_OK) {
        if (ok == false && !*mxg_last_error()) fail(MXG_ERR_HIP, "mxg_convolve_create: device setup failed");
        mxg_convolve_destroy(c);
        return nullptr;
    }
    return c;
}

int mxg_convolve_reset(mxg_convolve *c) {
    MXG_REQUIRE(c, "null object");
    const size_t tailBytes = sizeof(float) * (c->frames > 1 ? c->frames - 1 : 1) * c->bins;
    MXG_HIP(hipMemset(c->d_tailR, 0, tailBytes));  // FDL of blank frames, :63-68
    MXG_HIP(hipMemset(c->d_tailI, 0, tailBytes));
    MXG_HIP(hipMemset(c->d_sumR, 0, sizeof(float) * c->bins));  // :69-70
    MXG_HIP(hipMemset(c->d_sumI, 0, sizeof(float) * c->bins));
    MXG_HIP(hipMemset(c->d_obuf, 0, sizeof(float) * c->fftsize));
    return MXG_OK;
}

int mxg_convolve_frames(const mxg_convolve *c) { return c ? c->frames : 0; }

int mxg_convolve_impulse(const mxg_convolve *c, float *h_real, float *h_imag) {
    MXG_REQUIRE(c, "null object");
    if (h_real && !c->h_impR.empty()) memcpy(h_real, c->h_impR.data(), sizeof(float) * c->h_impR.size());
    if (h_imag && !c->h_impI.empty()) memcpy(h_imag, c->h_impI.data(), sizeof(float) * c->h_impI.size());
    return MXG_OK;
}

int mxg_convolve_play(mxg_convolve *c, const float *d_in, size_t nblocks, float *d_out, int mode, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(c && d_in && d_out, "null object or pointer");
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (as the reference computes) or 1 (as intended)");
    if (nblocks == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const int F = c->fftsize, bins = c->bins, nfr = c->frames;
    const size_t tail = nfr > 1 ? (size_t)nfr - 1 : 0;
    // scratch: history [tail + nblocks][bins] x2, sums [nblocks + 1][bins] x2
    float *scr = nullptr;
    const size_t hist = (tail + nblocks) * (size_t)bins, sums = (nblocks + 1) * (size_t)bins;
    if (int s = scratch_get(SCR_CONVOLVE, st, sizeof(float) * 2 * (hist + sums), (void **)&scr)) return s;
    float *HR = scr, *HI = scr + hist, *SR = scr + 2 * hist, *SI = SR + sums;
    if (tail) {
        MXG_HIP(hipMemcpyAsync(HR, c->d_tailR, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
        MXG_HIP(hipMemcpyAsync(HI, c->d_tailI, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
    }
    if (int s = mxg_fft_batch(c->fplan, d_in, (size_t)F, nblocks, HR + tail * bins, HI + tail * bins, nullptr, nullptr, st)) return s;
    MXG_HIP(hipMemcpyAsync(SR, c->d_sumR, sizeof(float) * bins, hipMemcpyDeviceToDevice, st));  // row 0: the carried sums
    MXG_HIP(hipMemcpyAsync(SI, c->d_sumI, sizeof(float) * bins, hipMemcpyDeviceToDevice, st));
    if (nfr > 0) {
        KernelTimer kt("conv_mac_kernel", st);
        const size_t n = nblocks * (size_t)bins;
        hipLaunchKernelGGL(conv_mac_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, nfr, bins, nblocks, c->d_impR,
                           c->d_impI, HR, HI, SR, SI);
        MXG_HIP(hipGetLastError());
    } else {  // an impulse shorter than one frame: the sums stay 0
        MXG_HIP(hipMemsetAsync(SR + bins, 0, sizeof(float) * nblocks * bins, st));
        MXG_HIP(hipMemsetAsync(SI + bins, 0, sizeof(float) * nblocks * bins, st));
    }
    if (int s = mxg_ifft_batch_complex(c->iplan, SR, SI, nblocks, mode == 0 ? 1 : 0, c->d_obuf, d_out, nullptr, st)) return s;
    // carry: the sums of the last frame, the newest frames-1 spectra
    MXG_HIP(hipMemcpyAsync(c->d_sumR, SR + nblocks * bins, sizeof(float) * bins, hipMemcpyDeviceToDevice, st));
    MXG_HIP(hipMemcpyAsync(c->d_sumI, SI + nblocks * bins, sizeof(float) * bins, hipMemcpyDeviceToDevice, st));
    if (tail) {
        MXG_HIP(hipMemcpyAsync(c->d_tailR, HR + nblocks * bins, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
        MXG_HIP(hipMemcpyAsync(c->d_tailI, HI + nblocks * bins, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
    }
    return MXG_OK;
}

// play() in the two halves a per-sample host needs (include/maxiConvolve.h): the output block of the NEXT fftsize samples depends
// only on the sums the previous input frame left (:76-107: maxiIFFT consumes them from the sample after that frame on), so it can
// be fetched before those samples' inputs exist; the input block then updates the delay line and the sums.
// mxg_convolve_output followed by mxg_convolve_input == mxg_convolve_play(nblocks = 1), bit for bit.
int mxg_convolve_output(mxg_convolve *c, float *d_out, int mode, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(c && d_out, "null object or pointer");
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (as the reference computes) or 1 (as intended)");
    return mxg_ifft_batch_complex(c->iplan, c->d_sumR, c->d_sumI, 1, mode == 0 ? 1 : 0, c->d_obuf, d_out, nullptr, resolve_stream(stream));
}

int mxg_convolve_input(mxg_convolve *c, const float *d_in, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(c && d_in, "null object or pointer");
    hipStream_t st = resolve_stream(stream);
    const int F = c->fftsize, bins = c->bins, nfr = c->frames;
    const size_t tail = nfr > 1 ? (size_t)nfr - 1 : 0;
    float *scr = nullptr;
    const size_t hist = (tail + 1) * (size_t)bins, sums = 2 * (size_t)bins;
    if (int s = scratch_get(SCR_CONVOLVE, st, sizeof(float) * 2 * (hist + sums), (void **)&scr)) return s;
    float *HR = scr, *HI = scr + hist, *SR = scr + 2 * hist, *SI = SR + sums;
    if (tail) {
        MXG_HIP(hipMemcpyAsync(HR, c->d_tailR, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
        MXG_HIP(hipMemcpyAsync(HI, c->d_tailI, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
    }
    if (int s = mxg_fft_batch(c->fplan, d_in, (size_t)F, 1, HR + tail * bins, HI + tail * bins, nullptr, nullptr, st)) return s;
    if (nfr > 0) {
        KernelTimer kt("conv_mac_kernel", st);
        hipLaunchKernelGGL(conv_mac_kernel, dim3((unsigned)((bins + 255) / 256)), dim3(256), 0, st, nfr, bins, (size_t)1, c->d_impR,
                           c->d_impI, HR, HI, SR, SI);
        MXG_HIP(hipGetLastError());
    } else {
        MXG_HIP(hipMemsetAsync(SR + bins, 0, sizeof(float) * bins, st));
        MXG_HIP(hipMemsetAsync(SI + bins, 0, sizeof(float) * bins, st));
    }
    MXG_HIP(hipMemcpyAsync(c->d_sumR, SR + bins, sizeof(float) * bins, hipMemcpyDeviceToDevice, st));
    MXG_HIP(hipMemcpyAsync(c->d_sumI, SI + bins, sizeof(float) * bins, hipMemcpyDeviceToDevice, st));
    if (tail) {
        MXG_HIP(hipMemcpyAsync(c->d_tailR, HR + bins, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
        MXG_HIP(hipMemcpyAsync(c->d_tailI, HI + bins, sizeof(float) * tail * bins, hipMemcpyDeviceToDevice, st));
    }
    return MXG_OK;
}

}  // extern "C"
