// mxg_spectral.h -- plans and device helpers shared by the spectral kernels: fft.hip (maxiFFT / maxiIFFT batches),
// mfcc.hip (maxiMFCC batch) and spectral.hip (the fused FFT -> magnitudes -> mel -> DCT kernel of config 4).
#pragma once
#include <math.h>

#include <vector>

#include "mxg_common.h"

struct mxg_fft_plan {
    int fftSize, hopSize, windowSize, bins, half, numBits;
    float *d_window;   // [fftSize]
    float2 *d_tw;      // stage twiddles: entry (h-1)+n = (ar0, ai0) of step n in a stage with BlockEnd h
    float2 *d_post;    // post-pass (wr, wi) for i = 1 .. half/2-1 at index i
};

// maxiIFFT::setup (L/maxiFFT.cpp:140-153): windowSize ? windowSize : fftSize, Hann over that, zero beyond
struct mxg_ifft_plan {
    int fftSize, hopSize, windowSize, bins, numBits;  // numBits = log2(fftSize): the inverse is a FULL-size complex FFT
    float *d_window;  // [fftSize]
    float2 *d_tw;     // inverse-direction stage twiddles, same indexing as mxg_fft_plan::d_tw
};

struct mxg_mfcc_plan {
    unsigned numBins, numFilters, numCoeffs, nbUsed;  // nbUsed = 1 + last bin with a non-zero weight
    std::vector<double> h_W;    // [filter + bin*numFilters]  (reference layout)
    std::vector<double> h_dct;  // [i + j*numCoeffs]
    int slots;                  // S: max number of simultaneously open filters (0 = stream kernel n/a)
    double *d_schedW;           // [nbUsed][S] weight of the filter occupying slot s at this bin (or 0)
    int *d_schedFin;            // [nbUsed][S] filter closing in slot s after this bin, or -1
    int *d_lo, *d_hi, *d_off;   // per filter: support [lo, hi] (hi < lo = empty), offset into d_Wc
    double *d_Wc;               // compacted weights, filter-major, increasing bin
    double *d_dct;              // [j*numCoeffs + i]
    double *d_Wpad;             // dense [kPad][nfPad] row-major by bin, for the MFMA path
    unsigned nfPad, kPad;
    // Slot schedule of the fused FFT+MFCC kernel (spectral.hip): the filters are packed into kFusedSlots lists of about
    // equal total support length; list s is walked one bin per step, so step t of slot s is (bin, weight, filter that
    // ends here or -1).  fsSteps = the longest list (shorter ones are padded with weight-0 steps AFTER their last
    // filter).  fsSteps == 0: the fused kernel is not applicable to this bank.
    int fsSteps;
    double *d_fsW;              // [fsSteps][kFusedSlots]
    int *d_fsMeta;              // [fsSteps][kFusedSlots]: bin | (filter + 1) << 16 on the last bin of a filter
};
constexpr int kFusedSlots = 8;

namespace mxg {
namespace {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ void wave_lds_sync() {
    // LDS traffic of ONE wavefront: the DS unit executes a wave's instructions in order, so only
    // the compiler has to be told not to move accesses across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One butterfly, op for op L/fft.cpp:184-192 (j = upper, k = lower input).
__device__ __forceinline__ void bfly(float2 &xj, float2 &xk, const float2 w) {
    float tr = w.x * xk.x - w.y * xk.y;
    float ti = w.x * xk.y + w.y * xk.x;
    xk.x = xj.x - tr;
    xk.y = xj.y - ti;
    xj.x += tr;
    xj.y += ti;
}

// Real split post-pass for the pair (i, i3 = half - i), L/fft.cpp:250-268.
__device__ __forceinline__ void post_pair(float2 &a, float2 &b, const float2 w) {
    const float wr = w.x, wi = w.y;
    float h1r = 0.5f * (a.x + b.x);
    float h1i = 0.5f * (a.y - b.y);
    float h2r = 0.5f * (a.y + b.y);
    float h2i = -0.5f * (a.x - b.x);
    a.x = h1r + wr * h2r - wi * h2i;
    a.y = h1i + wr * h2i + wi * h2r;
    b.x = h1r - wr * h2r + wi * h2i;
    b.y = -h1i + wr * h2i + wi * h2r;
}

// ---- K6a: fftSize 1024 (half = 512 = 8^3) ------------------------------------------------------
// LDS image of one frame: 512 float2 + 1 pad per 8 (index p = i + i/8): conflict-free for the
// stride-8 and stride-64 lane patterns of the two transposes (bank maths in DESIGN.md).
constexpr int kX1024 = 512 + 64;
#ifndef MXG_FFT_MINWAVES
#define MXG_FFT_MINWAVES 3
#endif

__device__ __forceinline__ int pad8(int i) { return i + (i >> 3); }

// three in-register radix-2 stages over the 8 points of a lane; w0: 1 twiddle (pairs e,e+1),
// w1[2]: pairs (e,e+2) with n-offset e&1, w2[4]: pairs (e,e+4) with n-offset e&3.
__device__ __forceinline__ void round3(float2 (&x)[8], const float2 w0, const float2 (&w1)[2],
                                       const float2 (&w2)[4]) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) bfly(x[e], x[e + 1], w0);
#pragma unroll
    for (int e = 0; e < 8; e++)
        if ((e & 2) == 0) bfly(x[e], x[e + 2], w1[e & 1]);
#pragma unroll
    for (int e = 0; e < 4; e++) bfly(x[e], x[e + 4], w2[e]);
}

// log-square of L/maxiMFCC.cpp:63
__device__ __forceinline__ double log_square(double mb) { return mb > 0.000001 ? log(mb * mb) : 0.0; }

}  // namespace
}  // namespace mxg
