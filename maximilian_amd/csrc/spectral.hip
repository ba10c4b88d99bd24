// spectral.hip -- BASELINE config 4 as ONE kernel: maxiFFT(1024) -> magnitudes -> maxiMFCC, frames in, 13 doubles out.
//
// Path (reference, L/ = src/libs/): per frame fft::powerSpectrum L/fft.cpp:519-524 (calcFFT :499-505, RealFFT :228-282,
// FFT :118-211, cartToPol :507-515 magnitudes only) then maxiMFCCAnalyser::mfcc L/maxiMFCC.h:77-81
// (melFilterAndLogSq_Part2 L/maxiMFCC.cpp:48-66, dct L/maxiMFCC.h:98-111) -- the loop of
// cpp/commandline/tests/mfcctest/mfcctest.cpp:21-32 for a batch of frames.
//
// Why fuse.  As two kernels the 2 KB magnitude row of every frame is written to HBM and read back (6144 + 2152 B per
// frame against 4200 B algorithmic: 4096 in, 104 out), and the spectral kernel computes and stores all 512 magnitudes
// although the mel bank only looks at the bins below maxFreq (233 of 512 for 20 kHz at 44.1 kHz).  Here a wavefront
// transforms 8 frames one after the other exactly like K6a (three register rounds of three radix-2 stages, two padded LDS
// transposes, the reference's 10-op butterflies and replayed fp32 twiddles => real/imag/magnitudes bit-exact), parks the
// magnitudes it needs in a private LDS tile M[8][bins], and then runs the mel/DCT stage for those 8 frames with every
// lane busy:
//   mel walk   lane = (frame j = lane/8, slot s = lane%8).  The host packs the filters into 8 lists of about equal total
//              support (LPT); a slot walks its list one bin per step, so each filter's band sum is the reference's
//              sequential sum over its support in increasing bin order (terms outside the support are exact +0.0 in the
//              reference: bit-identical, see mfcc.hip) -- 8 frames x 8 filters in flight per wave instead of one lane per
//              frame (which would need a 64-frame x 233-bin tile = 60 KB of LDS per wave);
//   log pass   the 8 x numFilters raw band sums, one per lane, log(mb*mb) (device log: the tolerance of method 0);
//   DCT        lane = (frame, coefficient): the 42-term sum in the reference's j order, / numCoeffs, stored coalesced.
// HBM traffic per frame = the algorithmic 4200 B (+2048 B when the magnitudes are also requested).
//
// sqrtf.  hipcc's correctly-rounded sqrtf expands to ~25 instructions (v_sqrt_f32, the +-1 ulp residual test, and a
// 2^32 pre-scale for inputs below 2^-96 whose v_sqrt_f32 result would be denormal-inaccurate).  exact_sqrtf() keeps the
// residual test and moves the rare small-input case into a branch that no lane of a wavefront normally takes.
#include "mxg_spectral.h"

namespace mxg {
namespace {

constexpr int kGroup = 8;  // frames per mel phase = 64 lanes / kFusedSlots

// correctly rounded sqrt of a finite non-negative float (0, inf and NaN pass through v_sqrt_f32 unchanged)
__device__ __forceinline__ float exact_sqrtf(float x) {
    if (__builtin_expect(x < 0x1p-96f && x > 0.0f, 0)) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);  // <= 1 ulp
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x);  // x - sd*s, one rounding
    const float ru = __builtin_fmaf(-su, s, x);
    float r = rd <= 0.0f ? sd : s;
    r = ru > 0.0f ? su : r;
    return r;
}

// the `a` half of post_pair (L/fft.cpp:250-262): bin i of the real transform from X[i] and X[half - i]
__device__ __forceinline__ float2 post_lo(const float2 a, const float2 b, const float2 w) {
    const float h1r = 0.5f * (a.x + b.x);
    const float h1i = 0.5f * (a.y - b.y);
    const float h2r = 0.5f * (a.y + b.y);
    const float h2i = -0.5f * (a.x - b.x);
    float2 r;
    r.x = h1r + w.x * h2r - w.y * h2i;
    r.y = h1i + w.x * h2i + w.y * h2r;
    return r;
}

struct FusedArgs {
    const float *signal;
    size_t frame_stride, nframes;
    const float *window;
    const float2 *tw, *post;
    unsigned numFilters, numCoeffs, nbUsed, mstride, nfp;
    int steps;
    const double *fsW;
    const int *fsMeta;
    const double *dct;
    float *mags;
    double *melraw, *melbands, *mfcc;
};

// FULL: magnitudes of all 512 bins are needed (written out, or the bank reaches beyond bin 256)
template <bool FULL, bool WRITE_MAGS, bool ALIGNED8>
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void fft_mfcc_kernel(const FusedArgs A) {
    extern __shared__ double s_dyn[];
    // [tw 512 float2][fsW steps*8 f64][dct NF*NC f64][fsMeta steps*8 i32] | per wave: X, mel, M
    float2 *s_tw = reinterpret_cast<float2 *>(s_dyn);
    double *s_w = reinterpret_cast<double *>(s_tw + 512);
    double *s_d = s_w + (size_t)A.steps * kFusedSlots;
    int *s_meta = reinterpret_cast<int *>(s_d + (size_t)A.numFilters * A.numCoeffs);
    const size_t metaInts = ((size_t)A.steps * kFusedSlots + 3) & ~(size_t)3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t perWaveBytes = sizeof(float2) * kX1024 + sizeof(float) * kGroup * A.mstride + sizeof(double) * kGroup * A.nfp;
    char *wbase = reinterpret_cast<char *>(s_meta + metaInts) + (size_t)wave * perWaveBytes;
    float2 *X = reinterpret_cast<float2 *>(wbase);
    double *s_mel = reinterpret_cast<double *>(wbase + sizeof(float2) * kX1024);
    float *M = reinterpret_cast<float *>(wbase + sizeof(float2) * kX1024 + sizeof(double) * kGroup * A.nfp);
    for (int i = threadIdx.x; i < 511; i += blockDim.x) s_tw[i] = A.tw[i];
    for (int i = threadIdx.x; i < A.steps * kFusedSlots; i += blockDim.x) {
        s_w[i] = A.fsW[i];
        s_meta[i] = A.fsMeta[i];
    }
    for (unsigned i = threadIdx.x; i < A.numFilters * A.numCoeffs; i += blockDim.x) s_d[i] = A.dct[i];
    for (unsigned i = lane; i < kGroup * A.nfp; i += 64) s_mel[i] = 0.0;  // filters with an empty support stay 0
    __syncthreads();

    const int lo = lane & 7, hi = lane >> 3;
    const int rev6 = (int)(__brev((unsigned)lane) >> 26);
    float2 wv[8];
    unsigned li[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int rev3 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);
        li[e] = 2u * (unsigned)(rev3 * 64 + rev6);
        wv[e] = make_float2(A.window[li[e]], A.window[li[e] + 1]);
    }
    asm volatile("" : "+v"(wv[0].x), "+v"(wv[0].y), "+v"(wv[1].x), "+v"(wv[1].y), "+v"(wv[2].x), "+v"(wv[2].y),
                 "+v"(wv[3].x), "+v"(wv[3].y));
    asm volatile("" : "+v"(wv[4].x), "+v"(wv[4].y), "+v"(wv[5].x), "+v"(wv[5].y), "+v"(wv[6].x), "+v"(wv[6].y),
                 "+v"(wv[7].x), "+v"(wv[7].y));
    // post-pass: twiddles of this lane's four pairs in registers; LDS slots of the pairs (i, 512 - i), i = 1 + lane + 64q,
    // are one base each plus a compile-time offset: pad8(i0 + 64q) = pad8(i0) + 72q
    float2 pw[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pw[q] = A.post[(1 + lane + 64 * q) < 256 ? 1 + lane + 64 * q : 255];
    asm volatile("" : "+v"(pw[0].x), "+v"(pw[0].y), "+v"(pw[1].x), "+v"(pw[1].y), "+v"(pw[2].x), "+v"(pw[2].y),
                 "+v"(pw[3].x), "+v"(pw[3].y));
    const int pa0 = pad8(1 + lane), pb0 = pad8(511 - lane);
    const int zidx = lane == 0 ? 0 : 256;  // lanes 0 / 63 also own bin 0 / the middle bin
    const size_t nframes = A.nframes;
    auto load_frame = [&](size_t fr, float2 (&dst)[8]) {
        const unsigned fu = __builtin_amdgcn_readfirstlane((unsigned)(fr < nframes ? fr : nframes - 1));
        const float *x = A.signal + (size_t)fu * A.frame_stride;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if constexpr (ALIGNED8) {
                dst[e] = *reinterpret_cast<const float2 *>(x + li[e]);
            } else {
                dst[e].x = x[li[e]];
                dst[e].y = x[li[e] + 1];
            }
        }
    };
    const size_t ngroups = (nframes + kGroup - 1) / kGroup;
    const size_t gstep = (size_t)gridDim.x * kWavesPerBlock;
    const size_t g0 = (size_t)blockIdx.x * kWavesPerBlock + wave;
    float2 nxt[8];
    load_frame(g0 * kGroup, nxt);
    const int mj = lane >> 3, ms = lane & 7;  // mel walk: frame of the group, slot
    for (size_t g = g0; g < ngroups; g += gstep) {
        const size_t f0 = g * kGroup;
#pragma unroll 1
        for (int j = 0; j < kGroup; j++) {
            float2 v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[e].x = nxt[e].x * wv[e].x;  // calcFFT L/fft.cpp:501-503
                v[e].y = nxt[e].y * wv[e].y;
            }
            load_frame(j + 1 < kGroup ? f0 + j + 1 : (g + gstep) * kGroup, nxt);
            const float2 a0 = s_tw[0];
            const float2 a1[2] = {s_tw[1], s_tw[2]};
            const float2 a2[4] = {s_tw[3], s_tw[4], s_tw[5], s_tw[6]};
            const float2 b0 = s_tw[7 + lo];
            const float2 b1[2] = {s_tw[15 + lo], s_tw[15 + 8 + lo]};
            const float2 b2[4] = {s_tw[31 + lo], s_tw[31 + 8 + lo], s_tw[31 + 16 + lo], s_tw[31 + 24 + lo]};
            const float2 c0 = s_tw[63 + lane];
            const float2 c1[2] = {s_tw[127 + lane], s_tw[127 + 64 + lane]};
            const float2 c2[4] = {s_tw[255 + lane], s_tw[255 + 64 + lane], s_tw[255 + 128 + lane], s_tw[255 + 192 + lane]};
            round3(v, a0, a1, a2);
#pragma unroll
            for (int e = 0; e < 8; e++) X[pad8(8 * lane + e)] = v[e];
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = X[pad8(hi * 64 + e * 8 + lo)];
            round3(v, b0, b1, b2);
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) X[pad8(hi * 64 + e * 8 + lo)] = v[e];
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = X[pad8(e * 64 + lane)];
            round3(v, c0, c1, c2);
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) X[pad8(e * 64 + lane)] = v[e];
            wave_lds_sync();
            // real split post-pass (L/fft.cpp:245-275) + magnitudes (cartToPol :510-511) into the tile row of frame j
            float *Mrow = M + j * A.mstride;
            const bool frame_live = f0 + (size_t)j < nframes;  // wave-uniform
            float *grow = nullptr;
            if constexpr (WRITE_MAGS)
                grow = A.mags + (size_t)__builtin_amdgcn_readfirstlane((unsigned)(f0 + j < nframes ? f0 + j : nframes - 1)) * 512;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float2 xa = X[pa0 + 72 * q], xb = X[pb0 - 72 * q];
                const bool own = q < 3 || lane < 63;  // lane 63's fourth pair would be bin 256: handled below
                if constexpr (FULL) {
                    float2 a = xa, b = xb;
                    post_pair(a, b, pw[q]);
                    const float ma = exact_sqrtf(a.x * a.x + a.y * a.y), mb = exact_sqrtf(b.x * b.x + b.y * b.y);
                    if (own) {
                        Mrow[1 + lane + 64 * q] = ma;
                        Mrow[511 - lane - 64 * q] = mb;
                        if constexpr (WRITE_MAGS)
                            if (frame_live) {
                                grow[1 + lane + 64 * q] = ma;
                                grow[511 - lane - 64 * q] = mb;
                            }
                    }
                } else {
                    const float2 a = post_lo(xa, xb, pw[q]);
                    const float ma = exact_sqrtf(a.x * a.x + a.y * a.y);  // L/fft.cpp:510-511
                    if (own) Mrow[1 + lane + 64 * q] = ma;
                }
            }
            {   // bin 0 packs DC and Nyquist (L/fft.cpp:274-275); bin 256 passes through untouched
                const float2 z = X[pad8(zidx)];
                const float zr = lane == 0 ? z.x + z.y : z.x, zi = lane == 0 ? z.x - z.y : z.y;
                const float mz = exact_sqrtf(zr * zr + zi * zi);
                if (lane == 0 || lane == 63) {
                    Mrow[zidx] = mz;
                    if constexpr (WRITE_MAGS)
                        if (frame_live) grow[zidx] = mz;
                }
            }
            wave_lds_sync();
        }
        // ---- mel walk: lane (mj, ms) walks slot ms's filter list over frame mj's magnitudes -----------------
        {
            const float *Mrow = M + mj * A.mstride;
            double *melrow = s_mel + mj * A.nfp;
            double acc = 0.0;  // L/maxiMFCC.cpp:52
            for (int t = 0; t < A.steps; t++) {
                const int meta = s_meta[t * kFusedSlots + ms];
                const double w = s_w[t * kFusedSlots + ms];
                const double x = (double)Mrow[meta & 0xffff];
                acc += (w * x);  // L/maxiMFCC.cpp:57
                const int fid = meta >> 16;
                if (fid) {
                    melrow[fid - 1] = acc;
                    acc = 0.0;
                }
            }
        }
        wave_lds_sync();
        // ---- log-square (L/maxiMFCC.cpp:63), one band per lane ---------------------------------------------
        for (unsigned idx = lane; idx < kGroup * A.numFilters; idx += 64) {
            const unsigned jj = idx / A.numFilters, ff = idx - jj * A.numFilters;
            const double raw = s_mel[jj * A.nfp + ff];
            const double lv = log_square(raw);
            s_mel[jj * A.nfp + ff] = lv;
            if (f0 + jj < nframes) {
                if (A.melraw) A.melraw[(f0 + jj) * A.numFilters + ff] = raw;
                if (A.melbands) A.melbands[(f0 + jj) * A.numFilters + ff] = lv;
            }
        }
        wave_lds_sync();
        // ---- DCT (L/maxiMFCC.h:98-111): lane = (frame, coefficient), j ascending ---------------------------
        for (unsigned p = lane; p < kGroup * A.numCoeffs; p += 64) {
            const unsigned jj = p / A.numCoeffs, i = p - jj * A.numCoeffs;
            const double *mrow = s_mel + jj * A.nfp;
            double c = 0.0;
            for (unsigned jf = 0; jf < A.numFilters; jf++) c += (s_d[jf * A.numCoeffs + i] * mrow[jf]);
            if (f0 + jj < nframes) A.mfcc[(f0 + jj) * A.numCoeffs + i] = c / (double)A.numCoeffs;
        }
        wave_lds_sync();
        // the raw sums of the next group land on the same cells; cells of empty filters hold log_square(0) = 0
    }
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" int mxg_fft_mfcc_batch(const mxg_fft_plan *fp, const mxg_mfcc_plan *mp, const float *d_signal,
                                  size_t frame_stride, size_t nframes, float *d_mags, double *d_melraw,
                                  double *d_melbands, double *d_mfcc, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(fp && mp && d_signal && d_mfcc, "null plan / signal / mfcc");
    MXG_REQUIRE(fp->fftSize == 1024, "the fused kernel is specialised for fftSize 1024 (use mxg_fft_batch + mxg_mfcc_batch)");
    MXG_REQUIRE(mp->numBins == 512, "the mfcc plan must be set up for the 512 bins of a 1024-point maxiFFT");
    MXG_REQUIRE(mp->fsSteps > 0 && mp->d_fsW, "this filter bank has no fused schedule (numFilters > 64?): use the two-kernel path");
    MXG_REQUIRE(nframes < ((size_t)1 << 32), "nframes must be < 2^32");
    if (nframes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    FusedArgs A;
    A.signal = d_signal; A.frame_stride = frame_stride; A.nframes = nframes;
    A.window = fp->d_window; A.tw = fp->d_tw; A.post = fp->d_post;
    A.numFilters = mp->numFilters; A.numCoeffs = mp->numCoeffs; A.nbUsed = mp->nbUsed;
    MXG_REQUIRE(mp->nbUsed <= 257, "mel bank reaches beyond bin 256");  // binFreq = sr/numBins*bin never does
    const bool full = d_mags != nullptr;
    A.mstride = full ? 520 : 264;                 // the post-pass writes bins 0..256 (0..511 with magnitudes out) + pad, = 8 mod 32
    A.nfp = mp->numFilters | 1u;                  // odd row stride for the band rows
    if (A.nfp == mp->numFilters) A.nfp += 2;
    A.steps = mp->fsSteps; A.fsW = mp->d_fsW; A.fsMeta = mp->d_fsMeta; A.dct = mp->d_dct;
    A.mags = d_mags; A.melraw = d_melraw; A.melbands = d_melbands; A.mfcc = d_mfcc;
    const size_t metaInts = ((size_t)A.steps * kFusedSlots + 3) & ~(size_t)3;
    const size_t perWave = sizeof(float2) * kX1024 + sizeof(float) * kGroup * A.mstride + sizeof(double) * kGroup * A.nfp;
    const size_t lds = sizeof(float2) * 512 + sizeof(double) * ((size_t)A.steps * kFusedSlots + (size_t)A.numFilters * A.numCoeffs) +
                       sizeof(int) * metaInts + kWavesPerBlock * perWave;
    MXG_REQUIRE(lds <= 160 * 1024, "filter bank too large for the fused kernel's LDS layout");
    const size_t ngroups = (nframes + kGroup - 1) / kGroup;
    size_t blocks = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    const size_t cap = 256 * 2;  // persistent: two workgroups per CU, grid-stride over groups of 8 frames
    if (blocks > cap) blocks = cap;
    const bool aligned8 = (((uintptr_t)d_signal) & 7) == 0 && (frame_stride & 1) == 0;
    typedef void (*kern_t)(const FusedArgs);
    kern_t k;
    if (d_mags)
        k = aligned8 ? fft_mfcc_kernel<true, true, true> : fft_mfcc_kernel<true, true, false>;
    else if (full)
        k = aligned8 ? fft_mfcc_kernel<true, false, true> : fft_mfcc_kernel<true, false, false>;
    else
        k = aligned8 ? fft_mfcc_kernel<false, false, true> : fft_mfcc_kernel<false, false, false>;
    if (lds > 64 * 1024) MXG_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer kt("fft_mfcc_kernel", st);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, st, A);
    return check_hip(hipGetLastError(), "fft_mfcc_kernel launch");
}
