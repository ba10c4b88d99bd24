// fft.hip -- maxiFFT batch on gfx950: windowed real FFT + polar conversion, one wavefront per frame.
//
// Path (reference, L/ = src/libs/): maxiFFT::setup/process L/maxiFFT.cpp:45-91 ->
// fft::powerSpectrum L/fft.cpp:519-524 -> calcFFT :499-505 (window multiply) -> RealFFT
// :228-282 (even/odd pack, half-size complex FFT :118-211, real split post-pass) -> cartToPol
// :507-515.  The reference runs this once per hop inside the per-sample process(); here a batch
// of frames (frame k = signal[k*frame_stride .. +fftSize)) is transformed per launch.
//
// Numerics: everything is fp32, and BIT-EXACT for real/imag/magnitudes.  The reference builds
// its twiddles by an fp32 recurrence seeded from double sin/cos (L/fft.cpp:161-182, :245-272);
// those sequences do not depend on the data, so the plan replays the very same recurrences on
// the host (host libm, fp32 ops, no contraction) and uploads them as tables.  Butterflies inside
// one stage are independent, so any parallel order gives identical bits as long as each
// butterfly uses the reference's op sequence (tr = ar*xr - ai*xi as mul,mul,sub ...), which
// -ffp-contract=off guarantees.  sqrtf is correctly rounded on gfx950 (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); atan2f is OCML's and carries a stated tolerance.
//
// Kernels.  K6a `fft1024_kernel` (fftSize 1024, the size of every config/test in the reference):
// the 512-point complex FFT is done as three register rounds of three radix-2 stages each, 8
// complex points per lane, with two padded LDS transposes between rounds; loads are direct
// global float2 reads in bit-reversed order (each wavefront instruction still covers one full
// 512-B segment).  K6b `fft_generic_kernel`: any power-of-two size 8..8192, one LDS pass per
// stage.  HBM bytes per frame: fftSize*4 read + bins*4 per requested output.
#include <math.h>

#include <vector>

#include "mxg_common.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "mxg_spectral.h"

namespace mxg {
namespace {

struct FftOut {
    float *real, *imag, *mags, *phases;
};

// compile-time output selection (bit 0 real, 1 imag, 2 mags, 3 phases): a statically known number
// of stores per frame lets the compiler use counted vmcnt waits for the prefetched next frame.
template <int OMASK>
__device__ __forceinline__ void emit_bin_t(const FftOut &o, size_t base, int bin, float2 v) {
    if constexpr (OMASK & 1) o.real[base + bin] = v.x;
    if constexpr (OMASK & 2) o.imag[base + bin] = v.y;
    if constexpr (OMASK & 12) {
        float power = v.x * v.x + v.y * v.y;  // L/fft.cpp:510
        if constexpr (OMASK & 4) o.mags[base + bin] = exact_sqrtf(power);
        if constexpr (OMASK & 8) o.phases[base + bin] = atan2f(v.y, v.x);
    }
}

// K6a form: `row` holds the four output pointers already advanced to this frame (wave-uniform, SGPRs) and the
// bin is an unsigned 32-bit offset, so the stores need no per-lane 64-bit address arithmetic.
template <int OMASK>
__device__ __forceinline__ void emit_row_t(const FftOut &row, unsigned bin, float2 v) {
    if constexpr (OMASK & 1) row.real[bin] = v.x;
    if constexpr (OMASK & 2) row.imag[bin] = v.y;
    if constexpr (OMASK & 12) {
        float power = v.x * v.x + v.y * v.y;  // L/fft.cpp:510
        if constexpr (OMASK & 4) row.mags[bin] = exact_sqrtf(power);
        if constexpr (OMASK & 8) row.phases[bin] = atan2f(v.y, v.x);
    }
}

__device__ __forceinline__ void emit_bin(const FftOut &o, size_t base, int bin, float2 v) {
    if (o.real) o.real[base + bin] = v.x;
    if (o.imag) o.imag[base + bin] = v.y;
    if (o.mags || o.phases) {
        float power = v.x * v.x + v.y * v.y;  // L/fft.cpp:510
        if (o.mags) o.mags[base + bin] = exact_sqrtf(power);
        if (o.phases) o.phases[base + bin] = atan2f(v.y, v.x);
    }
}

// Stages st0 .. numBits-1 of the radix-2 transform of `npts` complex points held in LDS (X, padded 1 per 32), for one wavefront:
// up to three consecutive stages per LDS pass on eight values in registers -- the same butterflies on the same operands as one
// stage per pass (a butterfly of stage s pairs values 2^s apart and takes twiddle tw[2^s - 1 + position inside the half-block]).
__device__ __forceinline__ void lds_stages(float2 *X, const int npts, const int numBits, const float2 *tw, int st, const int lane) {
    auto P = [](int idx) { return idx + (idx >> 5); };
    for (; st + 2 < numBits; st += 3) {
        const int h = 1 << st;
        for (int q = lane; q < (npts >> 3); q += 64) {
            const int nn = q & (h - 1);
            const int j = ((q >> st) << (st + 3)) | nn;
            float2 x[8];
#pragma unroll
            for (int mm = 0; mm < 8; mm++) x[mm] = X[P(j + mm * h)];
            const float2 w0 = tw[h - 1 + nn];
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) bfly(x[mm], x[mm + 1], w0);
            const float2 w1a = tw[2 * h - 1 + nn], w1b = tw[2 * h - 1 + nn + h];
            bfly(x[0], x[2], w1a);
            bfly(x[1], x[3], w1b);
            bfly(x[4], x[6], w1a);
            bfly(x[5], x[7], w1b);
#pragma unroll
            for (int mm = 0; mm < 4; mm++) bfly(x[mm], x[mm + 4], tw[4 * h - 1 + nn + mm * h]);
#pragma unroll
            for (int mm = 0; mm < 8; mm++) X[P(j + mm * h)] = x[mm];
        }
        wave_lds_sync();
    }
    for (; st + 1 < numBits; st += 2) {
        const int h = 1 << st;
        for (int q = lane; q < (npts >> 2); q += 64) {
            const int nn = q & (h - 1);
            const int j = ((q >> st) << (st + 2)) | nn;
            float2 x0 = X[P(j)], x1 = X[P(j + h)], x2 = X[P(j + 2 * h)], x3 = X[P(j + 3 * h)];
            const float2 w = tw[h - 1 + nn];
            bfly(x0, x1, w);
            bfly(x2, x3, w);
            bfly(x0, x2, tw[2 * h - 1 + nn]);
            bfly(x1, x3, tw[2 * h - 1 + nn + h]);
            X[P(j)] = x0;
            X[P(j + h)] = x1;
            X[P(j + 2 * h)] = x2;
            X[P(j + 3 * h)] = x3;
        }
        wave_lds_sync();
    }
    for (; st < numBits; st++) {
        const int h = 1 << st;
        for (int b = lane; b < (npts >> 1); b += 64) {
            const int nn = b & (h - 1);
            const int j = ((b >> st) << (st + 1)) | nn;
            float2 xj = X[P(j)], xk = X[P(j + h)];
            bfly(xj, xk, tw[h - 1 + nn]);
            X[P(j)] = xj;
            X[P(j + h)] = xk;
        }
        wave_lds_sync();
    }
}

// ---- K6b: generic size -----------------------------------------------------------------------

__global__ __launch_bounds__(64 * kWavesPerBlock) void fft_generic_kernel(
    const float *__restrict__ signal, size_t frame_stride, size_t nframes, int fftSize, int numBits,
    const float *__restrict__ window, const float2 *__restrict__ tw, const float2 *__restrict__ post,
    FftOut out) {
    extern __shared__ float2 s_dyn[];
    const int half = fftSize >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    // [stage twiddles half - 1 (+1)] | per wave: X
    float2 *s_tw = s_dyn;
    float2 *X = s_dyn + half + (size_t)wave * (half + (half >> 5) + 1);
    for (int i = threadIdx.x; i < half - 1; i += blockDim.x) s_tw[i] = tw[i];
    __syncthreads();
    auto P = [](int idx) { return idx + (idx >> 5); };  // one pad slot per 32: breaks pow-2 strides
    for (size_t f = (size_t)blockIdx.x * nwaves + wave; f < nframes; f += (size_t)gridDim.x * nwaves) {
        const float *x = signal + f * frame_stride;
        for (int i = lane; i < half; i += 64) {
            float2 v;
            v.x = x[2 * i] * window[2 * i];          // calcFFT L/fft.cpp:501-503
            v.y = x[2 * i + 1] * window[2 * i + 1];  // RealFFT pack :238-241
            int j = (int)(__brev((unsigned)i) >> (32 - numBits));  // :146-150
            X[P(j)] = v;
        }
        wave_lds_sync();
        lds_stages(X, half, numBits, s_tw, 0, lane);
        const size_t base = f * (size_t)half;
        for (int i = 1 + lane; i < (half >> 1); i += 64) {
            float2 a = X[P(i)], b = X[P(half - i)];
            post_pair(a, b, post[i]);
            emit_bin(out, base, i, a);
            emit_bin(out, base, half - i, b);
        }
        if (lane == 0) {  // L/fft.cpp:274-275 and the untouched middle bin
            float2 z = X[P(0)];
            float2 z0 = {z.x + z.y, z.x - z.y};
            emit_bin(out, base, 0, z0);
            if (half >= 2) emit_bin(out, base, half >> 1, X[P(half >> 1)]);
        }
        wave_lds_sync();
    }
}

// ---- K6i: maxiIFFT (L/maxiFFT.cpp:155-192 SPECTRUM mode; L/fft.cpp:590-611) -----------------------------
// polToCart (float cos/sin of the phase: device cosf/sinf => stated tolerance), zeroed negative
// frequencies, a full n-point complex inverse FFT with the reference's fp32 recurrence twiddles
// (replayed on the host for the inverse angle), /n, times the window.  One wavefront per frame, one
// LDS pass per stage (the structure of K6b).  ifft_out[f][i] = 0.0f + out_real[i]*window[i] is the
// zero-filled `ifftOut` of the reference after calcIFFT's `+=`.
// INPUT 0: SPECTRUM mode (magnitudes, phases -> polToCart).  1: cartesian inputs (mags = real, phases = imag: what
// inverseFFTComplex was meant to transform).  2: the transform inputs are all zero (what the reference's COMPLEX mode
// actually transforms on a fresh object, L/fft.cpp:613-619 -- see mxg_ifft_batch_complex).
// One frame: polToCart into bit-reversed order, the inverse transform in LDS (`X`, padded 1 per 32), for one wavefront.
template <int INPUT>
__device__ __forceinline__ void ifft_frame(const float *__restrict__ m, const float *__restrict__ ph, const int n, const int numBits,
                                           const float2 *s_tw, float2 *X, const int lane) {
    const int half = n >> 1;
    auto P = [](int idx) { return idx + (idx >> 5); };
    // polToCart (L/fft.cpp:590-604) into bit-reversed order.  Bins half .. n-1 are zero and land on the ODD slots, so the first
    // stage -- x_j +- (1, 0) * 0 -- only copies x_j to its neighbour (it can differ from the reference in the sign of a zero,
    // which no later sum, product or the final `0 + r * window` can turn into anything else): both slots are written here.
    for (int i = lane; i < half; i += 64) {
        float2 v = {0.0f, 0.0f};
        if constexpr (INPUT != 2) {
            const float mg = m[i], p = ph[i];
            if constexpr (INPUT == 0) {
                float sn, cs;
                sincosf(p, &sn, &cs);
                v.x = mg * cs;  // :597-598
                v.y = mg * sn;
            } else {
                v.x = mg;
                v.y = p;
            }
        }
        const int j = (int)(__brev((unsigned)i) >> (32 - numBits));  // even: i < half has its top bit clear
        X[P(j)] = v;
        X[P(j + 1)] = v;
    }
    wave_lds_sync();
    lds_stages(X, n, numBits, s_tw, 1, lane);
}

template <int INPUT>
__global__ void ifft_generic_kernel(const float *__restrict__ mags, const float *__restrict__ phases,
                                    size_t nframes, int n, int numBits, const float *__restrict__ window,
                                    const float2 *__restrict__ tw, float *__restrict__ ifft_out) {
    extern __shared__ float2 s_dyn[];
    const int half = n >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    // [twiddles n - 1 (+1)] | per wave: X, padded 1 per 32
    float2 *s_tw = s_dyn;
    float2 *X = s_dyn + n + (size_t)wave * (n + (n >> 5) + 1);
    for (int i = threadIdx.x; i < n - 1; i += blockDim.x) s_tw[i] = tw[i];
    __syncthreads();
    auto P = [](int idx) { return idx + (idx >> 5); };
    const float inv = 1.0f / (float)n;  // n is a power of two: x * inv is x / n, correctly rounded either way (L/fft.cpp:201-209 divides)
    for (size_t f = (size_t)blockIdx.x * nwaves + wave; f < nframes; f += (size_t)gridDim.x * nwaves) {
        ifft_frame<INPUT>(mags + f * (size_t)half, phases + f * (size_t)half, n, numBits, s_tw, X, lane);
        float *o = ifft_out + f * (size_t)n;
        for (int i = lane; i < n; i += 64) {
            const float r = X[P(i)].x * inv;  // L/fft.cpp:201-209
            o[i] = 0.0f + r * window[i];        // :608-610 into the zero-filled ifftOut
        }
        wave_lds_sync();
    }
}

// K6s: the same transform with maxiIFFT::process's hop buffer (L/maxiFFT.cpp:176-183) carried in LDS, so `ifftOut` never goes to
// HBM (4 KB written and 4 KB read back per 1024-point frame otherwise).  A wavefront owns `chunk` consecutive frames and streams
// through them as the reference does: shift the buffer left by hop, zero the tail, add the frame (two LDS images, so the shift is
// a read of one and a write of the other), hand out the first hop samples.  The buffer a chunk starts from is that of the frame
// before it, which only the R = ceil(n / hop) - 1 frames before THAT one still reach: the wavefront first runs those R frames from
// an empty buffer without output (everything older has been shifted out, so the additions that remain are the reference's, in
// its order); chunk 0 starts from the carried buffer instead.  The wavefront of the last frame leaves the buffer behind.
template <int INPUT>
__global__ void ifft_stream_kernel(const float *__restrict__ mags, const float *__restrict__ phases, size_t nframes, int n,
                                   int numBits, int hop, int chunk, const float *__restrict__ window,
                                   const float2 *__restrict__ tw, const float *__restrict__ buf_in, float *__restrict__ out,
                                   float *__restrict__ buf_out, float *__restrict__ ifft_out) {
    extern __shared__ float2 s_dyn[];
    const int half = n >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    // [twiddles n] | per wave: X (n + n/32 + 1 float2) | two hop-buffer images (2 n floats = n float2)
    const size_t per_wave = (size_t)(n + (n >> 5) + 1) + (size_t)n;
    float2 *s_tw = s_dyn;
    float2 *X = s_dyn + n + (size_t)wave * per_wave;
    float *B0 = reinterpret_cast<float *>(X + (n + (n >> 5) + 1)), *B1 = B0 + n;
    for (int i = threadIdx.x; i < n - 1; i += blockDim.x) s_tw[i] = tw[i];
    __syncthreads();
    auto P = [](int idx) { return idx + (idx >> 5); };
    const float inv = 1.0f / (float)n;
    const long long R = (long long)((n + hop - 1) / hop) - 1;
    const size_t nchunks = (nframes + (size_t)chunk - 1) / (size_t)chunk;
    for (size_t c = (size_t)blockIdx.x * nwaves + wave; c < nchunks; c += (size_t)gridDim.x * nwaves) {
        const long long a = (long long)c * chunk;
        const long long b = a + chunk < (long long)nframes ? a + chunk : (long long)nframes;
        long long f0 = a - R;
        float *cur = B0, *nxt = B1;
        if (f0 <= 0) {  // (chunk >= R: only chunk 0) the carried buffer, or a fresh object's zeros
            f0 = 0;
            for (int i = lane; i < n; i += 64) cur[i] = buf_in ? buf_in[i] : 0.0f;
        } else {
            for (int i = lane; i < n; i += 64) cur[i] = 0.0f;
        }
        wave_lds_sync();
        for (long long f = f0; f < b; f++) {
            ifft_frame<INPUT>(mags + (size_t)f * (size_t)half, phases + (size_t)f * (size_t)half, n, numBits, s_tw, X, lane);
            const bool live = f >= a;  // wave-uniform: the R frames before the chunk only rebuild the buffer
            for (int i = lane; i < n; i += 64) {
                const float r = X[P(i)].x * inv;          // L/fft.cpp:201-209
                const float io = 0.0f + r * window[i];    // :608-610 into the zero-filled ifftOut
                const float v = (i + hop < n ? cur[i + hop] : 0.0f) + io;  // L/maxiFFT.cpp:176-183
                nxt[i] = v;
                if (live) {
                    if (i < hop) out[(size_t)f * (size_t)hop + (size_t)i] = v;
                    if (ifft_out) ifft_out[(size_t)f * (size_t)n + (size_t)i] = io;
                }
            }
            wave_lds_sync();
            float *t = cur;
            cur = nxt;
            nxt = t;
        }
        if (b == (long long)nframes && buf_out)
            for (int i = lane; i < n; i += 64) buf_out[i] = cur[i];
        wave_lds_sync();
    }
}

// The hop buffer of maxiIFFT::process (L/maxiFFT.cpp:176-183): per frame, shift left by hop, zero the
// tail, add the frame.  Position i of the buffer after frame k is therefore the left-to-right sum
// (((carried or 0) + o_{m0}[..]) + ... ) + o_k[i] over the frames that overlap it, oldest first -- a closed
// form each output sample can evaluate on its own, in the reference's order of additions.
__device__ __forceinline__ float ola_value(const float *__restrict__ ifft_out, const float *__restrict__ buf_in,
                                           long long k, int i, int n, int hop) {
    // frames m = k - j contribute o_m[i + j*hop] while i + j*hop < n
    const int J = (n - 1 - i) / hop;
    float s;
    long long m0 = k - J;
    if (m0 <= 0) {  // the state carried into this launch is older than frame 0
        const long long q = (long long)i + (k + 1) * hop;
        s = (buf_in && q < n) ? buf_in[q] : 0.0f;
        m0 = 0;
    } else {
        s = 0.0f;
    }
    for (long long mfr = m0; mfr <= k; mfr++) s += ifft_out[(size_t)mfr * n + (size_t)(i + (k - mfr) * hop)];
    return s;
}

__global__ void ifft_ola_kernel(const float *__restrict__ ifft_out, size_t nframes, int n, int hop,
                                const float *__restrict__ buf_in, float *__restrict__ out,
                                float *__restrict__ buf_out) {
    const size_t total = nframes * (size_t)hop;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) {
        const long long k = (long long)(t / hop);
        out[t] = ola_value(ifft_out, buf_in, k, (int)(t % hop), n, hop);
    } else if (t < total + (size_t)n && buf_out) {
        const int i = (int)(t - total);
        buf_out[i] = ola_value(ifft_out, buf_in, (long long)nframes - 1, i, n, hop);
    }
}


// ---- K6a: fftSize 1024 (half = 512 = 8^3): LDS image, pad8 and round3 live in mxg_spectral.h ------------
template <int OMASK, bool ALIGNED8>
__global__ __launch_bounds__(64 * kWavesPerBlock, MXG_FFT_MINWAVES) __attribute__((target("no-load-store-opt"))) void fft1024_kernel(
    const float *__restrict__ signal, size_t frame_stride, size_t nframes,
    const float *__restrict__ window, const float2 *__restrict__ tw, const float2 *__restrict__ post,
    FftOut out) {
    // one __shared__ object: [tw 512][post 256][X per wave]
    __shared__ float2 s_all[512 + 256 + kWavesPerBlock * kX1024];
    float2 *s_tw = s_all, *s_post = s_all + 512;
    // the wave index is uniform: readfirstlane tells hipcc, so frame bases live in SGPRs and the global
    // loads/stores use the scalar-base + 32-bit-offset form instead of per-lane 64-bit address arithmetic
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *X = s_all + 768 + wave * kX1024;
    for (int i = threadIdx.x; i < 511; i += blockDim.x) s_tw[i] = tw[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_post[i] = post[i];
    __syncthreads();

    const int lo = lane & 7, hi = lane >> 3;
    const int rev6 = (int)(__brev((unsigned)lane) >> 26);
    // window coefficients of the 8 packed elements this lane loads (same for every frame)
    float2 wv[8];
    unsigned li[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int rev3 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);
        li[e] = 2u * (unsigned)(rev3 * 64 + rev6);
        wv[e] = make_float2(window[li[e]], window[li[e] + 1]);
    }
    asm volatile("" : "+v"(wv[0].x), "+v"(wv[0].y), "+v"(wv[1].x), "+v"(wv[1].y), "+v"(wv[2].x), "+v"(wv[2].y),
                 "+v"(wv[3].x), "+v"(wv[3].y));
    asm volatile("" : "+v"(wv[4].x), "+v"(wv[4].y), "+v"(wv[5].x), "+v"(wv[5].y), "+v"(wv[6].x), "+v"(wv[6].y),
                 "+v"(wv[7].x), "+v"(wv[7].y));
    // Frame loads are software-pipelined: the next frame's 8 reads are issued before this frame's
    // outputs are stored, so waiting for them never drains the store queue (in-order vmcnt).
    auto load_frame = [&](size_t fr, float2 (&dst)[8]) {
        // the frame index is wave-uniform: pin it to an SGPR so the row base is scalar arithmetic
        const unsigned fu = __builtin_amdgcn_readfirstlane((unsigned)(fr < nframes ? fr : nframes - 1));
        const float *x = signal + (size_t)fu * frame_stride;
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
    const size_t fstep = (size_t)gridDim.x * kWavesPerBlock;
    float2 nxt[8];
    load_frame((size_t)blockIdx.x * kWavesPerBlock + wave, nxt);
    for (size_t f = (size_t)blockIdx.x * kWavesPerBlock + wave; f < nframes; f += fstep) {
        float2 v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            v[e].x = nxt[e].x * wv[e].x;  // calcFFT L/fft.cpp:501-503
            v[e].y = nxt[e].y * wv[e].y;
        }
        load_frame(f + fstep, nxt);
    // round A twiddles are lane-uniform: stage 0 n=0; stage 1 n=0,1; stage 2 n=0..3
        const float2 a0 = s_tw[0];
        const float2 a1[2] = {s_tw[1], s_tw[2]};
        const float2 a2[4] = {s_tw[3], s_tw[4], s_tw[5], s_tw[6]};
        // round B: stages 3,4,5 (h = 8,16,32): n = lo, (e&1)*8+lo, (e&3)*8+lo
        const float2 b0 = s_tw[7 + lo];
        const float2 b1[2] = {s_tw[15 + lo], s_tw[15 + 8 + lo]};
        const float2 b2[4] = {s_tw[31 + lo], s_tw[31 + 8 + lo], s_tw[31 + 16 + lo], s_tw[31 + 24 + lo]};
        // round C: stages 6,7,8 (h = 64,128,256): n = lane, (e&1)*64+lane, (e&3)*64+lane
        const float2 c0 = s_tw[63 + lane];
        const float2 c1[2] = {s_tw[127 + lane], s_tw[127 + 64 + lane]};
        const float2 c2[4] = {s_tw[255 + lane], s_tw[255 + 64 + lane], s_tw[255 + 128 + lane],
                              s_tw[255 + 192 + lane]};

        // Round A input (already in v): lane holds idx = 8*lane + e  <-  packed element
        // i = rev9(idx) = rev3(e)*64 + rev6(lane): for each e one 512-B segment per wavefront.
        round3(v, a0, a1, a2);
        // transpose A->B: write idx = 8*lane + e, read idx = hi*64 + e*8 + lo
#pragma unroll
        for (int e = 0; e < 8; e++) X[pad8(8 * lane + e)] = v[e];
        wave_lds_sync();
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = X[pad8(hi * 64 + e * 8 + lo)];
        round3(v, b0, b1, b2);
        wave_lds_sync();
        // transpose B->C: write same positions, read idx = e*64 + lane
#pragma unroll
        for (int e = 0; e < 8; e++) X[pad8(hi * 64 + e * 8 + lo)] = v[e];
        wave_lds_sync();
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = X[pad8(e * 64 + lane)];
        round3(v, c0, c1, c2);
        wave_lds_sync();
        // natural order back to LDS for the (i, 512-i) pairing of the post-pass
#pragma unroll
        for (int e = 0; e < 8; e++) X[pad8(e * 64 + lane)] = v[e];
        wave_lds_sync();
        const size_t base = (size_t)__builtin_amdgcn_readfirstlane((unsigned)f) * (size_t)512;
        FftOut row = {out.real ? out.real + base : nullptr, out.imag ? out.imag + base : nullptr,
                      out.mags ? out.mags + base : nullptr, out.phases ? out.phases + base : nullptr};
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const unsigned i = 1u + (unsigned)lane + 64u * (unsigned)q;  // 1..256
            if (i < 256u) {
                float2 a = X[pad8((int)i)], b = X[pad8(512 - (int)i)];
                post_pair(a, b, s_post[i]);
                emit_row_t<OMASK>(row, i, a);
                emit_row_t<OMASK>(row, 512u - i, b);
            } else {  // i == 256 (lane 63, q 3): the untouched middle bin; and bin 0
                emit_row_t<OMASK>(row, 256u, X[pad8(256)]);
                float2 z = X[pad8(0)];
                float2 z0 = {z.x + z.y, z.x - z.y};
                emit_row_t<OMASK>(row, 0u, z0);
            }
        }
        wave_lds_sync();
    }
}

}  // namespace
}  // namespace mxg

using namespace mxg;

// ---- magsToDB / spectralFlatness / spectralCentroid (L/fft.cpp:526-534, L/maxiFFT.cpp:113-132) ----
// One wavefront per 64 frames.  The per-frame sums of the reference are sequential float
// accumulations over the bins, so a lane owns a frame and walks its bins in order; the [64 frames]
// x [64 bins] tile it walks is loaded coalesced (lane = bin) and transposed through LDS.  The dB
// conversion is elementwise and is done in the coalesced phase.
// Arithmetic: `in < 0.000001` compares in double; `20.0*log10(in+1)` = double product of the FLOAT
// log10 (float overload) rounded to float; flatness/centroid are all-float (fabs(float), size_t i
// converted to float).  logf/log10f/expf are the device's => stated tolerance; centroid has no
// transcendental and is bit-exact.
namespace mxg {
namespace {
__global__ __launch_bounds__(64) void fft_features_kernel(const float *__restrict__ mags, size_t nframes,
                                                           int bins, float binhz, float *__restrict__ db,
                                                           float *__restrict__ flat, float *__restrict__ cen) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    for (size_t f0 = (size_t)blockIdx.x * 64; f0 < nframes; f0 += (size_t)gridDim.x * 64) {
        float gm = 0, am = 0, x = 0, y = 0;
        const int rows = (nframes - f0 < 64) ? (int)(nframes - f0) : 64;
        for (int b0 = 0; b0 < bins; b0 += 64) {
            const int b = b0 + lane;
            const bool bvalid = b < bins;
#pragma unroll 8
            for (int r = 0; r < 64; r++) {
                float v = 0.0f;
                if (bvalid && r < rows) {
                    const size_t at = (f0 + r) * (size_t)bins + b;
                    v = mags[at];
                    if (db) db[at] = ((double)v < 0.000001) ? 0.0f : (float)(20.0 * (double)log10f(v + 1));
                }
                tile[r][lane] = v;
            }
            __syncthreads();
            if (flat || cen) {
                const int nb = (bins - b0 < 64) ? bins - b0 : 64;
                for (int j = 0; j < nb; j++) {
                    const float m = tile[lane][j];
                    if (flat) {
                        if (m != 0) gm += logf(m);
                        am += m;
                    }
                    if (cen) {
                        x += fabsf(m) * (float)(b0 + j);
                        y += fabsf(m);
                    }
                }
            }
            __syncthreads();
        }
        if (lane < rows) {
            if (flat) {
                gm = expf(gm / (float)bins);
                am /= (float)bins;
                flat[f0 + lane] = am != 0 ? gm / am : 0.0f;
            }
            if (cen) cen[f0 + lane] = y != 0 ? x / y * binhz : 0.0f;
        }
    }
}
}  // namespace
}  // namespace mxg

extern "C" {

mxg_fft_plan *mxg_fft_plan_create(int fftSize, int hopSize, int windowSize) {
    if (ensure_init_only()) return nullptr;
    if (fftSize < 8 || fftSize > 8192 || (fftSize & (fftSize - 1))) {
        fail(MXG_ERR_INVALID, "mxg_fft_plan_create: fftSize %d must be a power of two in [8, 8192] "
                              "(the reference exit(1)s on a non power of two, fft.cpp:129-132)", fftSize);
        return nullptr;
    }
    int win = windowSize > fftSize ? windowSize : fftSize;  // L/maxiFFT.cpp:48
    if (win > fftSize) {
        fail(MXG_ERR_INVALID, "mxg_fft_plan_create: windowSize %d > fftSize %d overruns the reference's "
                              "window/buffer vectors (maxiFFT.cpp:51-58); not supported", windowSize, fftSize);
        return nullptr;
    }
    if (hopSize <= 0 || hopSize > win) {
        fail(MXG_ERR_INVALID, "mxg_fft_plan_create: hopSize %d out of (0, %d]", hopSize, win);
        return nullptr;
    }
    const int half = fftSize / 2;
    int numBits = 0;
    while (!(half & (1 << numBits))) numBits++;
    // Hann window, type 3 of fft::genWindow (L/fft.cpp:409-413), double -> float
    std::vector<float> window(fftSize);
    for (int i = 0; i < win; i++) window[i] = 0.50 - 0.50 * cos(2 * M_PI * i / (win - 1));
    // stage twiddles: replay of L/fft.cpp:156-182 for the half-size complex transform
    std::vector<float2> tw(half);
    {
        const double angle_numerator = 2.0 * M_PI;
        int BlockEnd = 1;
        for (int BlockSize = 2; BlockSize <= half; BlockSize <<= 1) {
            double delta_angle = angle_numerator / (double)BlockSize;
            float sm2 = sin(-2 * delta_angle);
            float sm1 = sin(-delta_angle);
            float cm2 = cos(-2 * delta_angle);
            float cm1 = cos(-delta_angle);
            float w = 2 * cm1;
            float ar2 = cm2, ar1 = cm1, ai2 = sm2, ai1 = sm1;
            for (int n = 0; n < BlockEnd; n++) {
                float ar0 = w * ar1 - ar2;
                ar2 = ar1;
                ar1 = ar0;
                float ai0 = w * ai1 - ai2;
                ai2 = ai1;
                ai1 = ai0;
                tw[BlockEnd - 1 + n] = make_float2(ar0, ai0);
            }
            BlockEnd = BlockSize;
        }
    }
    // post-pass twiddles: replay of L/fft.cpp:233-272
    std::vector<float2> post(half / 2 > 0 ? half / 2 : 1);
    {
        float theta = M_PI / half;
        float wtemp = float(sin(0.5 * theta));
        float wpr = -2.0 * wtemp * wtemp;
        float wpi = float(sin(theta));
        float wr = 1.0 + wpr;
        float wi = wpi;
        post[0] = make_float2(0.f, 0.f);
        for (int i = 1; i < half / 2; i++) {
            post[i] = make_float2(wr, wi);
            wtemp = wr;
            wr = wtemp * wpr - wi * wpi + wr;
            wi = wi * wpr + wtemp * wpi + wi;
        }
    }
    mxg_fft_plan *p = new mxg_fft_plan();
    p->fftSize = fftSize;
    p->hopSize = hopSize;
    p->windowSize = win;
    p->bins = half;
    p->half = half;
    p->numBits = numBits;
    p->d_window = nullptr;
    p->d_tw = nullptr;
    p->d_post = nullptr;
    p->d_tw8 = nullptr;
    // the stage-opening twiddles of the first two stages are exactly (1, 0) (the recurrence w * ar1 - ar2 is exact there): the fused
    // kernel then skips those multiplications (spectral.hip, round3_s1); verified here rather than assumed
    p->round1Trivial = half >= 4 && tw[0].x == 1.0f && tw[0].y == 0.0f && tw[1].x == 1.0f && tw[1].y == 0.0f;
    if (fftSize == 1024) {
        // tolerance mode of the fused kernel (spectral.hip, round8_t): a lane's round is x_e *= T_e = e^(+i m_e theta), m = 4,2,6,1,5,3,7
        // for e = 1..7 (the sign convention of L/fft.cpp:161-182: twiddle n of a stage is (cos, +sin)(2 pi n / BlockSize)), then an
        // 8-point DFT; theta = 2 pi (lane & 7) / 64 in the second round, 2 pi lane / 512 in the third.  Correctly rounded from double.
        std::vector<float2> tw8((8 + 64) * 7);
        const int mexp[7] = {4, 2, 6, 1, 5, 3, 7};
        for (int l = 0; l < 8; l++)
            for (int e = 0; e < 7; e++) {
                const double a = 2.0 * M_PI * (double)(l * mexp[e]) / 64.0;
                tw8[(size_t)l * 7 + e] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int l = 0; l < 64; l++)
            for (int e = 0; e < 7; e++) {
                const double a = 2.0 * M_PI * (double)(l * mexp[e]) / 512.0;
                tw8[56 + (size_t)l * 7 + e] = make_float2((float)cos(a), (float)sin(a));
            }
        if (check_hip(hipMalloc(&p->d_tw8, sizeof(float2) * tw8.size()), "hipMalloc") ||
            check_hip(hipMemcpy(p->d_tw8, tw8.data(), sizeof(float2) * tw8.size(), hipMemcpyHostToDevice), "hipMemcpy")) {
            mxg_fft_plan_destroy(p);
            return nullptr;
        }
    }
    if (check_hip(hipMalloc(&p->d_window, sizeof(float) * fftSize), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_tw, sizeof(float2) * tw.size()), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_post, sizeof(float2) * post.size()), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_window, window.data(), sizeof(float) * fftSize, hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_post, post.data(), sizeof(float2) * post.size(), hipMemcpyHostToDevice), "hipMemcpy")) {
        mxg_fft_plan_destroy(p);
        return nullptr;
    }
    return p;
}

int mxg_fft_plan_destroy(mxg_fft_plan *p) {
    if (!p) return MXG_OK;
    if (p->d_window) (void)hipFree(p->d_window);
    if (p->d_tw8) (void)hipFree(p->d_tw8);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_post) (void)hipFree(p->d_post);
    delete p;
    return MXG_OK;
}

int mxg_fft_plan_bins(const mxg_fft_plan *p) { return p ? p->bins : MXG_ERR_INVALID; }

int mxg_fft_batch(const mxg_fft_plan *p, const float *d_signal, size_t frame_stride, size_t nframes,
                  float *d_real, float *d_imag, float *d_mags, float *d_phases, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(p && d_signal, "null plan or signal");
    MXG_REQUIRE(d_real || d_imag || d_mags || d_phases, "no output requested");
    MXG_REQUIRE(nframes < ((size_t)1 << 32), "nframes must be < 2^32");
    if (nframes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    FftOut out = {d_real, d_imag, d_mags, d_phases};
    size_t blocks = (nframes + kWavesPerBlock - 1) / kWavesPerBlock;
    const size_t cap = 256 * 5;  // persistent: <= 5 workgroups per CU, grid-stride over frames
    if (blocks > cap) blocks = cap;
    int force_generic = tune_get("fft_generic");
    const int omask = (d_real ? 1 : 0) | (d_imag ? 2 : 0) | (d_mags ? 4 : 0) | (d_phases ? 8 : 0);
    const bool fast_mask = omask == 3 || omask == 4 || omask == 12 || omask == 15;
    const bool fast = p->fftSize == 1024 && !force_generic && fast_mask;
    KernelTimer kt(fast ? "fft1024_kernel" : "fft_generic_kernel", st);
    if (fast) {
        const bool aligned8 = (((uintptr_t)d_signal) & 7) == 0 && (frame_stride & 1) == 0;
#define MXG_FFT_LAUNCH(M, A)                                                                            \
    hipLaunchKernelGGL((fft1024_kernel<M, A>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), 0, st, \
                       d_signal, frame_stride, nframes, p->d_window, p->d_tw, p->d_post, out)
#define MXG_FFT_LAUNCH2(M) \
    if (aligned8) MXG_FFT_LAUNCH(M, true); else MXG_FFT_LAUNCH(M, false)
        switch (omask) {
            case 3: MXG_FFT_LAUNCH2(3); break;
            case 4: MXG_FFT_LAUNCH2(4); break;
            case 12: MXG_FFT_LAUNCH2(12); break;
            default: MXG_FFT_LAUNCH2(15); break;
        }
#undef MXG_FFT_LAUNCH2
#undef MXG_FFT_LAUNCH
    } else {
        // one LDS frame per wave: 4 waves up to 2048 points, 2 at 4096, 1 at 8192
        const int waves = p->fftSize <= 2048 ? kWavesPerBlock : (p->fftSize <= 4096 ? 2 : 1);
        const size_t lds = sizeof(float2) * (waves * (size_t)(p->half + (p->half >> 5) + 1) + (size_t)p->half);  // X per wave + stage twiddles
        if (lds > 64 * 1024)  // (8192 points: 33.8 KB + 32 KB)
            MXG_HIP(hipFuncSetAttribute((const void *)fft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        blocks = (nframes + waves - 1) / waves;
        if (blocks > cap * (kWavesPerBlock / waves)) blocks = cap * (kWavesPerBlock / waves);
        hipLaunchKernelGGL(fft_generic_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, st,
                           d_signal, frame_stride, nframes, p->fftSize, p->numBits, p->d_window, p->d_tw,
                           p->d_post, out);
    }
    return check_hip(hipGetLastError(), "fft kernel launch");
}

int mxg_fft_features(const mxg_fft_plan *p, const float *d_mags, size_t nframes, float *d_db,
                     float *d_flatness, float *d_centroid, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(p && d_mags, "null plan or magnitudes");
    MXG_REQUIRE(d_db || d_flatness || d_centroid, "no output requested");
    if (nframes == 0) return MXG_OK;
    size_t blocks = (nframes + 63) / 64;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // (float) maxiSettings::sampleRate / fftSize, L/maxiFFT.cpp:131
    const float binhz = (float)settings().sampleRate / (float)p->fftSize;
    hipLaunchKernelGGL(fft_features_kernel, dim3((unsigned)blocks), dim3(64), 0, resolve_stream(stream), d_mags,
                       nframes, p->bins, binhz, d_db, d_flatness, d_centroid);
    return check_hip(hipGetLastError(), "fft_features_kernel launch");
}

mxg_ifft_plan *mxg_ifft_plan_create(int fftSize, int hopSize, int windowSize) {
    if (ensure_init_only()) return nullptr;
    if (fftSize < 8 || fftSize > 8192 || (fftSize & (fftSize - 1))) {
        fail(MXG_ERR_INVALID, "mxg_ifft_plan_create: fftSize %d must be a power of two in [8, 8192]", fftSize);
        return nullptr;
    }
    const int win = windowSize ? windowSize : fftSize;  // L/maxiFFT.cpp:143
    if (win < 2 || win > fftSize) {
        fail(MXG_ERR_INVALID, "mxg_ifft_plan_create: windowSize %d outside [2, fftSize] (genWindow would overrun "
                              "the reference's window vector, maxiFFT.cpp:151-152)", windowSize);
        return nullptr;
    }
    if (hopSize <= 0 || hopSize > fftSize) {
        fail(MXG_ERR_INVALID, "mxg_ifft_plan_create: hopSize %d out of (0, %d]", hopSize, fftSize);
        return nullptr;
    }
    int numBits = 0;
    while (!(fftSize & (1 << numBits))) numBits++;
    std::vector<float> window(fftSize, 0.0f);
    for (int i = 0; i < win; i++) window[i] = 0.50 - 0.50 * cos(2 * M_PI * i / (win - 1));  // L/fft.cpp:409-413
    std::vector<float2> tw(fftSize);
    {   // replay of L/fft.cpp:137-182 with InverseTransform = true
        const double angle_numerator = -(2.0 * M_PI);
        int BlockEnd = 1;
        for (int BlockSize = 2; BlockSize <= fftSize; BlockSize <<= 1) {
            double delta_angle = angle_numerator / (double)BlockSize;
            float sm2 = sin(-2 * delta_angle);
            float sm1 = sin(-delta_angle);
            float cm2 = cos(-2 * delta_angle);
            float cm1 = cos(-delta_angle);
            float w = 2 * cm1;
            float ar2 = cm2, ar1 = cm1, ai2 = sm2, ai1 = sm1;
            for (int n = 0; n < BlockEnd; n++) {
                float ar0 = w * ar1 - ar2;
                ar2 = ar1;
                ar1 = ar0;
                float ai0 = w * ai1 - ai2;
                ai2 = ai1;
                ai1 = ai0;
                tw[BlockEnd - 1 + n] = make_float2(ar0, ai0);
            }
            BlockEnd = BlockSize;
        }
    }
    mxg_ifft_plan *p = new mxg_ifft_plan();
    p->fftSize = fftSize; p->hopSize = hopSize; p->windowSize = win; p->bins = fftSize / 2; p->numBits = numBits;
    p->d_window = nullptr; p->d_tw = nullptr;
    if (check_hip(hipMalloc(&p->d_window, sizeof(float) * fftSize), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_tw, sizeof(float2) * fftSize), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_window, window.data(), sizeof(float) * fftSize, hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * fftSize, hipMemcpyHostToDevice), "hipMemcpy")) {
        mxg_ifft_plan_destroy(p);
        return nullptr;
    }
    return p;
}

int mxg_ifft_plan_destroy(mxg_ifft_plan *p) {
    if (!p) return MXG_OK;
    if (p->d_window) (void)hipFree(p->d_window);
    if (p->d_tw) (void)hipFree(p->d_tw);
    delete p;
    return MXG_OK;
}

static int ifft_batch_impl(int input, const mxg_ifft_plan *p, const float *d_a, const float *d_b, size_t nframes,
                           float *d_buffer, float *d_signal, float *d_ifft_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(p && d_signal && (input == 2 || (d_a && d_b)), "null plan or pointer");
    if (nframes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const int n = p->fftSize;
    // the streaming form: transform + hop buffer in one kernel (K6s) whenever a frame overlaps at most 15 later ones
    const long long R = (long long)((n + p->hopSize - 1) / p->hopSize) - 1;
    // Measured at 1024 points, 262 144 frames: hop 1024 1.75 ms streaming vs 1.88 ms as two kernels, hop 512 1.66 vs 1.61, hop 256
    // 1.68 vs 1.49 (the streaming wavefront carries 8 KB more LDS, re-runs R frames per chunk and adds an LDS pass per frame): it is
    // the default where a frame overlaps at most one later one; knob ifft_stream = 2 forces it wherever it fits.
    const int want = tune_get("ifft_stream");
    if (p->hopSize <= n && R <= 15 && n <= 4096 && (want == 2 || (want == 1 && R <= 1))) {  // (8192 points: X + two buffer images exceed the LDS)
        const float *buf_in = nullptr;
        if (d_buffer) {  // chunk 0 reads the carried buffer while the last chunk's wavefront writes the new one: stage the old one
            float *tmp = nullptr;
            if (int s = scratch_get(SCR_IFFT_BUF, st, sizeof(float) * n, (void **)&tmp)) return s;
            MXG_HIP(hipMemcpyAsync(tmp, d_buffer, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
            buf_in = tmp;
        }
        long long chunk = 64;
        if (nframes < (size_t)64 * 2048) {
            chunk = (long long)((nframes + 2047) / 2048);
            const long long lo = 4 * R > 8 ? 4 * R : 8;
            chunk = chunk < lo ? lo : chunk;
        }
        const int waves = n <= 1024 ? 4 : (n <= 2048 ? 2 : 1);
        const size_t per_wave = sizeof(float2) * ((size_t)(n + (n >> 5) + 1) + (size_t)n);
        const size_t lds = per_wave * waves + sizeof(float2) * (size_t)n;
        typedef void (*skern_t)(const float *, const float *, size_t, int, int, int, int, const float *, const float2 *, const float *,
                                float *, float *, float *);
        skern_t k = input == 0 ? ifft_stream_kernel<0> : (input == 1 ? ifft_stream_kernel<1> : ifft_stream_kernel<2>);
        if (lds > 64 * 1024) MXG_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const size_t nchunks = (nframes + (size_t)chunk - 1) / (size_t)chunk;
        size_t blocks = (nchunks + waves - 1) / waves;
        if (blocks > 256 * 8) blocks = 256 * 8;
        KernelTimer kt("ifft_stream_kernel", st);
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * waves), lds, st, d_a, d_b, nframes, n, p->numBits, p->hopSize, (int)chunk,
                           p->d_window, p->d_tw, buf_in, d_signal, d_buffer, d_ifft_out);
        return check_hip(hipGetLastError(), "ifft_stream_kernel launch");
    }
    float *io = d_ifft_out;
    if (!io) {  // grow-only scratch for the per-frame transforms
        if (int s = scratch_get(SCR_IFFT_OUT, st, nframes * (size_t)n * sizeof(float), (void **)&io)) return s;
    }
    const size_t per_wave = sizeof(float2) * (size_t)(n + (n >> 5) + 1);
    int waves = n <= 1024 ? 4 : (n <= 2048 ? 2 : 1);
    const size_t lds = per_wave * waves + sizeof(float2) * (size_t)n;  // + the stage twiddles
    typedef void (*kern_t)(const float *, const float *, size_t, int, int, const float *, const float2 *, float *);
    kern_t k = input == 0 ? ifft_generic_kernel<0> : (input == 1 ? ifft_generic_kernel<1> : ifft_generic_kernel<2>);
    if (lds > 64 * 1024) MXG_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    size_t blocks = (nframes + waves - 1) / waves;
    if (blocks > 256 * 8) blocks = 256 * 8;
    {
        KernelTimer kt("ifft_generic_kernel", st);
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * waves), lds, st, d_a, d_b, nframes, n, p->numBits, p->d_window,
                           p->d_tw, io);
    }
    MXG_HIP(hipGetLastError());
    // overlap-add: the carried buffer is read while the new one is written -> stage the old one
    const float *buf_in = nullptr;
    if (d_buffer) {
        float *tmp = nullptr;
        if (int s = scratch_get(SCR_IFFT_BUF, st, sizeof(float) * n, (void **)&tmp)) return s;
        MXG_HIP(hipMemcpyAsync(tmp, d_buffer, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        buf_in = tmp;
    }
    const size_t total = nframes * (size_t)p->hopSize + (size_t)n;
    hipLaunchKernelGGL(ifft_ola_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, io, nframes, n, p->hopSize,
                       buf_in, d_signal, d_buffer);
    return check_hip(hipGetLastError(), "ifft kernels launch");
}

int mxg_ifft_batch(const mxg_ifft_plan *p, const float *d_mags, const float *d_phases, size_t nframes,
                   float *d_buffer, float *d_signal, float *d_ifft_out, void *stream) {
    return ifft_batch_impl(0, p, d_mags, d_phases, nframes, d_buffer, d_signal, d_ifft_out, stream);
}

int mxg_ifft_batch_complex(const mxg_ifft_plan *p, const float *d_real, const float *d_imag, size_t nframes,
                           int as_reference, float *d_buffer, float *d_signal, float *d_ifft_out, void *stream) {
    return ifft_batch_impl(as_reference ? 2 : 1, p, d_real, d_imag, nframes, d_buffer, d_signal, d_ifft_out, stream);
}

}  // extern "C"
