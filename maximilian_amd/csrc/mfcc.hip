// mfcc.hip -- maxiMFCC batch on gfx950: mel filterbank -> log-square -> DCT over many spectra.
//
// Path (reference, L/ = src/libs/): maxiMFCCAnalyser<double>::setup L/maxiMFCC.h:56-75
// (calcMelFilterBank :118-182, createDCTCoeffs :183-203), mfcc() :77-81 ->
// melFilterAndLogSq_Part2 L/maxiMFCC.cpp:48-66 -> dct L/maxiMFCC.h:98-111.
//
// The tables are built ON THE HOST with the host libm by the same expressions (log10, pow, sqrt,
// cos), so they are bit-identical to the reference's.  Column 0 of melFilters is never written
// by the reference (its loop starts at filter 1, :149); it is defined as 0 here.
//
// Exactness argument.  The reference's melBands[f] = sum_bin W[f][bin]*spec[bin] runs over ALL
// bins in increasing order, but W is a bank of triangles: outside a filter's support every term
// is an exact +0.0 (spec >= 0), and x + (+0.0) == x.  Summing only the support, still in
// increasing bin order, is therefore bit-identical to the dense loop while doing ~1/50 of the
// work (411 non-zeros of 21 504 for 512/42).  The DCT is accumulated on the fly in the
// reference's j-order (each finished band updates every coefficient), then divided by numCoeffs.
// melraw (pre-log) is bit-exact; log() is OCML's, so melbands/mfcc carry the tolerance stated
// in DESIGN.md.
//
// K7a `mfcc_stream_kernel` (default).  One LANE owns one frame and streams through its spectrum
// once, bin-major: the triangles overlap by half, so at any bin at most S (=2 for mel banks)
// filters are "open"; each open filter lives in a register accumulator slot, and the host-built
// schedule (per bin: S weights + which slots close) is wave-uniform, staged in LDS and read as
// broadcasts.  Every lane is busy, no divergence, ~60 VGPRs => full occupancy; spectra are read
// straight from HBM/L2 as 16-B per-lane loads (each 128-B line is consumed by 8 consecutive
// loads of the same lane).  Algorithmic bytes/frame: nbUsed*4 read + numCoeffs*8 written.
// K7a' `mfcc_tile_kernel`: filter-major fallback for banks whose supports overlap more than 4 deep.
//
// K7b `mfcc_mfma_kernel` (method 1): the dense contraction frames[N x K] x W[K x 48] on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64), as the reference's own vDSP branch does with
// vDSP_mmulD (L/maxiMFCC.cpp:28-37).  FMA chains change the rounding => tolerance on melraw too.
// On gfx950 the fp64 MFMA rate equals the fp64 VALU FMA rate, so this path exists for dense
// (non-triangular) filter matrices and as the MFMA-utilisation measurement, not as the fast path.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "mxg_common.h"

#include "mxg_spectral.h"

namespace mxg {
namespace {

// ---- K7a: bin-major streaming, one lane per frame ------------------------------------------------
template <int S, int NC>
__global__ __launch_bounds__(256) void mfcc_stream_kernel(
    const float *__restrict__ mags, size_t mag_stride, size_t nframes, unsigned numFilters,
    unsigned nbUsed, const double *__restrict__ schedW, const int *__restrict__ schedFin,
    const int *__restrict__ lo, const int *__restrict__ hi, const double *__restrict__ dct,
    double *__restrict__ melraw, double *__restrict__ melbands, double *__restrict__ mfcc,
    int aligned16) {
    extern __shared__ double s_dyn[];  // [nbUsed*S] weights | [numFilters*NC] dct | [nbUsed*S] fin (int)
    double *s_w = s_dyn;
    double *s_d = s_dyn + (size_t)nbUsed * S;
    int *s_fin = reinterpret_cast<int *>(s_d + (size_t)numFilters * NC);
    for (unsigned i = threadIdx.x; i < nbUsed * S; i += blockDim.x) {
        s_w[i] = schedW[i];
        s_fin[i] = schedFin[i];
    }
    for (unsigned i = threadIdx.x; i < numFilters * NC; i += blockDim.x) s_d[i] = dct[i];
    __syncthreads();
    const size_t frame = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= nframes) return;
    const float *row = mags + frame * mag_stride;
    double acc[S], c[NC];
    if (melraw || melbands) {  // bands with an empty support never close: they are 0 / log-square 0
        for (unsigned f = 0; f < numFilters; f++)
            if (hi[f] < lo[f]) {  // wave-uniform
                if (melraw) melraw[frame * numFilters + f] = 0.0;
                if (melbands) melbands[frame * numFilters + f] = 0.0;
            }
    }
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = 0.0;  // L/maxiMFCC.cpp:52
#pragma unroll
    for (int i = 0; i < NC; i++) c[i] = 0.0;  // L/maxiMFCC.h:99-101
    for (unsigned b0 = 0; b0 < nbUsed; b0 += 4) {
        float x4[4];
        if (aligned16) {
            const float4 t = *reinterpret_cast<const float4 *>(row + b0);  // row padded to >= b0+4
            x4[0] = t.x; x4[1] = t.y; x4[2] = t.z; x4[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) x4[j] = (b0 + j < nbUsed) ? row[b0 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned b = b0 + j;
            if (b < nbUsed) {  // wave-uniform
                const double x = (double)x4[j];
#pragma unroll
                for (int s = 0; s < S; s++) acc[s] += (s_w[b * S + s] * x);  // L/maxiMFCC.cpp:57
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int f = s_fin[b * S + s];
                    if (f >= 0) {  // wave-uniform: band f is complete
                        if (melraw) melraw[frame * numFilters + f] = acc[s];
                        const double mb = log_square(acc[s]);
                        if (melbands) melbands[frame * numFilters + f] = mb;
                        const double *d = s_d + (size_t)f * NC;
#pragma unroll
                        for (int i = 0; i < NC; i++) c[i] += (d[i] * mb);  // L/maxiMFCC.h:105
                        acc[s] = 0.0;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NC; i++) mfcc[frame * NC + i] = c[i] / (double)NC;  // L/maxiMFCC.h:109
}

// ---- K7a-t: the same bin-major stream, spectra staged through LDS ---------------------------------------
// In K7a a wave's 16-B row loads touch 64 different cache lines per instruction (lane = frame, 2 KB
// apart) and 16 waves per CU overflow the 32 KB L1.  Here a wavefront first loads a [64 frames x 32
// bins] tile COALESCED (8 lanes cover the 128 B of one frame, one instruction = 8 frames), parks it in
// its private LDS tile (row stride 36 floats: 16-B aligned, conflict-free for the per-frame b128 reads)
// and then walks it lane = frame exactly like K7a.  The next tile's global loads are issued before the
// current tile is consumed.  Same additions in the same order: bit-identical to K7a.
constexpr int kTileBins = 32, kTileStride = 36;

template <int S, int NC>
__global__ __launch_bounds__(256) void mfcc_stream_tiled_kernel(
    const float *__restrict__ mags, size_t mag_stride, size_t nframes, unsigned numFilters,
    unsigned nbUsed, const double *__restrict__ schedW, const int *__restrict__ schedFin,
    const int *__restrict__ lo, const int *__restrict__ hi, const double *__restrict__ dct,
    double *__restrict__ melraw, double *__restrict__ melbands, double *__restrict__ mfcc) {
    extern __shared__ double s_dyn[];  // [nbUsed*S] weights | [numFilters*NC] dct | [nbUsed*S] fin (int) | 4 tiles
    double *s_w = s_dyn;
    double *s_d = s_dyn + (size_t)nbUsed * S;
    int *s_fin = reinterpret_cast<int *>(s_d + (size_t)numFilters * NC);
    // tiles start on a 16-B boundary after the int table
    float *s_tiles = reinterpret_cast<float *>(s_fin + (((size_t)nbUsed * S + 3) & ~(size_t)3));
    for (unsigned i = threadIdx.x; i < nbUsed * S; i += blockDim.x) {
        s_w[i] = schedW[i];
        s_fin[i] = schedFin[i];
    }
    for (unsigned i = threadIdx.x; i < numFilters * NC; i += blockDim.x) s_d[i] = dct[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = s_tiles + (size_t)wave * 64 * kTileStride;
    const size_t frame0 = ((size_t)blockIdx.x * 4 + wave) * 64;  // this wave's 64 frames
    if (frame0 >= nframes) return;
    const size_t frame = frame0 + lane;
    const bool live = frame < nframes;
    // coalesced tile load: lane -> (row = lane/8 + 8*k, 4 bins at 4*(lane%8)), k = 0..7
    const int lrow = lane >> 3, lcol = (lane & 7) * 4;
    const float *gsrc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        size_t fr = frame0 + lrow + 8 * k;
        if (fr >= nframes) fr = nframes - 1;  // clamped: valid memory, results of dead rows are discarded
        gsrc[k] = mags + fr * mag_stride + lcol;
    }
    const unsigned ntiles = (nbUsed + kTileBins - 1) / kTileBins;
    float nxt[8][4];  // plain floats: a float4 array here is not promoted to registers by hipcc
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float4 v4 = *reinterpret_cast<const float4 *>(gsrc[k]);
        nxt[k][0] = v4.x; nxt[k][1] = v4.y; nxt[k][2] = v4.z; nxt[k][3] = v4.w;
    }
    double acc[S], c[NC];
    if (live && (melraw || melbands)) {  // bands with an empty support never close: they are 0 / log-square 0
        for (unsigned f = 0; f < numFilters; f++)
            if (hi[f] < lo[f]) {  // wave-uniform
                if (melraw) melraw[frame * numFilters + f] = 0.0;
                if (melbands) melbands[frame * numFilters + f] = 0.0;
            }
    }
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = 0.0;  // L/maxiMFCC.cpp:52
#pragma unroll
    for (int i = 0; i < NC; i++) c[i] = 0.0;  // L/maxiMFCC.h:99-101
    for (unsigned t = 0; t < ntiles; t++) {
        // park tile t, then request tile t+1 (columns clamped to the last tile: surplus unused)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 v4 = {nxt[k][0], nxt[k][1], nxt[k][2], nxt[k][3]};
            *reinterpret_cast<float4 *>(tile + (lrow + 8 * k) * kTileStride + lcol) = v4;
        }
        const unsigned tn = (t + 1 < ntiles) ? t + 1 : t;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 v4 = *reinterpret_cast<const float4 *>(gsrc[k] + (size_t)tn * kTileBins);
            nxt[k][0] = v4.x; nxt[k][1] = v4.y; nxt[k][2] = v4.z; nxt[k][3] = v4.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float *row = tile + lane * kTileStride;
        const unsigned bbase = t * kTileBins;
        for (int j4 = 0; j4 < kTileBins / 4; j4++) {
            const float4 xv = *reinterpret_cast<const float4 *>(row + 4 * j4);
            const float x4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned b = bbase + 4 * j4 + j;
                if (b < nbUsed) {  // wave-uniform
                    const double x = (double)x4[j];
#pragma unroll
                    for (int s = 0; s < S; s++) acc[s] += (s_w[b * S + s] * x);  // L/maxiMFCC.cpp:57
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const int f = s_fin[b * S + s];
                        if (f >= 0) {  // wave-uniform: band f is complete
                            if (live && melraw) melraw[frame * numFilters + f] = acc[s];
                            const double mb = log_square(acc[s]);
                            if (live && melbands) melbands[frame * numFilters + f] = mb;
                            const double *d = s_d + (size_t)f * NC;
#pragma unroll
                            for (int i = 0; i < NC; i++) c[i] += (d[i] * mb);  // L/maxiMFCC.h:105
                            acc[s] = 0.0;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < NC; i++) mfcc[frame * NC + i] = c[i] / (double)NC;  // L/maxiMFCC.h:109
    }
}

// ---- K7a': filter-major over an LDS tile of 64 spectra (fallback, any bank / any numCoeffs) ----
__global__ __launch_bounds__(64) void mfcc_tile_kernel(
    const float *__restrict__ mags, size_t mag_stride, size_t nframes, unsigned numFilters,
    unsigned numCoeffs, unsigned nbUsed, unsigned tileStride, const int *__restrict__ lo,
    const int *__restrict__ hi, const int *__restrict__ off, const double *__restrict__ Wc,
    const double *__restrict__ dct, double *__restrict__ melraw, double *__restrict__ melbands,
    double *__restrict__ mfcc) {
    extern __shared__ float s_tile[];  // [64][tileStride], tileStride odd
    const int lane = threadIdx.x;
    for (size_t f0 = (size_t)blockIdx.x * 64; f0 < nframes; f0 += (size_t)gridDim.x * 64) {
        const size_t rows = nframes - f0 < 64 ? nframes - f0 : 64;
        for (size_t r = 0; r < rows; r++) {
            const float *src = mags + (f0 + r) * mag_stride;
            for (unsigned c = lane; c < nbUsed; c += 64) s_tile[r * tileStride + c] = src[c];
        }
        __syncthreads();
        if ((size_t)lane < rows) {
            const float *row = s_tile + (size_t)lane * tileStride;
            const size_t frame = f0 + lane;
            double *cl = mfcc + frame * numCoeffs;  // accumulate in the output row
            for (unsigned i = 0; i < numCoeffs; i++) cl[i] = 0.0;
            for (unsigned f = 0; f < numFilters; f++) {
                double acc = 0.0;
                const int b0 = lo[f], b1 = hi[f];
                const double *w = Wc + off[f];
                for (int b = b0; b <= b1; b++) acc += (w[b - b0] * (double)row[b]);
                if (melraw) melraw[frame * numFilters + f] = acc;
                const double mb = log_square(acc);
                if (melbands) melbands[frame * numFilters + f] = mb;
                const double *d = dct + (size_t)f * numCoeffs;
                for (unsigned i = 0; i < numCoeffs; i++) cl[i] += (d[i] * mb);
            }
            for (unsigned i = 0; i < numCoeffs; i++) cl[i] = cl[i] / (double)numCoeffs;
        }
        __syncthreads();
    }
}

// ---- K7b: dense fp64 MFMA contraction ----------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

// One wavefront = 16 frames x nfPad filters.  A (spectrum, widened to f64): lane l holds
// A[row = l&15][k = l>>4]; B (weights): lane l holds B[k = l>>4][col = l&15]; C/D (4 f64 per
// lane): col = l&15, row = (l>>4) + 4*reg   (f64 16x16x4 layout, cdna guide section 3).
template <int NT>
__global__ __launch_bounds__(64) void mfcc_mfma_kernel(
    const float *__restrict__ mags, size_t mag_stride, size_t nframes, unsigned numBins,
    unsigned numFilters, unsigned numCoeffs, unsigned kPad, unsigned nfPad,
    const double *__restrict__ Wpad, const double *__restrict__ dct, double *__restrict__ melraw,
    double *__restrict__ melbands, double *__restrict__ mfcc) {
    __shared__ double s_mb[16 * (NT * 16 + 1)];
    const int lane = threadIdx.x;
    const int r16 = lane & 15, kq = lane >> 4;
    for (size_t f0 = (size_t)blockIdx.x * 16; f0 < nframes; f0 += (size_t)gridDim.x * 16) {
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        const size_t frame = f0 + r16;
        const float *arow = mags + (frame < nframes ? frame : nframes - 1) * mag_stride;
        for (unsigned k0 = 0; k0 < kPad; k0 += 4) {
            const double a = (k0 + kq < numBins) ? (double)arow[k0 + kq] : 0.0;
            const double *brow = Wpad + (size_t)(k0 + kq) * nfPad + r16;
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, brow[t * 16], acc[t], 0, 0, 0);
        }
        // D -> LDS [frame row][filter], then the log / DCT epilogue with one lane per frame
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) s_mb[(kq + 4 * r) * (NT * 16 + 1) + t * 16 + r16] = acc[t][r];
        __syncthreads();
        if (lane < 16 && f0 + lane < nframes) {
            const size_t fr = f0 + lane;
            double *cl = mfcc + fr * numCoeffs;
            for (unsigned i = 0; i < numCoeffs; i++) cl[i] = 0.0;
            for (unsigned f = 0; f < numFilters; f++) {
                double v = s_mb[lane * (NT * 16 + 1) + f];
                if (melraw) melraw[fr * numFilters + f] = v;
                const double mb = log_square(v);
                if (melbands) melbands[fr * numFilters + f] = mb;
                const double *d = dct + (size_t)f * numCoeffs;
                for (unsigned i = 0; i < numCoeffs; i++) cl[i] += (d[i] * mb);
            }
            for (unsigned i = 0; i < numCoeffs; i++) cl[i] = cl[i] / (double)numCoeffs;
        }
        __syncthreads();
    }
}

// ---- K7b: the dense contraction as a tiled fp64 MFMA GEMM ---------------------------------------------
// C[frames x filters] = A[frames x K] (fp32 spectra, widened) x W[K x filters] on v_mfma_f64_16x16x4_f64.
// A workgroup = 4 wavefronts, each owning a 64-frame x (NT*16)-filter tile: 4 x NT independent accumulators per
// wave (16 x NT x 4 doubles), so the matrix pipe never waits on its own result.  K is walked in tiles of 32 bins:
//   A   every wave loads ITS [64 frames x 32 bins] fp32 tile coalesced (8 lanes = the 128 B of one frame row) into
//       registers one tile ahead, parks it in its private LDS tile with row stride 34 floats (bank = 2*frame + k: the
//       fragment read `frame = lane&15, k = lane>>4` is conflict-free) and widens to f64 after the LDS read -- the
//       spectrum crosses HBM once, as fp32;
//   W   the [32 bins x NT*16] f64 weight tile is loaded once per workgroup (L2-resident: 90-200 KB in all), double-
//       buffered in LDS and shared by the four waves; row stride NT*16 doubles => the fragment read `k = lane>>4,
//       col = lane&15` is conflict-free.
// One __syncthreads() per tile (for W; the A tile is wave-private).  Per 4-bin step a lane issues 4 + NT LDS reads and
// 4 cvt for 4*NT MFMAs of 64 cycles each.  ~60 KB of LDS and <= 256 registers => TWO workgroups per CU: while one wave
// of a SIMD runs its VALU epilogue or waits for its first tile, the other keeps the matrix pipe busy (with one wave
// per SIMD the pipe idled ~60 % of the time: 1.75 ms per 1 M frames at K = 512, profiles/).
// Epilogue (banks up to 64 filters), in two halves of 32 frames so the staging tile stays small: the accumulators go to
// LDS as [frame][filter]; lane = (frame = lane&31, h = lane>>5) takes the filters / coefficients of parity h:
// log-square, then the DCT in the reference's j order (L/maxiMFCC.h:98-111).  Larger banks (mfcctest's 512/256/13) run
// in groups of 64 filters, write the raw band sums and finish in mfcc_logdct_kernel.
// FMA chains inside the MFMA round differently from the reference's mul-then-add loop => tolerance.
constexpr int kGemmFrames = 64, kGemmKT = 32, kGemmAStride = 34;

template <int NT, bool EPILOGUE>
__global__ __launch_bounds__(256, 2) void mfcc_mfma_gemm_kernel(
    const float *__restrict__ mags, size_t mag_stride, size_t nframes, unsigned numFilters, unsigned numCoeffs,
    unsigned kTiles, unsigned wStride, unsigned fOff, const double *__restrict__ Wpad, const double *__restrict__ dct,
    double *__restrict__ melraw, double *__restrict__ melbands, double *__restrict__ mfcc, unsigned rawStride) {
    extern __shared__ double s_dyn[];
    constexpr int NW = NT * 16;
    constexpr unsigned ES = NW + 1;                                         // epilogue row stride (doubles)
    double *sW = s_dyn;                                                     // [2][32][NW]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per wave: the A tile [64][34] floats, reused as the epilogue tile [32][ES] doubles
    constexpr size_t kWaveBytes = (sizeof(float) * kGemmFrames * kGemmAStride > sizeof(double) * 32 * ES)
                                      ? sizeof(float) * kGemmFrames * kGemmAStride : sizeof(double) * 32 * ES;
    char *wbase = reinterpret_cast<char *>(s_dyn + 2 * kGemmKT * NW) + (size_t)wave * ((kWaveBytes + 15) & ~(size_t)15);
    float *myA = reinterpret_cast<float *>(wbase);
    double *tile = reinterpret_cast<double *>(wbase);
    const int r16 = lane & 15, kq = lane >> 4;
    const int lrow = lane >> 3, lcol = (lane & 7) * 4;
    const size_t tilesTotal = (nframes + 4 * kGemmFrames - 1) / (4 * kGemmFrames);
    constexpr int kWPer = (16 * NW + 255) / 256;  // double2 loads per thread per weight tile (32*NW doubles / 256 threads)
    for (size_t bt = blockIdx.x; bt < tilesTotal; bt += gridDim.x) {
        const size_t F0 = (bt * 4 + wave) * kGemmFrames;  // this wave's 64 frames (may start past the end: then all clamped)
        const float *gsrc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            size_t fr = F0 + lrow + 8 * i;
            if (fr >= nframes) fr = nframes - 1;
            gsrc[i] = mags + fr * mag_stride + lcol;
        }
        d4 acc[4][NT];
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < NT; cb++) acc[rb][cb] = (d4){0.0, 0.0, 0.0, 0.0};
        float an[8][4];
        double2v wn[kWPer];
        auto loadA = [&](unsigned kt) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 v4 = *reinterpret_cast<const float4 *>(gsrc[i] + (size_t)kt * kGemmKT);
                an[i][0] = v4.x; an[i][1] = v4.y; an[i][2] = v4.z; an[i][3] = v4.w;
            }
        };
        auto loadW = [&](unsigned kt) {
#pragma unroll
            for (int j = 0; j < kWPer; j++) {
                const unsigned e = threadIdx.x + 256u * j;  // double2 index inside the tile: row = e / (NW/2)
                if (e < 16u * NW) {
                    const unsigned row = e / (NW / 2), c2 = e % (NW / 2);
                    wn[j] = *reinterpret_cast<const double2v *>(Wpad + ((size_t)kt * kGemmKT + row) * wStride + fOff + 2 * c2);
                }
            }
        };
        loadA(0);
        loadW(0);
        for (unsigned kt = 0; kt < kTiles; kt++) {
            double *bufW = sW + (kt & 1) * kGemmKT * NW;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float *d = myA + (lrow + 8 * i) * kGemmAStride + lcol;
                *reinterpret_cast<float2 *>(d) = make_float2(an[i][0], an[i][1]);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(an[i][2], an[i][3]);
            }
#pragma unroll
            for (int j = 0; j < kWPer; j++) {
                const unsigned e = threadIdx.x + 256u * j;
                if (e < 16u * NW) *reinterpret_cast<double2v *>(bufW + 2 * e) = wn[j];
            }
            if (kt + 1 < kTiles) {
                loadA(kt + 1);
                loadW(kt + 1);
            }
            __syncthreads();  // W tile kt visible; every wave has left tile kt-1 (so W buffer (kt+1)&1 may be refilled next)
#pragma unroll
            for (int ks = 0; ks < kGemmKT / 4; ks++) {
                double a[4], b[NT];
#pragma unroll
                for (int rb = 0; rb < 4; rb++) a[rb] = (double)myA[(rb * 16 + r16) * kGemmAStride + ks * 4 + kq];
#pragma unroll
                for (int cb = 0; cb < NT; cb++) b[cb] = bufW[(ks * 4 + kq) * NW + cb * 16 + r16];
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
#pragma unroll
                    for (int cb = 0; cb < NT; cb++)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
            }
            wave_lds_sync();  // this wave's A tile is consumed before the next iteration overwrites it
        }
        if constexpr (EPILOGUE) {
            const int ef = lane & 31, eh = lane >> 5;
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int rbl = 0; rbl < 2; rbl++)
#pragma unroll
                    for (int cb = 0; cb < NT; cb++)
#pragma unroll
                        for (int r = 0; r < 4; r++) tile[(rbl * 16 + kq + 4 * r) * ES + cb * 16 + r16] = acc[half * 2 + rbl][cb][r];
                wave_lds_sync();
                const size_t fr = F0 + half * 32 + ef;
                const bool live = fr < nframes;
                double *row = tile + ef * ES;
                for (unsigned f = eh; f < numFilters; f += 2) {
                    const double v = row[f];
                    const double lv = log_square(v);
                    row[f] = lv;
                    if (live) {
                        if (melraw) melraw[fr * numFilters + f] = v;
                        if (melbands) melbands[fr * numFilters + f] = lv;
                    }
                }
                wave_lds_sync();
                for (unsigned i = eh; i < numCoeffs; i += 2) {
                    double c = 0.0;
                    for (unsigned f = 0; f < numFilters; f++) c += (dct[f * numCoeffs + i] * row[f]);  // L/maxiMFCC.h:105
                    if (live) mfcc[fr * numCoeffs + i] = c / (double)numCoeffs;                         // :109
                }
                wave_lds_sync();
            }
        } else {
            // raw band sums of this filter group straight to global [frame][rawStride]
#pragma unroll
            for (int rb = 0; rb < 4; rb++)
#pragma unroll
                for (int cb = 0; cb < NT; cb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const size_t fr = F0 + rb * 16 + kq + 4 * r;
                        const unsigned f = fOff + cb * 16 + r16;
                        if (fr < nframes && f < numFilters) melraw[fr * rawStride + f] = acc[rb][cb][r];
                    }
        }
        __syncthreads();  // nobody refills W buffer 0 for the next frame tile while a slower wave still multiplies
    }
}

// log-square + DCT over raw band sums [nframes][numFilters] (banks wider than one MFMA filter group): lane = frame
__global__ __launch_bounds__(256) void mfcc_logdct_kernel(size_t nframes, unsigned numFilters, unsigned numCoeffs,
                                                          const double *__restrict__ raw, const double *__restrict__ dct,
                                                          double *__restrict__ melbands, double *__restrict__ mfcc) {
    const size_t fr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (fr >= nframes) return;
    double *out = mfcc + fr * numCoeffs;
    for (unsigned i = 0; i < numCoeffs; i++) out[i] = 0.0;
    for (unsigned f = 0; f < numFilters; f++) {
        const double lv = log_square(raw[fr * numFilters + f]);
        if (melbands) melbands[fr * numFilters + f] = lv;
        for (unsigned i = 0; i < numCoeffs; i++) out[i] += (dct[f * numCoeffs + i] * lv);
    }
    for (unsigned i = 0; i < numCoeffs; i++) out[i] = out[i] / (double)numCoeffs;
}

double hzToMel(double hz) { return 2595.0 * (log10(hz / 700.0 + 1.0)); }       // L/maxiMFCC.h:30-32
double melToHz(double mel) { return 700.0 * (pow(10, mel / 2595.0) - 1.0); }  // L/maxiMFCC.h:36-38

void free_device(mxg_mfcc_plan *p) {
    void *ptrs[] = {p->d_schedW, p->d_schedFin, p->d_lo, p->d_hi, p->d_off, p->d_Wc, p->d_dct, p->d_Wpad, p->d_fs8, p->d_fs16,
                    p->d_mmW, p->d_mmD};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    p->d_mmW = p->d_mmD = nullptr;
    p->d_schedW = nullptr;
    p->d_schedFin = nullptr;
    p->d_lo = p->d_hi = p->d_off = nullptr;
    p->d_Wc = p->d_dct = p->d_Wpad = nullptr;
    p->d_fs8 = nullptr;
    p->d_fs16 = nullptr;
}

template <typename T>
bool upload(T **dst, const std::vector<T> &src) {
    size_t n = src.empty() ? 1 : src.size();
    if (check_hip(hipMalloc(dst, sizeof(T) * n), "hipMalloc")) return false;
    if (!src.empty() &&
        check_hip(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice), "hipMemcpy"))
        return false;
    return true;
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

mxg_mfcc_plan *mxg_mfcc_plan_create(unsigned numBins, unsigned numFilters, unsigned numCoeffs,
                                    double minFreq, double maxFreq) {
    if (numBins == 0 || numFilters < 2 || numCoeffs == 0 || numBins > (1u << 20) || numFilters > 4096 ||
        numCoeffs > 4096) {
        fail(MXG_ERR_INVALID, "mxg_mfcc_plan_create: bad sizes (bins %u, filters %u, coeffs %u)", numBins,
             numFilters, numCoeffs);
        return nullptr;
    }
    mxg_mfcc_plan *p = new mxg_mfcc_plan();
    p->numBins = numBins;
    p->numFilters = numFilters;
    p->numCoeffs = numCoeffs;
    p->slots = 0;
    p->d_schedW = nullptr;
    p->d_schedFin = nullptr;
    p->d_lo = p->d_hi = p->d_off = nullptr;
    p->d_Wc = p->d_dct = p->d_Wpad = nullptr;
    p->fsSteps = 0;
    p->d_fs8 = nullptr;
    p->fs16Steps = 0;
    p->fsMinBin = 0;
    p->d_fs16 = nullptr;
    p->mmOk = 0;
    p->mmBatches = 0;
    p->d_mmW = p->d_mmD = nullptr;
    // ---- calcMelFilterBank (L/maxiMFCC.h:118-182): `sampleRate` is an unsigned int member
    const double sampleRate = (double)(unsigned int)settings().sampleRate;
    const double nyquist = sampleRate / 2;
    if (maxFreq > nyquist) maxFreq = nyquist;
    const double maxMel = hzToMel(maxFreq), minMel = hzToMel(minFreq);
    const double dMel = (maxMel - minMel) / (numFilters + 2 - 1);
    std::vector<double> filtPos(numFilters + 2);
    double mel = minMel;
    for (unsigned i = 0; i < numFilters + 2; i++) {
        filtPos[i] = melToHz(mel);
        mel += dMel;
    }
    p->h_W.assign((size_t)numFilters * numBins, 0.0);
    for (unsigned filter = 1; filter < numFilters; filter++) {
        const double thisF = filtPos[filter], nextF = filtPos[filter + 1], prevF = filtPos[filter - 1];
        for (unsigned bin = 0; bin < numBins; bin++) {
            const double binFreq = sampleRate / (double)numBins * (double)bin;  // /numBins, sic (:153)
            double wgt = 0;
            if (!(binFreq > nextF || binFreq < prevF)) {
                const double height = 2.0 / (nextF - prevF);
                if (binFreq < thisF)
                    wgt = (binFreq - prevF) * (height / (thisF - prevF));
                else
                    wgt = height + ((binFreq - thisF) * (-height / (nextF - thisF)));
            }
            p->h_W[filter + (size_t)bin * numFilters] = wgt;
        }
    }
    // ---- createDCTCoeffs (L/maxiMFCC.h:183-203)
    p->h_dct.assign((size_t)numCoeffs * numFilters, 0.0);
    {
        const double k = 3.14159265358979323846 / numFilters;
        const double w1 = 1.0 / (sqrt((double)numFilters));
        const double w2 = sqrt(2.0 / numFilters);
        for (unsigned i = 0; i < numCoeffs; i++)
            for (unsigned j = 0; j < numFilters; j++)
                p->h_dct[i + (size_t)j * numCoeffs] = (i == 0 ? w1 : w2) * cos(k * (i + 1) * (j + 0.5));
    }
    // ---- supports, compact weights
    std::vector<int> lo(numFilters), hi(numFilters), off(numFilters + 1);
    std::vector<double> Wc;
    unsigned nbUsed = 1;
    bool nonneg = true;
    for (unsigned f = 0; f < numFilters; f++) {
        int a = (int)numBins, b = -1;
        for (unsigned bin = 0; bin < numBins; bin++) {
            const double w = p->h_W[f + (size_t)bin * numFilters];
            if (w != 0.0) {
                if ((int)bin < a) a = (int)bin;
                b = (int)bin;
            }
            if (w < 0.0) nonneg = false;
        }
        lo[f] = a;
        hi[f] = b;  // b < a: empty support
        off[f] = (int)Wc.size();
        for (int bin = a; bin <= b; bin++) Wc.push_back(p->h_W[f + (size_t)bin * numFilters]);
        if (b + 1 > (int)nbUsed) nbUsed = b + 1;
    }
    off[numFilters] = (int)Wc.size();
    p->nbUsed = nbUsed;
    // ---- bin-major schedule: greedy slot assignment in filter order.  Usable when the bands
    // close in filter order (so the on-the-fly DCT sees j ascending) and <= 4 are open at once.
    std::vector<double> schedW;
    std::vector<int> schedFin;
    {
        bool ordered = nonneg;
        int lastHi = -1;
        for (unsigned f = 0; f < numFilters; f++)
            if (hi[f] >= lo[f]) {
                if (hi[f] < lastHi) ordered = false;
                lastHi = hi[f];
            }
        int S = 0;
        std::vector<int> slotOf(numFilters, -1);
        if (ordered) {
            std::vector<int> busyUntil;  // per slot: hi of the filter occupying it
            for (unsigned f = 0; f < numFilters && ordered; f++) {
                if (hi[f] < lo[f]) continue;
                int s = -1;
                for (size_t k = 0; k < busyUntil.size(); k++)
                    if (busyUntil[k] < lo[f]) {
                        s = (int)k;
                        break;
                    }
                if (s < 0) {
                    busyUntil.push_back(-1);
                    s = (int)busyUntil.size() - 1;
                }
                busyUntil[s] = hi[f];
                slotOf[f] = s;
            }
            S = (int)busyUntil.size();
            // two bands closing at the same bin must close in filter order: slots are visited in
            // slot order, so require slot order == filter order for such ties
            for (unsigned f = 0; f + 1 < numFilters && ordered; f++)
                for (unsigned g = f + 1; g < numFilters; g++)
                    if (slotOf[f] >= 0 && slotOf[g] >= 0 && hi[f] == hi[g] && slotOf[f] > slotOf[g]) ordered = false;
        }
        if (ordered && S >= 1 && S <= 4) {
            p->slots = S == 3 ? 4 : S;  // instantiated for 1, 2, 4
            const int SS = p->slots;
            schedW.assign((size_t)nbUsed * SS, 0.0);
            schedFin.assign((size_t)nbUsed * SS, -1);
            for (unsigned f = 0; f < numFilters; f++) {
                if (slotOf[f] < 0) continue;
                for (int bin = lo[f]; bin <= hi[f]; bin++)
                    schedW[(size_t)bin * SS + slotOf[f]] = p->h_W[f + (size_t)bin * numFilters];
                schedFin[(size_t)hi[f] * SS + slotOf[f]] = (int)f;
            }
        }
    }
    // ---- slot schedules of the fused kernel: longest-processing-time packing of the filters into 8 / 16 lists
    std::vector<mxg_fs_entry> fs8, fs16;
    if (numFilters <= 64 && numCoeffs <= 32 && nbUsed <= 512) {
        std::vector<unsigned> order;
        int minBin = (int)numBins;
        for (unsigned f = 0; f < numFilters; f++)
            if (hi[f] >= lo[f]) {
                order.push_back(f);
                minBin = lo[f] < minBin ? lo[f] : minBin;
            }
        std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return hi[a] - lo[a] > hi[b] - lo[b]; });
        auto pack = [&](int slots, std::vector<mxg_fs_entry> &tab) {
            std::vector<std::vector<unsigned>> lists(slots);
            std::vector<int> load(slots, 0);
            for (unsigned f : order) {
                int s = 0;
                for (int k = 1; k < slots; k++)
                    if (load[k] < load[s]) s = k;
                lists[s].push_back(f);
                load[s] += hi[f] - lo[f] + 1;
            }
            int T = 1;
            for (int s = 0; s < slots; s++) T = load[s] > T ? load[s] : T;
            T = (T + 2 * kMelBatch - 1) / (2 * kMelBatch) * (2 * kMelBatch);  // the walk alternates between two batches
            // padding steps (after a list's last filter, and the two look-ahead batches): weight 0 on a bin that is always formed
            const mxg_fs_entry pad = {0.0, (order.empty() ? 1 : minBin) * 4, 0};
            tab.assign((size_t)(T + 2 * kMelBatch) * slots, pad);
            for (int s = 0; s < slots; s++) {
                int t = 0;
                for (unsigned f : lists[s])
                    for (int bin = lo[f]; bin <= hi[f]; bin++, t++)
                        tab[(size_t)t * slots + s] = {p->h_W[f + (size_t)bin * numFilters], bin * 4, bin == hi[f] ? (int)(f + 1) * 8 : 0};
            }
            return T;
        };
        p->fsSteps = pack(kFusedSlots, fs8);
        p->fs16Steps = pack(kFusedSlots16, fs16);
        p->fsMinBin = order.empty() ? 1 : minBin;
    }
    // ---- matrix-pipe tables of the fused kernel (mxg_spectral.h: quads of filters, pairs of quads, banded K ranges)
    std::vector<double> mmW, mmD;
    if (numFilters <= 4 * 2 * kMmPairs && numCoeffs <= 4 * kMmCoefQuads && numBins == 512) {
        bool ok = true;
        int total = 0;
        for (int pr = 0; pr < kMmPairs && ok; pr++) {
            int need = 1;
            for (int gs = 0; gs < 2; gs++) {
                int a = (int)numBins, b = -1;  // the quad's band
                for (unsigned f = 4 * (2 * pr + gs); f < 4 * (2 * pr + gs) + 4 && f < numFilters; f++)
                    if (hi[f] >= lo[f]) {
                        a = lo[f] < a ? lo[f] : a;
                        b = hi[f] > b ? hi[f] : b;
                    }
                if (b < a) {  // an empty quad reads the first bins with zero weights
                    p->mmBase[pr][gs] = 0;
                    continue;
                }
                if (a < 1 || b > 255) ok = false;  // bin 0 is never formed by the half-spectrum post-pass, the tile ends at bin 256
                p->mmBase[pr][gs] = a / 4 * 4;
                const int len = b - p->mmBase[pr][gs] + 1;
                need = (len + 15) / 16 > need ? (len + 15) / 16 : need;
            }
            p->mmNb[pr] = need;
            for (int gs = 0; gs < 2; gs++)  // every lane's reads stay inside the 260-float row
                if (p->mmBase[pr][gs] + 16 * need > 260) p->mmBase[pr][gs] = (260 - 16 * need) / 4 * 4;
            for (int gs = 0; gs < 2; gs++)
                if (p->mmBase[pr][gs] < 0) ok = false;
            total += need;
        }
        if (ok) {
            p->mmBatches = total;
            mmW.assign((size_t)(total + 1) * 128, 0.0);
            int t0 = 0;
            for (int pr = 0; pr < kMmPairs; pr++) {
                const int S = 4 * p->mmNb[pr];
                for (int s4 = 0; s4 < p->mmNb[pr]; s4++)
                    for (int h = 0; h < 2; h++)
                        for (int l32 = 0; l32 < 32; l32++)
                            for (int e = 0; e < 2; e++) {
                                const int k = l32 >> 3, gs = (l32 >> 2) & 1, i = l32 & 3;
                                const unsigned f = 4 * (2 * pr + gs) + i;
                                const int bin = p->mmBase[pr][gs] + k * S + 4 * s4 + 2 * h + e;
                                double w = 0.0;
                                if (f < numFilters && bin >= 0 && bin < (int)numBins) w = p->h_W[f + (size_t)bin * numFilters];
                                mmW[(((size_t)(t0 + s4) * 2 + h) * 32 + l32) * 2 + e] = w;
                            }
                t0 += p->mmNb[pr];
            }
            // every non-zero weight must have landed in the table exactly once (a band that was shifted to stay inside the row
            // could lose its first bins): count them
            size_t nzTab = 0, nzRef = 0;
            for (double w : mmW) nzTab += w != 0.0;
            for (double w : p->h_W) nzRef += w != 0.0;
            if (nzTab != nzRef) ok = false;
            mmD.assign((size_t)kMmCoefQuads * kMmPairs * 32, 0.0);
            for (int q = 0; q < kMmCoefQuads; q++)
                for (int pr = 0; pr < kMmPairs; pr++)
                    for (int l32 = 0; l32 < 32; l32++) {
                        const int k = l32 >> 3, gs = (l32 >> 2) & 1, i = l32 & 3;
                        const unsigned coef = 4 * q + i, f = 4 * (2 * pr + gs) + k;
                        if (coef < numCoeffs && f < numFilters)
                            mmD[((size_t)q * kMmPairs + pr) * 32 + l32] = p->h_dct[coef + (size_t)f * numCoeffs];
                    }
        }
        p->mmOk = ok ? 1 : 0;
        if (ok) {
            p->h_mmW = mmW;
            p->h_mmD = mmD;
        }
    }
    p->nfPad = (numFilters + 15) / 16 * 16;
    p->kPad = (nbUsed + 3) / 4 * 4;  // rows >= numBins carry zero weights; the kernel guards the A read
    // the GEMM kernel walks K in tiles of 32 bins: rows up to the next multiple of 32 of numBins exist (zeros)
    const unsigned kRows = (numBins + 31) / 32 * 32;
    std::vector<double> dctT(p->h_dct);  // [j*numCoeffs + i] is the reference's own linear index
    std::vector<double> Wpad((size_t)kRows * p->nfPad, 0.0);
    for (unsigned bin = 0; bin < numBins; bin++)
        for (unsigned f = 0; f < numFilters; f++) Wpad[(size_t)bin * p->nfPad + f] = p->h_W[f + (size_t)bin * numFilters];
    if (ensure_init_only() || !upload(&p->d_lo, lo) || !upload(&p->d_hi, hi) || !upload(&p->d_off, off) ||
        !upload(&p->d_Wc, Wc) || !upload(&p->d_dct, dctT) || !upload(&p->d_Wpad, Wpad) ||
        !upload(&p->d_schedW, schedW) || !upload(&p->d_schedFin, schedFin) || !upload(&p->d_fs8, fs8) ||
        !upload(&p->d_fs16, fs16) || !upload(&p->d_mmW, mmW) || !upload(&p->d_mmD, mmD)) {
        // Host tables stay valid (mxg_mfcc_plan_tables works without a device); compute calls fail.
        free_device(p);
    }
    return p;
}

int mxg_mfcc_plan_destroy(mxg_mfcc_plan *p) {
    if (!p) return MXG_OK;
    free_device(p);
    delete p;
    return MXG_OK;
}

int mxg_mfcc_plan_tables(const mxg_mfcc_plan *p, double *h_melFilters, double *h_dct) {
    MXG_REQUIRE(p, "null plan");
    if (h_melFilters) memcpy(h_melFilters, p->h_W.data(), sizeof(double) * p->h_W.size());
    if (h_dct) memcpy(h_dct, p->h_dct.data(), sizeof(double) * p->h_dct.size());
    return (int)p->nbUsed;
}

int mxg_mfcc_plan_matrix_tables(const mxg_mfcc_plan *p, int *h_nb, int *h_base, double *h_W, size_t capW, double *h_D) {
    MXG_REQUIRE(p, "null plan");
    if (!p->mmOk) return 0;
    if (h_nb) memcpy(h_nb, p->mmNb, sizeof(p->mmNb));
    if (h_base) memcpy(h_base, p->mmBase, sizeof(p->mmBase));
    if (h_W) {
        MXG_REQUIRE(capW >= p->h_mmW.size(), "h_W too small (need (batches + 1) * 128 doubles)");
        memcpy(h_W, p->h_mmW.data(), sizeof(double) * p->h_mmW.size());
    }
    if (h_D) memcpy(h_D, p->h_mmD.data(), sizeof(double) * p->h_mmD.size());
    return p->mmBatches;
}

int mxg_mfcc_batch(const mxg_mfcc_plan *p, const float *d_mags, size_t mag_stride, size_t nframes,
                   double *d_melraw, double *d_melbands, double *d_mfcc, int method, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(p && d_mags && d_mfcc, "null plan / mags / mfcc");
    MXG_REQUIRE(p->d_Wc, "plan has no device tables (created without a HIP device)");
    MXG_REQUIRE(mag_stride >= p->numBins, "mag_stride < numBins");
    MXG_REQUIRE(method >= 0 && method <= 2, "method must be 0 (exact), 1 (mfma) or 2 (exact, tile kernel)");
    if (nframes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const unsigned NC = p->numCoeffs;
    const bool stream_ok = p->slots > 0 && (NC == 13 || NC == 20) &&
                           ((size_t)p->nbUsed * p->slots * 12 + (size_t)p->numFilters * NC * 8) <= 60 * 1024;
    if (method == 0 && stream_ok) {
        const int S = p->slots;
        const size_t lds = (size_t)p->nbUsed * S * (sizeof(double) + sizeof(int)) + (size_t)p->numFilters * NC * sizeof(double);
        // 16-B row loads need aligned rows and must not run past the row: nbUsed rounded up to 4 <= stride
        const int aligned16 = ((((uintptr_t)d_mags) & 15) == 0 && (mag_stride & 3) == 0 &&
                               ((p->nbUsed + 3) / 4 * 4) <= mag_stride) ? 1 : 0;
        const int block = 256;
        dim3 grid((unsigned)((nframes + block - 1) / block));
        // tiled variant: every 32-bin tile the kernel touches must lie inside the (16-B aligned) row
        const size_t tiledCols = (size_t)((p->nbUsed + kTileBins - 1) / kTileBins) * kTileBins;
        const size_t ldsT = (((size_t)p->nbUsed * S * sizeof(double) + (size_t)p->numFilters * NC * sizeof(double) +
                              (((size_t)p->nbUsed * S + 3) & ~(size_t)3) * sizeof(int)) + 15) / 16 * 16 +
                            4 * 64 * kTileStride * sizeof(float);
        if (aligned16 && tiledCols <= mag_stride && ldsT <= 64 * 1024 && tune_get("mfcc_tiled")) {
            KernelTimer kt("mfcc_stream_tiled_kernel", st);
#define MXG_TILED_LAUNCH(SS, CC)                                                                              \
    hipLaunchKernelGGL((mfcc_stream_tiled_kernel<SS, CC>), grid, dim3(block), ldsT, st, d_mags, mag_stride, nframes, \
                       p->numFilters, p->nbUsed, p->d_schedW, p->d_schedFin, p->d_lo, p->d_hi, p->d_dct, d_melraw, d_melbands, \
                       d_mfcc)
            if (NC == 13) {
                if (S == 1) MXG_TILED_LAUNCH(1, 13); else if (S == 2) MXG_TILED_LAUNCH(2, 13); else MXG_TILED_LAUNCH(4, 13);
            } else {
                if (S == 1) MXG_TILED_LAUNCH(1, 20); else if (S == 2) MXG_TILED_LAUNCH(2, 20); else MXG_TILED_LAUNCH(4, 20);
            }
#undef MXG_TILED_LAUNCH
            return check_hip(hipGetLastError(), "mfcc kernel launch");
        }
#define MXG_STREAM_LAUNCH(SS, CC)                                                                     \
    hipLaunchKernelGGL((mfcc_stream_kernel<SS, CC>), grid, dim3(block), lds, st, d_mags, mag_stride, nframes, \
                       p->numFilters, p->nbUsed, p->d_schedW, p->d_schedFin, p->d_lo, p->d_hi, p->d_dct, d_melraw, d_melbands, \
                       d_mfcc, aligned16)
        KernelTimer kt("mfcc_stream_kernel", st);
        if (NC == 13) {
            if (S == 1) MXG_STREAM_LAUNCH(1, 13); else if (S == 2) MXG_STREAM_LAUNCH(2, 13); else MXG_STREAM_LAUNCH(4, 13);
        } else {
            if (S == 1) MXG_STREAM_LAUNCH(1, 20); else if (S == 2) MXG_STREAM_LAUNCH(2, 20); else MXG_STREAM_LAUNCH(4, 20);
        }
#undef MXG_STREAM_LAUNCH
    } else if (method == 0 || method == 2) {
        unsigned tileStride = p->nbUsed | 1u;
        size_t lds = sizeof(float) * 64 * tileStride;
        MXG_REQUIRE(lds <= 64 * 1024, "filter support too wide for the LDS tile");
        size_t blocks = (nframes + 63) / 64;
        if (blocks > 256 * 4) blocks = 256 * 4;
        KernelTimer kt("mfcc_tile_kernel", st);
        hipLaunchKernelGGL(mfcc_tile_kernel, dim3((unsigned)blocks), dim3(64), lds, st, d_mags, mag_stride, nframes,
                           p->numFilters, p->numCoeffs, p->nbUsed, tileStride, p->d_lo, p->d_hi, p->d_off, p->d_Wc,
                           p->d_dct, d_melraw, d_melbands, d_mfcc);
    } else if ((((uintptr_t)d_mags) & 15) == 0 && (mag_stride & 3) == 0 && mag_stride >= (p->numBins + 31) / 32 * 32) {
        // K7b as a tiled GEMM.  K = the bins that carry weight (rounded up to 32), or all of them with mfcc_mfma_fullk.
        const unsigned kTiles = tune_get("mfcc_mfma_fullk") ? (p->numBins + 31) / 32 : (p->nbUsed + 31) / 32;
        const size_t tilesTotal = (nframes + 4 * kGemmFrames - 1) / (4 * kGemmFrames);
        const unsigned blocks = (unsigned)(tilesTotal < 512 ? tilesTotal : 512);  // two ~60 KB workgroups per CU
        auto lds_for = [](int NT, bool) {
            const size_t NW = (size_t)NT * 16;
            const size_t a = sizeof(float) * kGemmFrames * kGemmAStride, e = sizeof(double) * 32 * (NW + 1);
            return sizeof(double) * 2 * kGemmKT * NW + 4 * (((a > e ? a : e) + 15) & ~(size_t)15);
        };
#define MXG_GEMM_LAUNCH(NT, EPI, FOFF, RAW, RAWSTRIDE)                                                                  \
    do {                                                                                                                \
        const size_t lds = lds_for(NT, EPI);                                                                            \
        MXG_HIP(hipFuncSetAttribute((const void *)mfcc_mfma_gemm_kernel<NT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                                         \
        hipLaunchKernelGGL((mfcc_mfma_gemm_kernel<NT, EPI>), dim3(blocks), dim3(256), lds, st, d_mags, mag_stride, nframes,   \
                           p->numFilters, p->numCoeffs, kTiles, p->nfPad, FOFF, p->d_Wpad, p->d_dct, RAW, d_melbands, d_mfcc,  \
                           RAWSTRIDE);                                                                                  \
    } while (0)
        if (p->nfPad <= 64) {
            KernelTimer kt("mfcc_mfma_gemm_kernel", st);
            switch (p->nfPad / 16) {
                case 1: MXG_GEMM_LAUNCH(1, true, 0u, d_melraw, p->numFilters); break;
                case 2: MXG_GEMM_LAUNCH(2, true, 0u, d_melraw, p->numFilters); break;
                case 3: MXG_GEMM_LAUNCH(3, true, 0u, d_melraw, p->numFilters); break;
                default: MXG_GEMM_LAUNCH(4, true, 0u, d_melraw, p->numFilters); break;
            }
        } else {  // wide banks (mfcctest: 256 filters): groups of 64 filters, raw sums to memory, then log + DCT
            double *raw = d_melraw;
            if (!raw)
                if (int e = scratch_get(SCR_MFCC_RAW, st, sizeof(double) * nframes * p->numFilters, (void **)&raw)) return e;
            {
                KernelTimer kt("mfcc_mfma_gemm_kernel", st);
                for (unsigned fo = 0; fo < p->nfPad; fo += 64) {
                    const unsigned left = p->nfPad - fo;
                    if (left >= 64) MXG_GEMM_LAUNCH(4, false, fo, raw, p->numFilters);
                    else if (left >= 48) MXG_GEMM_LAUNCH(3, false, fo, raw, p->numFilters);
                    else if (left >= 32) MXG_GEMM_LAUNCH(2, false, fo, raw, p->numFilters);
                    else MXG_GEMM_LAUNCH(1, false, fo, raw, p->numFilters);
                }
            }
            KernelTimer kt2("mfcc_logdct_kernel", st);
            hipLaunchKernelGGL(mfcc_logdct_kernel, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, st, nframes,
                               p->numFilters, p->numCoeffs, raw, p->d_dct, d_melbands, d_mfcc);
        }
#undef MXG_GEMM_LAUNCH
    } else {  // unaligned spectra: the one-wave-per-16-frames kernel
        MXG_REQUIRE(p->nfPad <= 64, "mfma method on unaligned spectra supports up to 64 filters");
        size_t blocks = (nframes + 15) / 16;
        if (blocks > 256 * 16) blocks = 256 * 16;
#define MXG_MFMA_LAUNCH(NT)                                                                            \
    hipLaunchKernelGGL((mfcc_mfma_kernel<NT>), dim3((unsigned)blocks), dim3(64), 0, st, d_mags, mag_stride,   \
                       nframes, p->numBins, p->numFilters, p->numCoeffs, p->kPad, p->nfPad, p->d_Wpad, p->d_dct, \
                       d_melraw, d_melbands, d_mfcc)
        KernelTimer kt("mfcc_mfma_kernel", st);
        switch (p->nfPad / 16) {
            case 1: MXG_MFMA_LAUNCH(1); break;
            case 2: MXG_MFMA_LAUNCH(2); break;
            case 3: MXG_MFMA_LAUNCH(3); break;
            default: MXG_MFMA_LAUNCH(4); break;
        }
#undef MXG_MFMA_LAUNCH
    }
    return check_hip(hipGetLastError(), "mfcc kernel launch");
}

}  // extern "C"
