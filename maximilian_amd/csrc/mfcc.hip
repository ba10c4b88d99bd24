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
// K7a `mfcc_exact_kernel` (default).  The reference's melBands[f] = sum_bin W[f][bin]*spec[bin]
// runs over ALL bins in increasing order, but W is a bank of triangles: outside a filter's
// support every term is an exact +0.0 (spec >= 0), and x + (+0.0) == x.  Summing only the
// support, still in increasing bin order, is therefore bit-identical to the dense loop while
// doing ~1/45 of the work (411 non-zeros of 21 504 for 512/42).  One LANE owns one frame (64
// frames per wavefront): the sparsity pattern and the coefficients are wave-uniform (scalar
// loads), every lane is busy, and the spectra of the 64 frames are staged through LDS as a
// [64][odd stride] tile so that the per-lane row reads are bank-conflict free.  The DCT is
// accumulated on the fly in the reference's j-order (each finished band updates all
// coefficients), then divided by numCoeffs.  melraw (pre-log) is bit-exact; log() is OCML's, so
// melbands/mfcc carry the tolerance stated in DESIGN.md.
//
// K7b `mfcc_mfma_kernel` (method 1): the dense contraction frames[Nx512] x W[512x48] on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64), as the reference's own vDSP branch does with
// vDSP_mmulD (L/maxiMFCC.cpp:28-37).  FMA chains change the rounding => tolerance on melraw too.
// On gfx950 the fp64 MFMA rate equals the fp64 VALU rate, so this path is kept for dense
// (non-triangular) filter matrices and as the MFMA-utilisation measurement, not as the fast path.
#include <math.h>
#include <string.h>

#include <vector>

#include "mxg_common.h"

struct mxg_mfcc_plan {
    unsigned numBins, numFilters, numCoeffs, nbUsed;  // nbUsed = 1 + last bin with a non-zero weight
    std::vector<double> h_W;    // [numFilters + bin*numFilters]  (reference layout)
    std::vector<double> h_dct;  // [i + j*numCoeffs]
    int *d_lo, *d_hi, *d_off;   // per filter: support [lo, hi] (hi < lo = empty), offset into d_Wc
    double *d_Wc;               // compacted weights, filter-major, increasing bin
    double *d_dct;              // [j*numCoeffs + i]
    double *d_Wpad;             // dense [binsPad][48-multiple] row-major by bin, for the MFMA path
    unsigned nfPad, kPad;
};

namespace mxg {
namespace {

constexpr int kMaxCoeffs = 32;

template <int NC>
__global__ __launch_bounds__(64) void mfcc_exact_kernel(
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
            double c[NC > 0 ? NC : 1];
            double *cl = nullptr;
#pragma unroll
            for (int i = 0; i < (NC > 0 ? NC : 1); i++) c[i] = 0.0;
            if constexpr (NC == 0) {  // generic coefficient count: accumulate in the output row
                cl = mfcc + frame * numCoeffs;
                for (unsigned i = 0; i < numCoeffs; i++) cl[i] = 0.0;
            }
            for (unsigned f = 0; f < numFilters; f++) {
                double acc = 0.0;  // L/maxiMFCC.cpp:52
                const int b0 = lo[f], b1 = hi[f];
                const double *w = Wc + off[f];
                for (int b = b0; b <= b1; b++) acc += (w[b - b0] * (double)row[b]);  // :57
                if (melraw) melraw[frame * numFilters + f] = acc;
                double mb = acc > 0.000001 ? log(acc * acc) : 0.0;  // :63
                if (melbands) melbands[frame * numFilters + f] = mb;
                const double *d = dct + (size_t)f * numCoeffs;  // dctMatrix[i + j*numCoeffs], j = f
                if constexpr (NC > 0) {
#pragma unroll
                    for (int i = 0; i < NC; i++) c[i] += (d[i] * mb);  // L/maxiMFCC.h:105
                } else {
                    for (unsigned i = 0; i < numCoeffs; i++) cl[i] += (d[i] * mb);
                }
            }
            if constexpr (NC > 0) {
#pragma unroll
                for (int i = 0; i < NC; i++) mfcc[frame * NC + i] = c[i] / (double)numCoeffs;  // :109
            } else {
                for (unsigned i = 0; i < numCoeffs; i++) cl[i] = cl[i] / (double)numCoeffs;
            }
        }
        __syncthreads();
    }
}

// ---- K7b: dense fp64 MFMA contraction ----------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

// One wavefront = 16 frames x nfPad filters.  A (spectrum, widened to f64): lane l holds
// A[row = l&15][k = l>>4]; B (weights): lane l holds B[k = l>>4][col = l&15]; C/D (4 f64 per
// lane): col = l&15, row = (l>>4) + 4*reg   (f64 16x16x4 layout, cdna guide 3).
template <int NT>
__global__ __launch_bounds__(64) void mfcc_mfma_kernel(
    const float *__restrict__ mags, size_t mag_stride, size_t nframes, unsigned numBins,
    unsigned numFilters, unsigned numCoeffs, unsigned kPad, unsigned nfPad,
    const double *__restrict__ Wpad,
    const double *__restrict__ dct, double *__restrict__ melraw, double *__restrict__ melbands,
    double *__restrict__ mfcc) {
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
                double mb = v > 0.000001 ? log(v * v) : 0.0;
                if (melbands) melbands[fr * numFilters + f] = mb;
                const double *d = dct + (size_t)f * numCoeffs;
                for (unsigned i = 0; i < numCoeffs; i++) cl[i] += (d[i] * mb);
            }
            for (unsigned i = 0; i < numCoeffs; i++) cl[i] = cl[i] / (double)numCoeffs;
        }
        __syncthreads();
    }
}

double hzToMel(double hz) { return 2595.0 * (log10(hz / 700.0 + 1.0)); }       // L/maxiMFCC.h:30-32
double melToHz(double mel) { return 700.0 * (pow(10, mel / 2595.0) - 1.0); }  // L/maxiMFCC.h:36-38

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
    p->d_lo = p->d_hi = p->d_off = nullptr;
    p->d_Wc = p->d_dct = p->d_Wpad = nullptr;
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
    // ---- device images
    std::vector<int> lo(numFilters), hi(numFilters), off(numFilters + 1);
    std::vector<double> Wc;
    unsigned nbUsed = 1;
    for (unsigned f = 0; f < numFilters; f++) {
        int a = (int)numBins, b = -1;
        for (unsigned bin = 0; bin < numBins; bin++)
            if (p->h_W[f + (size_t)bin * numFilters] != 0.0) {
                if ((int)bin < a) a = (int)bin;
                b = (int)bin;
            }
        lo[f] = a;
        hi[f] = b;  // b < a: empty support
        off[f] = (int)Wc.size();
        for (int bin = a; bin <= b; bin++) Wc.push_back(p->h_W[f + (size_t)bin * numFilters]);
        if (b + 1 > (int)nbUsed) nbUsed = b + 1;
    }
    off[numFilters] = (int)Wc.size();
    if (Wc.empty()) Wc.push_back(0.0);
    p->nbUsed = nbUsed;
    p->nfPad = (numFilters + 15) / 16 * 16;
    p->kPad = (nbUsed + 3) / 4 * 4;  // rows >= numBins are zero weights; the kernel guards the A read
    std::vector<double> dctT((size_t)numFilters * numCoeffs);  // [j*numCoeffs + i] == reference index
    for (size_t x = 0; x < dctT.size(); x++) dctT[x] = p->h_dct[x];
    std::vector<double> Wpad((size_t)p->kPad * p->nfPad, 0.0);
    for (unsigned bin = 0; bin < p->kPad && bin < numBins; bin++)
        for (unsigned f = 0; f < numFilters; f++) Wpad[(size_t)bin * p->nfPad + f] = p->h_W[f + (size_t)bin * numFilters];
    if (ensure_init() ||
        check_hip(hipMalloc(&p->d_lo, sizeof(int) * numFilters), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_hi, sizeof(int) * numFilters), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_off, sizeof(int) * (numFilters + 1)), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_Wc, sizeof(double) * Wc.size()), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_dct, sizeof(double) * dctT.size()), "hipMalloc") ||
        check_hip(hipMalloc(&p->d_Wpad, sizeof(double) * Wpad.size()), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_lo, lo.data(), sizeof(int) * numFilters, hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_hi, hi.data(), sizeof(int) * numFilters, hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_off, off.data(), sizeof(int) * (numFilters + 1), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_Wc, Wc.data(), sizeof(double) * Wc.size(), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_dct, dctT.data(), sizeof(double) * dctT.size(), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_Wpad, Wpad.data(), sizeof(double) * Wpad.size(), hipMemcpyHostToDevice), "hipMemcpy")) {
        // Host tables stay valid (mxg_mfcc_plan_tables works without a device); compute calls will fail.
        if (p->d_lo) (void)hipFree(p->d_lo);
        if (p->d_hi) (void)hipFree(p->d_hi);
        if (p->d_off) (void)hipFree(p->d_off);
        if (p->d_Wc) (void)hipFree(p->d_Wc);
        if (p->d_dct) (void)hipFree(p->d_dct);
        if (p->d_Wpad) (void)hipFree(p->d_Wpad);
        p->d_lo = p->d_hi = p->d_off = nullptr;
        p->d_Wc = p->d_dct = p->d_Wpad = nullptr;
    }
    return p;
}

int mxg_mfcc_plan_destroy(mxg_mfcc_plan *p) {
    if (!p) return MXG_OK;
    if (p->d_lo) (void)hipFree(p->d_lo);
    if (p->d_hi) (void)hipFree(p->d_hi);
    if (p->d_off) (void)hipFree(p->d_off);
    if (p->d_Wc) (void)hipFree(p->d_Wc);
    if (p->d_dct) (void)hipFree(p->d_dct);
    if (p->d_Wpad) (void)hipFree(p->d_Wpad);
    delete p;
    return MXG_OK;
}

int mxg_mfcc_plan_tables(const mxg_mfcc_plan *p, double *h_melFilters, double *h_dct) {
    MXG_REQUIRE(p, "null plan");
    if (h_melFilters) memcpy(h_melFilters, p->h_W.data(), sizeof(double) * p->h_W.size());
    if (h_dct) memcpy(h_dct, p->h_dct.data(), sizeof(double) * p->h_dct.size());
    return (int)p->nbUsed;
}

int mxg_mfcc_batch(const mxg_mfcc_plan *p, const float *d_mags, size_t mag_stride, size_t nframes,
                   double *d_melraw, double *d_melbands, double *d_mfcc, int method, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(p && d_mags && d_mfcc, "null plan / mags / mfcc");
    MXG_REQUIRE(p->d_Wc, "plan has no device tables (created without a HIP device)");
    MXG_REQUIRE(mag_stride >= p->numBins, "mag_stride < numBins");
    MXG_REQUIRE(method == 0 || method == 1, "method must be 0 (exact) or 1 (mfma)");
    if (nframes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    if (method == 0) {
        unsigned tileStride = p->nbUsed | 1u;
        size_t lds = sizeof(float) * 64 * tileStride;
        MXG_REQUIRE(lds <= 160 * 1024, "filter support too wide for the LDS tile");
        size_t blocks = (nframes + 63) / 64;
        if (blocks > 256 * 4) blocks = 256 * 4;
#define MXG_MFCC_LAUNCH(NC)                                                                       \
    {                                                                                             \
        if (lds > 64 * 1024)                                                                      \
            MXG_HIP(hipFuncSetAttribute((const void *)mfcc_exact_kernel<NC>,                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
        hipLaunchKernelGGL((mfcc_exact_kernel<NC>), dim3((unsigned)blocks), dim3(64), lds, st, d_mags, \
                           mag_stride, nframes, p->numFilters, p->numCoeffs, p->nbUsed, tileStride, \
                           p->d_lo, p->d_hi, p->d_off, p->d_Wc, p->d_dct, d_melraw, d_melbands, d_mfcc); \
    }
        if (p->numCoeffs == 13)
            MXG_MFCC_LAUNCH(13)
        else
            MXG_MFCC_LAUNCH(0)
#undef MXG_MFCC_LAUNCH
    } else {
        MXG_REQUIRE(p->nfPad <= 64, "mfma method supports up to 64 filters");
        size_t blocks = (nframes + 15) / 16;
        if (blocks > 256 * 16) blocks = 256 * 16;
#define MXG_MFMA_LAUNCH(NT)                                                                          \
    hipLaunchKernelGGL((mfcc_mfma_kernel<NT>), dim3((unsigned)blocks), dim3(64), 0, st, d_mags, mag_stride, \
                       nframes, p->numBins, p->numFilters, p->numCoeffs, p->kPad, p->nfPad, p->d_Wpad, p->d_dct, \
                       d_melraw, d_melbands, d_mfcc)
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
