// mxg_spectral.h -- plans and device helpers shared by the spectral kernels: fft.hip (maxiFFT / maxiIFFT batches),
// mfcc.hip (maxiMFCC batch) and spectral.hip (the fused FFT -> magnitudes -> mel -> DCT kernel of config 4).
#pragma once
#include <math.h>

#include <vector>

#include "mxg_common.h"
#include "mxg_log.h"

struct mxg_fft_plan {
    int fftSize, hopSize, windowSize, bins, half, numBits;
    float *d_window;   // [fftSize]
    float2 *d_tw;      // stage twiddles: entry (h-1)+n = (ar0, ai0) of step n in a stage with BlockEnd h
    float2 *d_post;    // post-pass (wr, wi) for i = 1 .. half/2-1 at index i
    float2 *d_tw8;     // fftSize 1024 only, tolerance mode (knob fft_exact = 0): correctly rounded radix-8 input twiddles,
                       // [8][7] for the second register round (lane & 7) followed by [64][7] for the third (lane); see round8_t
    int round1Trivial; // d_tw[0] and d_tw[1] (the opening twiddles of stages 1 and 2) are exactly (1, 0)
};

// maxiIFFT::setup (L/maxiFFT.cpp:140-153): windowSize ? windowSize : fftSize, Hann over that, zero beyond
struct mxg_ifft_plan {
    int fftSize, hopSize, windowSize, bins, numBits;  // numBits = log2(fftSize): the inverse is a FULL-size complex FFT
    float *d_window;  // [fftSize]
    float2 *d_tw;     // inverse-direction stage twiddles, same indexing as mxg_fft_plan::d_tw
};

struct alignas(16) mxg_fs_entry {  // one step of one slot of the mel walk (16 bytes, 16-byte aligned: ONE ds_read_b128 -- as an
                                    // 8-byte-aligned struct hipcc read it with ds_read2_b64, twice the LDS cycles)
    double w;
    int off, fid;
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
    // Slot schedules of the fused FFT+MFCC kernels (spectral.hip): the filters are packed into kFusedSlots (8-wave form) and
    // kFusedSlots16 (16-wave form) lists of about equal total support length; list s is walked one bin per step, so step t of
    // slot s is the entry {weight, byte offset of the bin in a magnitude row, 8 * (filter + 1) on the last bin of a filter else 0}.
    // fsSteps / fs16Steps = the longest list rounded up to 2 * kMelBatch; shorter lists are padded AFTER their last filter, and two
    // look-ahead batches of padding rows follow (weight 0 on bin fsMinBin).  fsSteps == 0: the fused kernel does not apply.
    int fsSteps, fs16Steps;
    mxg_fs_entry *d_fs8;   // [fsSteps + 2 * kMelBatch][kFusedSlots]
    mxg_fs_entry *d_fs16;  // [fs16Steps + 2 * kMelBatch][kFusedSlots16]
    int fsMinBin;          // lowest bin any filter reads (>= 1: bin 0 is then never formed)
    // Matrix-pipe tables of the fused kernel (knob fused_mel, spectral.hip): the mel contraction and the DCT as products of
    // 4 x 4 blocks on v_mfma_f64_4x4x4_4b_f64.  The filters are cut into quads (4 consecutive filters), two quads form a pair
    // p = 0 .. kMmPairs - 1 (quads 2p and 2p + 1); a quad's weights are non-zero only inside a band of bins, so quad (p, gs)
    // contracts over the 16 * mmNb[p] bins from mmBase[p][gs] on (a multiple of 4; the pair's two bands are padded to the same
    // length with zero weights) -- ~3 x fewer multiply-adds than a dense 16-filter column block and 21 x fewer than the dense
    // product.  Operand lanes (A: lane 16 k + 4 b + i = A_b[i][k]; B: lane 16 k + 4 b + j = B_b[k][j]; D: lane 16 i + 4 b + j):
    // block b = 2 gs + fh (gs: which quad of the pair, fh: which half of the 8 frames), i = filter of the quad, j = frame of
    // the half, k = quarter of the band (lane k walks bins base + k * 4 * nb ... one by one: its magnitudes of four
    // consecutive steps are ONE 16-byte LDS read).
    //   d_mmW [batch t][half h][lane32 = 8 k + 4 gs + i][e]   weight of step 4 t' + 2 h + e (t' = batch inside its pair),
    //                                                          batches of all pairs back to back + one padding batch
    //   d_mmD [q][p][lane32 = 8 k + 4 gs + i]                 dct[coefficient 4 q + i][filter 4 (2 p + gs) + k]
    // mmOk == 0: the bank does not fit (more than 48 filters or 16 coefficients, a filter reading bin 0 or beyond bin 255).
    int mmOk, mmBatches;
    int mmNb[6], mmBase[6][2];
    double *d_mmW, *d_mmD;
    std::vector<double> h_mmW, h_mmD;  // host copies (mxg_mfcc_plan_matrix_tables: the CPU suite replays the blocks with them)
};
constexpr int kMmPairs = 6, kMmCoefQuads = 4;
constexpr int kFusedSlots = 8;
constexpr int kFusedSlots16 = 16;
constexpr int kMelBatch = 4;

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

typedef float v2f __attribute__((ext_vector_type(2)));

// TWO butterflies (L/fft.cpp:184-192 each) as ten packed instructions -- the reference's ten roundings per butterfly, none
// fused -- with the operand swizzles and sign flips done by the VOP3P modifiers instead of register moves:
//   p = (w.x*k.x, w.x*k.y)    q = (-(w.y*k.y), w.y*k.x)    t = p + q = (tr, ti)    k = j - t    j = j + t
// (a - b and a + (-b) are the same IEEE operation and the sign of a product is exact, so these are the reference's bits.)
// hipcc's own packing of bfly() spends 7-8 instructions per butterfly (both a - b and a + b are formed as packed pairs
// and one half of each is kept with a v_mov).  gfx950 needs one wait state between a packed write and a read of the
// result: inside the block the two butterflies are interleaved so no dependent pair is adjacent; the s_nop at either end
// covers whatever the compiler schedules next to the block.
#define MXG_BFLY2_BODY(WC)                                                                                                \
    v2f p1, q1, p2, q2;                                                                                                   \
    asm("s_nop 0\n\t"                                                                                                     \
        "v_pk_mul_f32 %[p1], %[w1], %[k1] op_sel_hi:[0,1]\n\t"                                                            \
        "v_pk_mul_f32 %[q1], %[w1], %[k1] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"                                  \
        "v_pk_mul_f32 %[p2], %[w2], %[k2] op_sel_hi:[0,1]\n\t"                                                            \
        "v_pk_mul_f32 %[q2], %[w2], %[k2] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"                                  \
        "v_pk_add_f32 %[p1], %[p1], %[q1]\n\t"                                                                            \
        "v_pk_add_f32 %[p2], %[p2], %[q2]\n\t"                                                                            \
        "v_pk_add_f32 %[k1], %[j1], %[p1] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                                  \
        "v_pk_add_f32 %[j1], %[j1], %[p1]\n\t"                                                                            \
        "v_pk_add_f32 %[k2], %[j2], %[p2] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                                  \
        "v_pk_add_f32 %[j2], %[j2], %[p2]\n\t"                                                                            \
        "s_nop 0"                                                                                                         \
        : [j1] "+v"(j1), [k1] "+v"(k1), [j2] "+v"(j2), [k2] "+v"(k2), [p1] "=&v"(p1), [q1] "=&v"(q1), [p2] "=&v"(p2),     \
          [q2] "=&v"(q2)                                                                                                  \
        : [w1] WC(w1), [w2] WC(w2))
__device__ __forceinline__ void bfly2(v2f &j1, v2f &k1, const v2f w1, v2f &j2, v2f &k2, const v2f w2) { MXG_BFLY2_BODY("v"); }
// the same with wave-uniform twiddles held in scalar register pairs (each instruction reads one of them: one constant-bus operand)
__device__ __forceinline__ void bfly2_s(v2f &j1, v2f &k1, const v2f w1, v2f &j2, v2f &k2, const v2f w2) { MXG_BFLY2_BODY("s"); }
#undef MXG_BFLY2_BODY

__device__ __forceinline__ v2f as_v2f(const float2 a) { return v2f{a.x, a.y}; }

// three in-register radix-2 stages over the 8 points of a lane; w0: 1 twiddle (pairs e,e+1),
// w1[2]: pairs (e,e+2) with n-offset e&1, w2[4]: pairs (e,e+4) with n-offset e&3.
// (A round as ONE 60-instruction asm block -- 2 s_nop instead of 12 -- was measured against this form on the same device,
// tools/ab_config4.sh: 1.393 / 1.389 / 1.398 ms against 1.387 / 1.378 / 1.402 ms for 1 M frames, no difference.)
__device__ __forceinline__ void round3(v2f (&x)[8], const v2f w0, const v2f (&w1)[2], const v2f (&w2)[4]) {
    bfly2(x[0], x[1], w0, x[2], x[3], w0);
    bfly2(x[4], x[5], w0, x[6], x[7], w0);
    bfly2(x[0], x[2], w1[0], x[1], x[3], w1[1]);
    bfly2(x[4], x[6], w1[0], x[5], x[7], w1[1]);
    bfly2(x[0], x[4], w2[0], x[1], x[5], w2[1]);
    bfly2(x[2], x[6], w2[2], x[3], x[7], w2[3]);
}
// round 1 of K6a: the seven twiddles are the same for every lane (scalar registers)
__device__ __forceinline__ void round3_s(v2f (&x)[8], const v2f (&w)[7]) {
    bfly2_s(x[0], x[1], w[0], x[2], x[3], w[0]);
    bfly2_s(x[4], x[5], w[0], x[6], x[7], w[0]);
    bfly2_s(x[0], x[2], w[1], x[1], x[3], w[2]);
    bfly2_s(x[4], x[6], w[1], x[5], x[7], w[2]);
    bfly2_s(x[0], x[4], w[3], x[1], x[5], w[4]);
    bfly2_s(x[2], x[6], w[5], x[3], x[7], w[6]);
}
// round 1 when w[0] and w[1] are exactly (1, 0) (mxg_fft_plan::round1Trivial): for finite k the product (1, 0) * k is k -- tr = 1*k.x -
// 0*k.y, ti = 1*k.y + 0*k.x, the zero products only decide the SIGN OF A ZERO result -- so the six butterflies of stages 1 and 2
// that open a block are an add and a subtract (2 packed instructions instead of 5).  Values downstream are the reference's except
// for signs of zeros, which the magnitudes (re^2 + im^2) do not see: used by the fused kernel only (it emits magnitudes and
// mfcc, never real / imag).  For Inf / NaN the products matter (0 * Inf = NaN: the reference turns a frame holding one such
// sample into NaN in every bin); the fused kernel detects those frames behind its square-root range test and writes the NaNs.
__device__ __forceinline__ void round3_s1(v2f (&x)[8], const v2f (&w)[7]) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const v2f t = x[e + 1];
        x[e + 1] = x[e] - t;
        x[e] = x[e] + t;
    }
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
        const v2f t = x[e + 2];
        x[e + 2] = x[e] - t;
        x[e] = x[e] + t;
    }
    bfly2_s(x[1], x[3], w[2], x[5], x[7], w[2]);
    bfly2_s(x[0], x[4], w[3], x[1], x[5], w[4]);
    bfly2_s(x[2], x[6], w[5], x[3], x[7], w[6]);
}
__device__ __forceinline__ void round3(float2 (&x)[8], const float2 w0, const float2 (&w1)[2], const float2 (&w2)[4]) {
    v2f y[8];
#pragma unroll
    for (int e = 0; e < 8; e++) y[e] = as_v2f(x[e]);
    const v2f u1[2] = {as_v2f(w1[0]), as_v2f(w1[1])};
    const v2f u2[4] = {as_v2f(w2[0]), as_v2f(w2[1]), as_v2f(w2[2]), as_v2f(w2[3])};
    round3(y, as_v2f(w0), u1, u2);
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] = make_float2(y[e].x, y[e].y);
}

// ---- tolerance mode (knob fft_exact = 0): the same three register rounds as TRUE radix-8 butterflies ------------------------
// A round of the exact kernel is three radix-2 stages, every butterfly a full complex multiply by a twiddle the reference
// generates with an fp32 recurrence (5 packed instructions each, 60 per round).  Mathematically the seven twiddles of a lane are
// w2[0]^m times eighth roots of unity, so the round is: multiply the inputs x1..x7 by T_e = e^(i m_e theta), m = 4,2,6,1,5,3,7
// (7 complex multiplies, 2 packed instructions each with an FMA), then an 8-point DFT whose only non-trivial factors are
// e^(i pi/4), e^(i 3pi/4) (one add + one multiply by 1/sqrt(2) each) and i (a free operand swizzle): 42 packed instructions per
// round, 28 in the first round (theta = 0).  The T_e are correctly rounded from double (mxg_fft_plan::d_tw8) -- closer to the true
// transform than the reference's recurrences -- and the operations are reordered and fused: NOT the reference's bits.  The
// reference's recurrences drift by ~1e-4 of a frame's peak; this kernel is within a few fp32 ulps of the TRUE transform, hence
// ~1e-4 away from the reference: both bounds are stated and tested in tests/test_gpu_spectral.py.
__device__ __forceinline__ void cmul2_t(v2f &k1, const v2f w1, v2f &k2, const v2f w2) {  // k <- k * w, two at a time
    v2f p1, p2, r1, r2;
    asm("s_nop 0\n\t"
        "v_pk_mul_f32 %[p1], %[w1], %[k1] op_sel_hi:[0,1]\n\t"                                           // (w.x*k.x, w.x*k.y)
        "v_pk_mul_f32 %[p2], %[w2], %[k2] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[r1], %[w1], %[k1], %[p1] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"   // (p.x - w.y*k.y, p.y + w.y*k.x)
        "v_pk_fma_f32 %[r2], %[w2], %[k2], %[p2] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "s_nop 0"
        : [p1] "=&v"(p1), [p2] "=&v"(p2), [r1] "=&v"(r1), [r2] "=&v"(r2)
        : [w1] "v"(w1), [k1] "v"(k1), [w2] "v"(w2), [k2] "v"(k2));
    k1 = r1;
    k2 = r2;
}
// (j, k) <- (j + i*k, j - i*k) for two pairs: i*k = (-k.y, k.x) is an operand swizzle
__device__ __forceinline__ void bft_rot2(v2f &j1, v2f &k1, v2f &j2, v2f &k2) {
    v2f a1, b1, a2, b2;
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[a1], %[j1], %[k1] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[a2], %[j2], %[k2] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[b1], %[j1], %[k1] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[b2], %[j2], %[k2] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "s_nop 0"
        : [a1] "=&v"(a1), [b1] "=&v"(b1), [a2] "=&v"(a2), [b2] "=&v"(b2)
        : [j1] "v"(j1), [k1] "v"(k1), [j2] "v"(j2), [k2] "v"(k2));
    j1 = a1; k1 = b1; j2 = a2; k2 = b2;
}
// pair 1: k1 <- e^(i pi/4) k1 = c (k.x - k.y, k.x + k.y); pair 2: k2 <- e^(i 3pi/4) k2 = c (-k.x - k.y, k.x - k.y); then (j +- k).
// c = (1/sqrt(2), .) in the low half of `c`.
__device__ __forceinline__ void bft_w8pair(v2f &j1, v2f &k1, v2f &j2, v2f &k2, const v2f c) {
    v2f t1, t2, a1, b1, a2, b2;
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[t1], %[k1], %[k1] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[t2], %[k2], %[k2] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]\n\t"
        "v_pk_mul_f32 %[t1], %[t1], %[c] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %[t2], %[t2], %[c] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %[a1], %[j1], %[t1]\n\t"
        "v_pk_add_f32 %[a2], %[j2], %[t2]\n\t"
        "v_pk_add_f32 %[b1], %[j1], %[t1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[b2], %[j2], %[t2] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "s_nop 0"
        : [t1] "=&v"(t1), [t2] "=&v"(t2), [a1] "=&v"(a1), [b1] "=&v"(b1), [a2] "=&v"(a2), [b2] "=&v"(b2)
        : [j1] "v"(j1), [k1] "v"(k1), [j2] "v"(j2), [k2] "v"(k2), [c] "v"(c));
    j1 = a1; k1 = b1; j2 = a2; k2 = b2;
}
// the 8-point DFT of a round (inputs already carry their twiddles), element pairing as round3()
__device__ __forceinline__ void radix8_t(v2f (&x)[8], const v2f c) {
    v2f y[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {  // stage A: (0,1) (2,3) (4,5) (6,7)
        y[e] = x[e] + x[e + 1];
        y[e + 1] = x[e] - x[e + 1];
    }
    // stage B: (0,2) (4,6) plain; (1,3) (5,7) with k <- i k
    x[0] = y[0] + y[2]; x[2] = y[0] - y[2];
    x[4] = y[4] + y[6]; x[6] = y[4] - y[6];
    bft_rot2(y[1], y[3], y[5], y[7]);
    x[1] = y[1]; x[3] = y[3]; x[5] = y[5]; x[7] = y[7];
    // stage C: (0,4) plain; (2,6) with k <- i k; (1,5) with k <- e^(i pi/4) k; (3,7) with k <- e^(i 3pi/4) k
    y[0] = x[0] + x[4]; y[4] = x[0] - x[4];
    v2f d0 = x[2], d1 = x[6], d2 = x[2], d3 = x[6];  // (one rot2 block on the pair twice would waste two instructions: do it by hand)
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[a], %[j], %[k] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[b], %[j], %[k] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "s_nop 0"
        : [a] "=&v"(d2), [b] "=&v"(d3)
        : [j] "v"(d0), [k] "v"(d1));
    y[2] = d2; y[6] = d3;
    bft_w8pair(x[1], x[5], x[3], x[7], c);
    x[0] = y[0]; x[4] = y[4]; x[2] = y[2]; x[6] = y[6];
}
// one tolerance-mode round: input twiddles T[0..6] for x1..x7, then the 8-point DFT
__device__ __forceinline__ void round8_t(v2f (&x)[8], const v2f (&T)[7], const v2f c) {
    cmul2_t(x[1], T[0], x[2], T[1]);
    cmul2_t(x[3], T[2], x[4], T[3]);
    cmul2_t(x[5], T[4], x[6], T[5]);
    v2f dummy = x[7];
    cmul2_t(x[7], T[6], dummy, T[6]);
    radix8_t(x, c);
}

// sqrtf.  hipcc's correctly-rounded sqrtf expands to ~25 instructions (v_sqrt_f32, the +-1 ulp residual test, and a 2^32
// pre-scale for inputs below 2^-96 whose v_sqrt_f32 result would be denormal-inaccurate).  exact_sqrtf() is Markstein's
// sequence from the hardware reciprocal square root: g = x*r, h = r/2, g + (x - g*g)*h -- the residual is exact in an fma and
// the final fma rounds ONCE, so the result is the correctly rounded root whenever the candidate g is close enough; whether
// that holds depends on the values v_rsq_f32 actually returns, so it was checked EXHAUSTIVELY on the device: all 1 879 048 192
// floats in [2^-96, FLT_MAX] give the bits of (float)sqrt((double)x) (tools/ubench/sqrt_probe.hip, profiles/r02_sqrt_probe.txt;
// so does the previous form, v_sqrt_f32 + a two-sided residual test, at twice the instructions).  Zero, inputs below 2^-96,
// Inf and NaN take the generic routine in a branch no wavefront normally enters.
__device__ __forceinline__ float exact_sqrtf(float x) {
    if (__builtin_expect(__float_as_uint(x) - 0x0F800000u >= 0x7F800000u - 0x0F800000u, 0)) return sqrtf(x);
    const float r = __builtin_amdgcn_rsqf(x);
    const float g = x * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, x);  // x - g*g, one rounding
    return __builtin_fmaf(d, h, g);
}

// exact_sqrtf of FOUR values behind ONE range test: r[] holds Markstein's result for every value; returns true when one of them
// is zero, below 2^-96, Inf or NaN -- the caller then takes sqrtf() (correctly rounded too: the same bits) for all four.
__device__ __forceinline__ bool exact_sqrtf4_try(const float (&x)[4], float (&r)[4]) {
    unsigned worst = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned u = __float_as_uint(x[i]) - 0x0F800000u;
        worst = u > worst ? u : worst;
    }
    // Markstein's sequence two values at a time (v_pk_mul_f32 / v_pk_fma_f32: each half is the IEEE operation of exact_sqrtf(),
    // the same bits): 4 reciprocal square roots + 8 packed instructions instead of 4 + 16
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const v2f xx = {x[i], x[i + 1]};
        const v2f rr = {__builtin_amdgcn_rsqf(x[i]), __builtin_amdgcn_rsqf(x[i + 1])};
        const v2f g = xx * rr, h = rr * 0.5f;
        const v2f d = __builtin_elementwise_fma(-g, g, xx);
        const v2f q = __builtin_elementwise_fma(d, h, g);
        r[i] = q.x;
        r[i + 1] = q.y;
    }
    return worst >= 0x7F800000u - 0x0F800000u;
}

// log-square of L/maxiMFCC.cpp:63
__device__ __forceinline__ double log_square(double mb) { return mb > 0.000001 ? fast_log(mb * mb) : 0.0; }

}  // namespace
}  // namespace mxg
