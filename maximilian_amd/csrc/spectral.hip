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
// Magnitudes use exact_sqrtf (mxg_spectral.h): correctly rounded for every float, checked exhaustively on the device.
#include "mxg_spectral.h"

#ifndef MXG_ABLATE
#define MXG_ABLATE 0  // timing experiments only (wrong results), bit mask: 1 no mel / log / DCT phase, 2 no butterflies, 4 no transposes, 8 no post-pass, 16 no frame loads (the prologue's frames are transformed again and again), 32 only the FIRST in-loop transpose removed
#endif
#ifndef MXG_FUSED_NTLOAD
#define MXG_FUSED_NTLOAD 1  // frame loads as non-temporal loads: every input byte is read once (same device: 1.315 -> 1.292 ms exact, 1.061 -> 1.043 ms tolerance mode)
#endif
#ifndef MXG_FUSED_PF
#define MXG_FUSED_PF 1  // frame sets in flight from HBM per wavefront: 1 = the next iteration's only; 2 = two iterations ahead (+32 VGPRs), measured the SAME time (profiles/r06_config4.md): A/B only
#endif
#ifndef MXG_FUSED_SKEW
#define MXG_FUSED_SKEW 1  // frame-by-frame phase order of the fused kernel's two frames in flight (0: both frames per phase; A/B)
#endif

namespace mxg {
namespace {

constexpr int kGroup = 8;  // frames per mel phase = 64 lanes / kFusedSlots

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

// post_lo for TWO pairs, squares of the results: q = (r.x*r.x, r.y*r.y) with r = post_lo(a, b, w) -- the same operations in
// the same order (h1 = 0.5*(a.x+b.x, a.y-b.y); h2 = (0.5*(a.y+b.y), -0.5*(a.x-b.x)); r = (h1 + w.x*h2) -/+ w.y*h2 swapped),
// 9 packed instructions per pair with the swizzles and signs in the VOP3P modifiers (see bfly2 in mxg_spectral.h; a
// source negation is exact, and -0.5*d == 0.5*(-d)).  The 0.5 is an inline constant read from the low half for both lanes.
__device__ __forceinline__ void post_lo_sq2(const v2f a1, const v2f b1, const v2f w1, const v2f a2, const v2f b2, const v2f w2,
                                            v2f &q1, v2f &q2) {
    v2f s1, t1, s2, t2, m1, n1, m2, n2;
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[s1], %[a1], %[b1] neg_hi:[0,1]\n\t"                              // (a.x+b.x, a.y-b.y)
        "v_pk_add_f32 %[t1], %[a1], %[b1] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]\n\t" // (a.y+b.y, a.x-b.x)
        "v_pk_add_f32 %[s2], %[a2], %[b2] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[t2], %[a2], %[b2] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[s1], %[s1], 0.5 op_sel_hi:[1,0]\n\t"                             // h1
        "v_pk_mul_f32 %[t1], %[t1], 0.5 op_sel_hi:[1,0] neg_hi:[1,0]\n\t"                // h2 = (h2r, h2i)
        "v_pk_mul_f32 %[s2], %[s2], 0.5 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %[t2], %[t2], 0.5 op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
        "v_pk_mul_f32 %[m1], %[w1], %[t1] op_sel_hi:[0,1]\n\t"                           // (wr*h2r, wr*h2i)
        "v_pk_mul_f32 %[n1], %[w1], %[t1] op_sel:[1,1] op_sel_hi:[1,0]\n\t"              // (wi*h2i, wi*h2r)
        "v_pk_mul_f32 %[m2], %[w2], %[t2] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[n2], %[w2], %[t2] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %[s1], %[s1], %[m1]\n\t"                                           // (h1r + wr*h2r, h1i + wr*h2i)
        "v_pk_add_f32 %[s2], %[s2], %[m2]\n\t"
        "v_pk_add_f32 %[s1], %[s1], %[n1] neg_lo:[0,1]\n\t"                              // (.. - wi*h2i, .. + wi*h2r)
        "v_pk_add_f32 %[s2], %[s2], %[n2] neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %[q1], %[s1], %[s1]\n\t"
        "v_pk_mul_f32 %[q2], %[s2], %[s2]\n\t"
        "s_nop 0"
        : [s1] "=&v"(s1), [t1] "=&v"(t1), [s2] "=&v"(s2), [t2] "=&v"(t2), [m1] "=&v"(m1), [n1] "=&v"(n1), [m2] "=&v"(m2),
          [n2] "=&v"(n2), [q1] "=&v"(q1), [q2] "=&v"(q2)
        : [a1] "v"(a1), [b1] "v"(b1), [w1] "v"(w1), [a2] "v"(a2), [b2] "v"(b2), [w2] "v"(w2));
}

// Tolerance mode: the same pair with FMAs and the halving folded into the twiddle (wh = 0.5 * w, exact):
//   s = (a.x+b.x, a.y-b.y)   t = (a.y+b.y, a.x-b.x)   r = 0.5 s + (wh.x t.x + wh.y t.y, -wh.x t.y + wh.y t.x)   q = r * r
// six packed instructions per pair instead of nine (reordered and fused: not the reference's roundings).
__device__ __forceinline__ void post_lo_sq2_t(const v2f a1, const v2f b1, const v2f w1, const v2f a2, const v2f b2, const v2f w2,
                                              v2f &q1, v2f &q2) {
    v2f s1, t1, s2, t2, m1, m2;
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[s1], %[a1], %[b1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[t1], %[a1], %[b1] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[s2], %[a2], %[b2] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[t2], %[a2], %[b2] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[m1], %[w1], %[t1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"                        // (wx tx, -wx ty)
        "v_pk_mul_f32 %[m2], %[w2], %[t2] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[m1], %[w1], %[t1], %[m1] op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"             // + (wy ty, wy tx)
        "v_pk_fma_f32 %[m2], %[w2], %[t2], %[m2] op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[s1], %[s1], 0.5, %[m1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[s2], %[s2], 0.5, %[m2] op_sel_hi:[1,0,1]\n\t"
        "v_pk_mul_f32 %[q1], %[s1], %[s1]\n\t"
        "v_pk_mul_f32 %[q2], %[s2], %[s2]\n\t"
        "s_nop 0"
        : [s1] "=&v"(s1), [t1] "=&v"(t1), [s2] "=&v"(s2), [t2] "=&v"(t2), [m1] "=&v"(m1), [m2] "=&v"(m2), [q1] "=&v"(q1),
          [q2] "=&v"(q2)
        : [a1] "v"(a1), [b1] "v"(b1), [w1] "v"(w1), [a2] "v"(a2), [b2] "v"(b2), [w2] "v"(w2));
}

// One coefficient of the DCT (L/maxiMFCC.h:98-111): c = sum_j dct[j][i] * band[j], j ascending -- the reference's sequential
// sum.  The table and band values of eight terms are requested before the first is used and the next eight before these are
// consumed: as a plain loop hipcc waits for every pair of LDS reads (one exposed LDS latency per term; for 42 filters x 13
// coefficients that was a quarter of the fused kernel's time).
__device__ __forceinline__ double dct_dot(const double *d, const unsigned dstride, const double *m, const unsigned nf) {
    constexpr int U = 8;
    double c = 0.0, dv[U], mv[U];
    const unsigned full = nf / U * U;
    if (full) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            dv[u] = d[u * dstride];
            mv[u] = m[u];
        }
    }
    for (unsigned jf = 0; jf < full; jf += U) {
        double dn[U], mn[U];
        const unsigned nxt = jf + U < full ? jf + U : jf;  // the last batch re-reads itself (unused)
#pragma unroll
        for (int u = 0; u < U; u++) {
            dn[u] = d[(nxt + u) * dstride];
            mn[u] = m[nxt + u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) c += (dv[u] * mv[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            dv[u] = dn[u];
            mv[u] = mn[u];
        }
    }
    for (unsigned jf = full; jf < nf; jf++) c += (d[jf * dstride] * m[jf]);
    return c;
}

// The same sum with the table stored coefficient-major (row i = the nf factors of coefficient i, rows and band rows 16-byte aligned):
// a term pair is one ds_read_b128 of each operand at a compile-time offset from two pointers -- as the strided form above hipcc
// spent as many instructions on addresses as on the arithmetic.  Batches of eight terms, the next batch requested before this one
// is consumed; c accumulates in j order with unfused multiply-adds, as the reference does.
typedef double d2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double dct_dot_rows(const double *d, const double *m, const unsigned nf) {
    constexpr int U = 4;  // pairs per batch
    const d2f *dp = reinterpret_cast<const d2f *>(d), *mp = reinterpret_cast<const d2f *>(m);
    const unsigned pairs = nf / 2, full = pairs / U * U;
    double c = 0.0;
    d2f dv[U], mv[U];
    if (full) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            dv[u] = dp[u];
            mv[u] = mp[u];
        }
    }
    for (unsigned pf = 0; pf < full; pf += U) {
        d2f dn[U], mn[U];
        const unsigned nxt = pf + U < full ? pf + U : pf;  // the last batch re-reads itself (unused)
#pragma unroll
        for (int u = 0; u < U; u++) {
            dn[u] = dp[nxt + u];
            mn[u] = mp[nxt + u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            c += (dv[u].x * mv[u].x);
            c += (dv[u].y * mv[u].y);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            dv[u] = dn[u];
            mv[u] = mn[u];
        }
    }
    for (unsigned j = 2 * full; j < nf; j++) c += (d[j] * m[j]);
    return c;
}

// The mel walk of one wavefront: lane = (frame of the group, slot); the slot's filter list is walked one table entry per step,
// so every band sum is the reference's sequential sum over the filter's support in increasing bin order (L/maxiMFCC.cpp:52-60;
// terms outside the support are exact +0.0 in the reference: bit-identical, see mfcc.hip).  An entry is one ds_read_b128, a
// magnitude one LDS gather; the loop is software-pipelined two batches deep (entries two batches ahead, gathers one) so no
// LDS latency sits between the dependent fp64 adds.  `fs` points at the slot's column of a table with `steps` rows
// (a multiple of 2 * kMelBatch) + two batches of padding rows.
typedef int i4e __attribute__((ext_vector_type(4)));
template <int SLOTS>
__device__ __forceinline__ void mel_walk(const char *Mrow, double *melrow, const mxg_fs_entry *fs, const int steps) {
    double acc = 0.0;  // L/maxiMFCC.cpp:52
    // Two entry batches P, Q and their gathered magnitudes, used alternately (`steps` is a multiple of 2 * kMelBatch): while a
    // batch is consumed, the other batch's magnitudes are gathered and each consumed entry's registers are refilled with the
    // entry two batches on -- every LDS result has a whole batch of dependent fp64 adds to arrive, and no register is copied
    // (as a rotating three-buffer pipeline hipcc spent 24 of 81 instructions per 8 steps on v_mov).  An entry is fetched as ONE
    // 16-byte vector (x, y = the weight's bits, z = byte offset of the bin, w = 8 * (filter + 1) on a filter's last bin else 0):
    // read member by member hipcc uses ds_read2_b64, twice the LDS cycles of ds_read_b128.
    const i4e *tab = reinterpret_cast<const i4e *>(fs);
    i4e P[kMelBatch], Q[kMelBatch];
    float xP[kMelBatch], xQ[kMelBatch];
#pragma unroll
    for (int i = 0; i < kMelBatch; i++) P[i] = tab[i * SLOTS];
#pragma unroll
    for (int i = 0; i < kMelBatch; i++) Q[i] = tab[(kMelBatch + i) * SLOTS];
#pragma unroll
    for (int i = 0; i < kMelBatch; i++) xP[i] = *reinterpret_cast<const float *>(Mrow + P[i].z);
    auto consume = [&](const i4e &e, const float x) {
        acc += (__hiloint2double(e.y, e.x) * (double)x);  // L/maxiMFCC.cpp:57
        if (e.w) {  // the entry closes filter e.w / 8 - 1: the byte offset of the slot AFTER its band sum
            *reinterpret_cast<double *>(reinterpret_cast<char *>(melrow) + e.w - 8) = acc;
            acc = 0.0;
        }
    };
    for (int t0 = 0; t0 < steps; t0 += 2 * kMelBatch) {
#pragma unroll
        for (int i = 0; i < kMelBatch; i++) {
            xQ[i] = *reinterpret_cast<const float *>(Mrow + Q[i].z);
            consume(P[i], xP[i]);
            P[i] = tab[(t0 + 2 * kMelBatch + i) * SLOTS];
        }
#pragma unroll
        for (int i = 0; i < kMelBatch; i++) {
            xP[i] = *reinterpret_cast<const float *>(Mrow + P[i].z);
            consume(Q[i], xQ[i]);
            Q[i] = tab[(t0 + 3 * kMelBatch + i) * SLOTS];
        }
    }
}

// ---- tolerance mode: the mel / log / DCT stage in fp32 --------------------------------------------------------------------
// (band sums of <= 40 terms and 42-term DCT sums in fp32 with FMAs: ~1e-6 relative on the sums, ~1e-5 absolute on a coefficient --
// an order below the 1.3e-4 by which the reference's own transform misses the true one; stated with the mode's other tolerances)
struct alignas(16) fs32_entry {  // the slot table re-staged for fp32: 16 bytes, one ds_read_b128
    float w;
    int off, fid, pad;  // fid: 4 * (filter + 1) on the last bin of a filter, else 0
};
typedef int i4v __attribute__((ext_vector_type(4)));
template <int SLOTS>
__device__ __forceinline__ void mel_walk_t(const char *Mrow, float *melrow, const fs32_entry *fs, const int steps) {
    float acc = 0.0f;
    // (an entry is fetched as ONE 16-byte vector: read member by member hipcc fetches the three used words with ds_read_b96,
    // which takes twice the LDS cycles of ds_read_b128)
    const i4v *tab = reinterpret_cast<const i4v *>(fs);
    i4v P[kMelBatch], Q[kMelBatch];  // x = weight bits, y = byte offset of the bin, z = 4 * (filter + 1) on a filter's last bin else 0
    float xP[kMelBatch], xQ[kMelBatch];
#pragma unroll
    for (int i = 0; i < kMelBatch; i++) P[i] = tab[i * SLOTS];
#pragma unroll
    for (int i = 0; i < kMelBatch; i++) Q[i] = tab[(kMelBatch + i) * SLOTS];
#pragma unroll
    for (int i = 0; i < kMelBatch; i++) xP[i] = *reinterpret_cast<const float *>(Mrow + P[i].y);
    auto consume = [&](const i4v &e, const float x) {
        acc = __builtin_fmaf(__int_as_float(e.x), x, acc);
        if (e.z) {
            *reinterpret_cast<float *>(reinterpret_cast<char *>(melrow) + e.z - 4) = acc;
            acc = 0.0f;
        }
    };
    for (int t0 = 0; t0 < steps; t0 += 2 * kMelBatch) {
#pragma unroll
        for (int i = 0; i < kMelBatch; i++) {
            xQ[i] = *reinterpret_cast<const float *>(Mrow + Q[i].y);
            consume(P[i], xP[i]);
            P[i] = tab[(t0 + 2 * kMelBatch + i) * SLOTS];
        }
#pragma unroll
        for (int i = 0; i < kMelBatch; i++) {
            xP[i] = *reinterpret_cast<const float *>(Mrow + P[i].y);
            consume(Q[i], xQ[i]);
            Q[i] = tab[(t0 + 3 * kMelBatch + i) * SLOTS];
        }
    }
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dct_dot_rows_f(const float *d, const float *m, const unsigned nf) {
    const f4v *dp = reinterpret_cast<const f4v *>(d), *mp = reinterpret_cast<const f4v *>(m);
    const unsigned octs = (nf + 7) / 8;  // (both rows are padded with zeros to a multiple of 8 floats)
    float c = 0.0f;
    for (unsigned o = 0; o < octs; o++) {
        const f4v d0 = dp[2 * o], m0 = mp[2 * o], d1 = dp[2 * o + 1], m1 = mp[2 * o + 1];
        c = __builtin_fmaf(d0.x, m0.x, c); c = __builtin_fmaf(d0.y, m0.y, c);
        c = __builtin_fmaf(d0.z, m0.z, c); c = __builtin_fmaf(d0.w, m0.w, c);
        c = __builtin_fmaf(d1.x, m1.x, c); c = __builtin_fmaf(d1.y, m1.y, c);
        c = __builtin_fmaf(d1.z, m1.z, c); c = __builtin_fmaf(d1.w, m1.w, c);
    }
    return c;
}

// ---- the mel contraction and the DCT on the matrix pipe (knob fused_mel; tables: mxg_mfcc_plan::d_mmW / d_mmD) ------------------
// v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 x 4 blocks: A lane 16 k + 4 b + i = A_b[i][k], B lane 16 k + 4 b + j =
// B_b[k][j], D lane 16 i + 4 b + j = D_b[i][j] (tools/ubench/mfma_probe.hip, profiles/r02_ubench.md).  For a group of 8 frames
// block b = 2 gs + fh multiplies the weights of quad gs of a filter pair (A_b[i][k]: filter i of the quad, bin of quarter k) by
// the magnitudes of the frames 4 fh .. 4 fh + 3 (B_b[k][j]); four instructions consume one 16-byte magnitude read and two 16-byte
// weight reads per lane.  Sixteen clocks of the matrix pipe per instruction, no vector ALU work except the fp32 -> fp64
// conversions.  The band sums leave in D layout (lane = filter-of-quad i, block, frame-of-half j), which is exactly the B layout
// of the DCT's instructions (k = filter of the quad): the logs are taken in place and the DCT is 24 more instructions
// (4 coefficient quads x 6 pairs) followed by ONE cross-lane add (the two quads of a pair sit 8 lanes apart).  Sums are fused
// multiply-adds in the matrix pipe's order: not the reference's sequential sums -- within 1e-13 of a frame's largest band.
struct MmLane {
    int k, gs, fh, ij, lane32;
};
__device__ __forceinline__ MmLane mm_lane(const int lane) {
    MmLane L;
    L.k = lane >> 4; L.gs = (lane >> 3) & 1; L.fh = (lane >> 2) & 1; L.ij = lane & 3;
    L.lane32 = 8 * L.k + 4 * L.gs + L.ij;
    return L;
}
typedef float f4e __attribute__((ext_vector_type(4)));
struct MmArgs {
    int nb[kMmPairs], base[kMmPairs][2];
};
__device__ __forceinline__ void mel_mfma(const float *M, const unsigned mstride, const double *s_mmW, const MmArgs &mm,
                                         const MmLane &L, double (&acc)[kMmPairs]) {
    const float *mrow = M + (4 * L.fh + L.ij) * mstride;
    const d2f *wq = reinterpret_cast<const d2f *>(s_mmW) + L.lane32;  // [(batch * 2 + half) * 32 + lane32]: 64 per batch, pairs back to back
    const f4e *mp0[kMmPairs + 1];
#pragma unroll
    for (int p = 0; p < kMmPairs; p++)
        mp0[p] = reinterpret_cast<const f4e *>(mrow + (L.gs ? mm.base[p][1] : mm.base[p][0]) + L.k * 4 * mm.nb[p]);
    mp0[kMmPairs] = mp0[kMmPairs - 1];  // (what the last batch "prefetches": its own pair's start; the table has a padding batch)
    // one batch ahead: the three 16-byte reads of batch t + 1 are requested before the four matrix instructions of batch t issue
    // (left to itself hipcc puts the reads at the top of the loop body and waits for them at once: one exposed LDS latency per batch)
    f4e f = mp0[0][0];
    d2f w0 = wq[0], w1 = wq[32];
#pragma unroll
    for (int p = 0; p < kMmPairs; p++) {
        const int nb = mm.nb[p];
        double a = 0.0;
        for (int s4 = 0; s4 < nb; s4++) {
            const f4e *np = s4 + 1 < nb ? mp0[p] + (s4 + 1) : mp0[p + 1];
            const f4e fn = *np;
            const d2f w0n = wq[64], w1n = wq[96];
            __builtin_amdgcn_sched_barrier(0);
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(w0.x, (double)f.x, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(w0.y, (double)f.y, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(w1.x, (double)f.z, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(w1.y, (double)f.w, a, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f = fn; w0 = w0n; w1 = w1n;
            wq += 64;
        }
        acc[p] = a;
    }
}
// log-square in place, then the DCT; C[q] of the lanes with gs == 0 = coefficient 4 q + (lane >> 4) of frame 4 fh + ij, not yet
// divided by numCoeffs
__device__ __forceinline__ void dct_mfma(const double (&lg)[kMmPairs], const double *s_mmD, const MmLane &L,
                                         double (&C)[kMmCoefQuads]) {
    const double *dq = s_mmD + L.lane32;
#pragma unroll
    for (int q = 0; q < kMmCoefQuads; q++) {
        double dv[kMmPairs];
#pragma unroll
        for (int p = 0; p < kMmPairs; p++) dv[p] = dq[(q * kMmPairs + p) * 32];
        double c = 0.0;
#pragma unroll
        for (int p = 0; p < kMmPairs; p++) c = __builtin_amdgcn_mfma_f64_4x4x4f64(dv[p], lg[p], c, 0, 0, 0);
        // the other quad of the pair: 8 lanes on inside the row of 16 (row_ror:8)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(c), 0x128, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(c), 0x128, 0xf, 0xf, false);
        C[q] = c + __hiloint2double(hi, lo);
    }
}

struct FusedArgs {
    const float *signal;
    size_t frame_stride, nframes;
    const float *window;
    const float2 *tw, *post;
    const float2 *tw8;  // tolerance mode: radix-8 input twiddles ([8][7] by lane & 7, then [64][7] by lane; mxg_fft_plan::d_tw8)
    unsigned numFilters, numCoeffs, nbUsed, mstride, nfp, dctPad;
    unsigned nfpf;  // tolerance mode: row stride (floats) of the fp32 band rows / DCT rows: >= numFilters rounded up to 8, nfpf / 4 odd
    int steps;
    int edgeBins;  // the bank reads bin 0 or bin 256 (or the magnitudes are written out): form them
    int mUncond;   // every lane may store its four magnitudes (bins 1 + lane + 64 q <= 256) without a test: the row is long enough, or
                   // `mslack` floats follow the tile -- a row's overhang lands on the NEXT row's first bins, which that frame's own
                   // post-pass rewrites afterwards (the DS unit executes a wavefront's stores in order), the last row's on the slack
    unsigned mslack;
    const mxg_fs_entry *fs;
    const double *dct;
    float *mags;
    double *melraw, *melbands, *mfcc;
    const double *mmW, *mmD;  // MEL >= 1: the matrix-pipe tables (mxg_mfcc_plan::d_mmW / d_mmD)
    int mmBatches;
    MmArgs mm;
};

// FULL: magnitudes of all 512 bins are needed (written out, or the bank reaches beyond bin 256)
//
// Two frames are in flight per wavefront (A, B: two X images in LDS, two register sets): between two LDS round trips the
// kernel issues the same phase of both frames, so frame B's transpose latency is covered by frame A's butterflies and vice
// versa -- at two waves per SIMD nothing else would cover it (measured: 34 % of wave cycles waiting on a counter, LDS pipe
// 61 % busy with the twiddles still in LDS).  The 21 stage twiddles a lane needs never change: rounds 2 and 3 hold theirs in
// 28 VGPRs, round 1's seven are wave-uniform and live in SGPRs; LDS carries only the transposes, the magnitude rows and the
// mel tables.
//
// TOL (knob fft_exact = 0, tolerance mode): the three register rounds are true radix-8 butterflies with correctly rounded input
// twiddles and FMAs (round8_t, mxg_spectral.h: 112 packed instructions per frame instead of 180), the magnitudes take the
// hardware square root (v_sqrt_f32, <= 1 ulp) and log2 instead of the correctly rounded sequences.  Not the reference's bits: within
// 6e-7 of a frame's peak of the TRUE transform, and -- the reference's fp32 twiddle recurrences drift by ~1e-4 -- within 4e-4 of the
// reference's magnitudes, 5e-4 of its mfcc (tests/test_gpu_spectral.py); everything else as in the exact kernel.
// MODE: 0 exact; 1 exact with the opening twiddles of stages 1 and 2 known to be (1, 0) (round3_s1); 2 tolerance mode.
// NF, WAVES: the layout.  NF = 2, WAVES = 4 is the two-frames-in-flight form above (two workgroups per CU: 2 wavefronts per SIMD,
// <= 256 VGPRs).  NF = 1, WAVES = 12 keeps ONE frame in flight per wavefront and one 768-thread workgroup per CU: 3 wavefronts per
// SIMD (<= 168 VGPRs; one X image instead of two is what lets twelve 8-frame magnitude tiles fit the 160 KB) -- the third
// wavefront covers the transposes' latency instead of the second frame, and the VALU issues faster with three to pick from.
// MEL (knob fused_mel): 0 = the sparse mel walk, logs one band per lane, DCT one coefficient per lane (vector ALU, the reference's
// summation orders); 2 = the same walk (band sums bit-exact), then the logs in matrix layout and the DCT on the matrix pipe
// (dct_mfma: the DCT's 42-term sums are fused multiply-adds -- they only ever saw the device log's values, the mfcc tolerance is
// unchanged); 1 = the mel contraction on the matrix pipe as well (mel_mfma: band sums within 1e-13 of the largest band).
// MEL >= 1 runs as ONE 8-wave workgroup per CU (the tables once per CU) on rows of 260 floats (bins 0 .. 256, bin 0 kept at 0).
// (no-load-store-opt: see the Makefile -- the transposes want plain 8-byte LDS accesses, not hipcc's fused ds_read2 / ds_write2 forms)
template <bool FULL, bool WRITE_MAGS, bool ALIGNED8, int MODE, int NF, int WAVES, int MEL = 0>
__global__ __launch_bounds__(64 * WAVES, NF == 1 ? 1 : 2) __attribute__((target("no-load-store-opt"))) void fft_mfcc_kernel(const FusedArgs A) {
    constexpr bool TOL = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    // [fs (steps + 2 batches) * 8 entries | mmW][dct NF*NC f64, padded to 16 B | mmD] | per wave: XA (= band rows), XB, M
    mxg_fs_entry *s_fs = reinterpret_cast<mxg_fs_entry *>(s_dyn);
    const int fsRows = A.steps + 2 * kMelBatch;
    double *s_mmW = reinterpret_cast<double *>(s_dyn);
    double *s_d = MEL == 1 ? s_mmW + (size_t)(A.mmBatches + 1) * 128 : reinterpret_cast<double *>(s_fs + (size_t)fsRows * kFusedSlots);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // SHARED (the matrix-pipe forms): the NF frames in flight of a wavefront use ONE X image.  The phases are issued frame by frame
    // -- frame f's transpose store, its transposed reads, THEN frame f + 1's store -- and the DS unit executes a wavefront's
    // operations in order, so f's reads have been performed when f + 1's stores land on the same addresses: no wait, the LDS that
    // the second image took is what lets more frames be in flight (NF = 4: eight per SIMD instead of four).
    constexpr bool SHARED = MEL >= 1;
    static_assert(!SHARED || MXG_FUSED_SKEW, "the shared X image needs the frame-by-frame phase order");
    constexpr int kImages = SHARED ? 1 : NF;
    const size_t perWaveBytes = kImages * sizeof(float2) * kX1024 + sizeof(float) * (kGroup * A.mstride + A.mslack);
    v2f *s_twl = reinterpret_cast<v2f *>(s_d + (MEL >= 1 ? kMmCoefQuads * kMmPairs * 32 : A.dctPad));  // NF == 1: [8][7] round-2 twiddles by lane & 7, then [64][7] round-3 by lane
    constexpr int kTwl = NF == 1 ? (8 + 64) * 7 : 0;
    char *wbase = reinterpret_cast<char *>(s_twl + kTwl) + (size_t)wave * perWaveBytes;
    v2f *X[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) X[f] = reinterpret_cast<v2f *>(wbase) + (SHARED ? 0 : f) * kX1024;
    double *s_mel = reinterpret_cast<double *>(wbase);  // the band rows live on X[0] between the last post-pass and the next frame
    float *M = reinterpret_cast<float *>(wbase + kImages * sizeof(float2) * kX1024);
    fs32_entry *s_fs32 = reinterpret_cast<fs32_entry *>(s_dyn);  // tolerance mode: the same tables in fp32, in the same place
    float *s_df = reinterpret_cast<float *>(s_d);
    if constexpr (MEL >= 1) {
        if constexpr (MEL == 1) {
            for (int i = threadIdx.x; i < (A.mmBatches + 1) * 128; i += blockDim.x) s_mmW[i] = A.mmW[i];
        } else {
            for (int i = threadIdx.x; i < fsRows * kFusedSlots; i += blockDim.x) s_fs[i] = A.fs[i];
        }
        for (int i = threadIdx.x; i < kMmCoefQuads * kMmPairs * 32; i += blockDim.x) s_d[i] = A.mmD[i];
        if (lane < kGroup) M[lane * A.mstride] = 0.0f;  // bin 0 of every row: never formed, read with zero weights
        // ... and the row's tail behind bin 256 (floats 257 .. mstride - 1), which nobody stores: a band moved down to end at the row's
        // last float (mfcc.hip: base + 16 nb <= 260) reads it with zero weights, and 0 x (whatever bit pattern the LDS held) may be NaN
        // (reachable with e.g. setup(512, 46, 13, 20, 21900))
        if (lane < kGroup * 3) {
            const unsigned tail = 257u + (unsigned)(lane / kGroup);
            if (tail < A.mstride) M[(lane % kGroup) * A.mstride + tail] = 0.0f;
        }
    } else if constexpr (TOL) {
        for (int i = threadIdx.x; i < fsRows * kFusedSlots; i += blockDim.x) {
            const mxg_fs_entry e = A.fs[i];
            s_fs32[i] = fs32_entry{(float)e.w, e.off, e.fid / 2, 0};
        }
        for (unsigned t = threadIdx.x; t < A.numCoeffs * A.nfpf; t += blockDim.x) s_df[t] = 0.0f;
        __syncthreads();
        for (unsigned t = threadIdx.x; t < A.numFilters * A.numCoeffs; t += blockDim.x) {
            const unsigned j = t / A.numCoeffs, i = t - j * A.numCoeffs;
            s_df[i * A.nfpf + j] = (float)A.dct[t];
        }
    } else {
        for (int i = threadIdx.x; i < fsRows * kFusedSlots; i += blockDim.x) s_fs[i] = A.fs[i];
        for (unsigned t = threadIdx.x; t < A.numFilters * A.numCoeffs; t += blockDim.x) {  // A.dct is [j][i]; LDS rows are [i][nfp]
            const unsigned j = t / A.numCoeffs, i = t - j * A.numCoeffs;
            s_d[i * A.nfp + j] = A.dct[t];
        }
    }
    if constexpr (NF == 1) {
        const int bi[7] = {7, 15, 23, 31, 39, 47, 55}, ci[7] = {63, 127, 191, 255, 319, 383, 447};
        for (int t = threadIdx.x; t < kTwl; t += blockDim.x) {
            const int row = t / 7, i = t - row * 7;  // rows 0..7: round 2 (lane & 7 = row); rows 8..71: round 3 (lane = row - 8)
            s_twl[t] = TOL ? as_v2f(A.tw8[t]) : as_v2f(row < 8 ? A.tw[bi[i] + row] : A.tw[ci[i] + row - 8]);
        }
    }
    __syncthreads();

    const int lo = lane & 7, hi = lane >> 3;
    const int rev6 = (int)(__brev((unsigned)lane) >> 26);
    // element e of a lane is sample pair 64 * rev3(e) + rev6(lane) of the frame: one per-lane offset + a compile-time one
    constexpr int kRev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    // stage twiddles (L/fft.cpp:161-182 replayed on the host): round 1 wave-uniform (scalar registers), rounds 2 / 3 per lane,
    // loaded once and pinned so that they are not re-loaded per frame.  (Re-loading them from global memory at the top of every
    // group of 8 frames, to have them dead during the mel / log / DCT phase, was measured: the exposed L2 latency costs 6 %.)
    // The 12-wave layout (168 VGPRs) keeps a copy of the round 2 / 3 twiddles in LDS instead and re-reads THAT per group
    // (14 ds_read_b64): 28 registers free while the logs and the DCT run, which is where hipcc spilled.
    v2f wv[8], ta[7], tb[7], tc[7], pw[4];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float2 t = A.tw[i];
        ta[i].x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t.x)));
        ta[i].y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t.y)));
    }
    auto load_tables = [&]() {
        const float *win = A.window + 2 * rev6;
        const float2 *tw = A.tw, *tw8 = A.tw8, *post = A.post;
#pragma unroll
        for (int e = 0; e < 8; e++) wv[e] = v2f{win[128 * kRev3[e]], win[128 * kRev3[e] + 1]};
        const int bi[7] = {7, 15, 23, 31, 39, 47, 55}, ci[7] = {63, 127, 191, 255, 319, 383, 447};
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if constexpr (TOL) {  // the input twiddles T_1..T_7 of the second / third round
                tb[i] = as_v2f(tw8[lo * 7 + i]);
                tc[i] = as_v2f(tw8[56 + lane * 7 + i]);
            } else {
                tb[i] = as_v2f(tw[bi[i] + lo]);
                tc[i] = as_v2f(tw[ci[i] + lane]);
            }
        }
        // post-pass: twiddles of this lane's four pairs; LDS slots of the pairs (i, 512 - i), i = 1 + lane + 64q, are one base each
        // plus a compile-time offset: pad8(i0 + 64q) = pad8(i0) + 72q
#pragma unroll
        for (int q = 0; q < 4; q++) pw[q] = as_v2f(post[(1 + lane + 64 * q) < 256 ? 1 + lane + 64 * q : 255]);
        if constexpr (TOL && !FULL) {  // post_lo_sq2_t takes half the twiddle
#pragma unroll
            for (int q = 0; q < 4; q++) pw[q] = pw[q] * 0.5f;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) asm volatile("" : "+v"(wv[e]));
#pragma unroll
        for (int i = 0; i < 7; i++) asm volatile("" : "+v"(tb[i]), "+v"(tc[i]));
#pragma unroll
        for (int q = 0; q < 4; q++) asm volatile("" : "+v"(pw[q]));
    };
    const v2f c8 = {0.70710678118654752440f, 0.70710678118654752440f};
    auto msqrt = [](float x) { return TOL ? __builtin_amdgcn_sqrtf(x) : exact_sqrtf(x); };
    const int pa0 = pad8(1 + lane), pb0 = pad8(511 - lane);
    const int zidx = lane == 0 ? 0 : 256;  // lanes 0 / 63 also own bin 0 / the middle bin
    const size_t nframes = A.nframes;
    auto load_frame = [&](size_t fr, v2f (&dst)[8]) {
        const unsigned fu = __builtin_amdgcn_readfirstlane((unsigned)(fr < nframes ? fr : nframes - 1));
        const float *x = A.signal + (size_t)fu * A.frame_stride + 2 * rev6;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if constexpr (ALIGNED8) {
#if MXG_FUSED_NTLOAD
                dst[e] = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(x + 128 * kRev3[e]));
#else
                dst[e] = *reinterpret_cast<const v2f *>(x + 128 * kRev3[e]);
#endif
            } else {
                dst[e].x = x[128 * kRev3[e]];
                dst[e].y = x[128 * kRev3[e] + 1];
            }
        }
    };
    // real split post-pass (L/fft.cpp:245-275) + magnitudes (cartToPol :510-511) of one frame into its row of the tile
    // bin 0 packs DC and Nyquist (L/fft.cpp:274-275); bin 256 passes through untouched
    auto post_edge = [&](const v2f *X, const int j, const size_t f0) {
        float *Mrow = M + j * A.mstride;
        const v2f z = X[pad8(zidx)];
        const float zr = lane == 0 ? z.x + z.y : z.x, zi = lane == 0 ? z.x - z.y : z.y;
        const float mz = msqrt(zr * zr + zi * zi);
        if (lane == 0 || lane == 63) {
            Mrow[zidx] = mz;
            if constexpr (WRITE_MAGS)
                if (f0 + (size_t)j < nframes)
                    A.mags[(size_t)__builtin_amdgcn_readfirstlane((unsigned)(f0 + j)) * 512 + zidx] = mz;
        }
    };
    // the full post-pass (both halves of every pair) + magnitudes of one frame, for the launches that need all 512 bins
    auto post_frame = [&](const v2f *X, const int j, const size_t f0) {
        float *Mrow = M + j * A.mstride;
        const bool frame_live = f0 + (size_t)j < nframes;  // wave-uniform
        float *grow = nullptr;
        if constexpr (WRITE_MAGS)
            grow = A.mags + (size_t)__builtin_amdgcn_readfirstlane((unsigned)(f0 + j < nframes ? f0 + j : nframes - 1)) * 512;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const v2f xa = X[pa0 + 72 * q], xb = X[pb0 - 72 * q];
            float2 a = make_float2(xa.x, xa.y), b = make_float2(xb.x, xb.y);
            post_pair(a, b, make_float2(pw[q].x, pw[q].y));
            const float ma = msqrt(a.x * a.x + a.y * a.y), mb = msqrt(b.x * b.x + b.y * b.y);
            if (q < 3 || lane < 63) {
                Mrow[1 + lane + 64 * q] = ma;
                Mrow[511 - lane - 64 * q] = mb;
                if constexpr (WRITE_MAGS)
                    if (frame_live) {
                        grow[1 + lane + 64 * q] = ma;
                        grow[511 - lane - 64 * q] = mb;
                    }
            }
        }
        post_edge(X, j, f0);
    };
    const size_t ngroups = (nframes + kGroup - 1) / kGroup;
    const size_t gstep = (size_t)gridDim.x * WAVES;
    const size_t g0 = (size_t)blockIdx.x * WAVES + wave;
    // Frames in flight from HBM: PF sets (round 6 tried two -- the set an iteration consumes requested two iterations = ~4 us earlier
    // instead of one: the same time, the kernel does not wait for late frames; profiles/r06_config4.md).  The loop over the group is
    // unrolled by two so that the sets are fixed registers.
    constexpr int PF = (MEL >= 1 && !FULL) ? MXG_FUSED_PF : 1;  // (the vector forms have no 32 registers to spare: one set, as before)
    static_assert(kGroup % (2 * NF) == 0, "the group loop is unrolled by two sets of NF frames");
    v2f nxb[PF][NF][8];
#pragma unroll
    for (int p = 0; p < PF; p++)
#pragma unroll
        for (int f = 0; f < NF; f++) load_frame(g0 * kGroup + p * NF + f, nxb[p][f]);
    const int mj = lane >> 3, ms = lane & 7;  // mel walk: frame of the group, slot
    load_tables();
    for (size_t g = g0; g < ngroups; g += gstep) {
        const size_t f0 = g * kGroup;
        if constexpr (NF == 1) {
            const v2f *tl = s_twl;
            asm volatile("" : "+v"(tl));  // (not hoisted out of the group loop)
#pragma unroll
            for (int i = 0; i < 7; i++) {
                tb[i] = tl[lo * 7 + i];
                tc[i] = tl[56 + lane * 7 + i];
            }
#pragma unroll
            for (int i = 0; i < 7; i++) asm volatile("" : "+v"(tb[i]), "+v"(tc[i]));
        }
        const v2f b1[2] = {tb[1], tb[2]}, b2[4] = {tb[3], tb[4], tb[5], tb[6]};
        const v2f c1[2] = {tc[1], tc[2]}, c2[4] = {tc[3], tc[4], tc[5], tc[6]};
#pragma unroll 1
        for (int jj = 0; jj < kGroup; jj += 2 * NF) {
          auto body = [&](const int j, v2f (&nx)[NF][8]) {
            v2f v[NF][8];
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) v[f][e] = nx[f][e] * wv[e];  // calcFFT L/fft.cpp:501-503
            const int jn = j + PF * NF;  // the set that takes this one's registers: PF iterations ahead
            const size_t fnext = jn < kGroup ? f0 + jn : (g + gstep) * kGroup + (jn - kGroup);
#if !(MXG_ABLATE & 16)
#pragma unroll
            for (int f = 0; f < NF; f++) load_frame(fnext + f, nx[f]);
#else
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) asm volatile("" : "+v"(nx[f][e]));
#endif
#if MXG_FUSED_SKEW
            // Frame by frame inside a phase: a frame's butterflies, its transpose stores and the transpose READS are issued together,
            // then the other frame's -- so a frame's LDS round trip flies while the other frame's 60 packed butterflies run, instead
            // of both frames' reads being requested just before the first butterfly that needs them (the DS unit executes one
            // wavefront's instructions in order: the write -> read sequence of a frame needs no wait, only the compiler fence).
#pragma unroll
            for (int f = 0; f < NF; f++) {
#if !(MXG_ABLATE & 2)
                if constexpr (TOL)
                    radix8_t(v[f], c8);
                else if constexpr (MODE == 1)
                    round3_s1(v[f], ta);
                else
                    round3_s(v[f], ta);
#endif
#if !(MXG_ABLATE & (4 | 32))
#pragma unroll
                for (int e = 0; e < 8; e++) X[f][pad8(8 * lane + e)] = v[f][e];
                wave_lds_sync();
#pragma unroll
                for (int e = 0; e < 8; e++) v[f][e] = X[f][pad8(hi * 64 + e * 8 + lo)];
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int f = 0; f < NF; f++) {
#if !(MXG_ABLATE & 2)
                if constexpr (TOL)
                    round8_t(v[f], tb, c8);
                else
                    round3(v[f], tb[0], b1, b2);
#endif
                wave_lds_sync();
#if !(MXG_ABLATE & 4)
#pragma unroll
                for (int e = 0; e < 8; e++) X[f][pad8(hi * 64 + e * 8 + lo)] = v[f][e];
                wave_lds_sync();
#pragma unroll
                for (int e = 0; e < 8; e++) v[f][e] = X[f][pad8(e * 64 + lane)];
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int f = 0; f < NF; f++) {
#if !(MXG_ABLATE & 2)
                if constexpr (TOL)
                    round8_t(v[f], tc, c8);
                else
                    round3(v[f], tc[0], c1, c2);
#endif
                wave_lds_sync();
#pragma unroll
                for (int e = 0; e < 8; e++) X[f][pad8(e * 64 + lane)] = v[f][e];
                if constexpr (SHARED) {  // the post-pass reads of THIS frame before the next frame's store lands on the image
                    static_assert(!SHARED || !FULL, "the shared image serves the half-spectrum post-pass only");
                    wave_lds_sync();
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        v[f][2 * q] = X[f][pa0 + 72 * q];
                        v[f][2 * q + 1] = X[f][pb0 - 72 * q];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_sync();
#else
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if constexpr (TOL)
                    radix8_t(v[f], c8);
                else if constexpr (MODE == 1)
                    round3_s1(v[f], ta);
                else
                    round3_s(v[f], ta);
            }
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) X[f][pad8(8 * lane + e)] = v[f][e];
            wave_lds_sync();
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) v[f][e] = X[f][pad8(hi * 64 + e * 8 + lo)];
            __builtin_amdgcn_sched_barrier(0);  // every frame's reads are in flight before the first butterfly waits
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if constexpr (TOL)
                    round8_t(v[f], tb, c8);
                else
                    round3(v[f], tb[0], b1, b2);
            }
            wave_lds_sync();
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) X[f][pad8(hi * 64 + e * 8 + lo)] = v[f][e];
            wave_lds_sync();
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) v[f][e] = X[f][pad8(e * 64 + lane)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if constexpr (TOL)
                    round8_t(v[f], tc, c8);
                else
                    round3(v[f], tc[0], c1, c2);
            }
            wave_lds_sync();
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int e = 0; e < 8; e++) X[f][pad8(e * 64 + lane)] = v[f][e];
            wave_lds_sync();
#endif
            if constexpr (!FULL) {
                // low half of the post-pass: all the LDS reads are requested before the first is used
                v2f pa[NF][4], pb[NF][4];
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if constexpr (SHARED) {  // (requested inside the third phase, frame by frame)
                            pa[f][q] = v[f][2 * q];
                            pb[f][q] = v[f][2 * q + 1];
                        } else {
                            pa[f][q] = X[f][pa0 + 72 * q];
                            pb[f][q] = X[f][pb0 - 72 * q];
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
                bool garbage[NF];
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    float *Mrow = M + (j + f) * A.mstride;
                    garbage[f] = false;
                    v2f sq[4];
#if MXG_ABLATE & 8
                    sq[0] = pa[f][0]; sq[1] = pa[f][1]; sq[2] = pb[f][2]; sq[3] = pb[f][3];
                    if constexpr (false) {
#else
                    if constexpr (TOL) {
#endif
                        post_lo_sq2_t(pa[f][0], pb[f][0], pw[0], pa[f][1], pb[f][1], pw[1], sq[0], sq[1]);
                        post_lo_sq2_t(pa[f][2], pb[f][2], pw[2], pa[f][3], pb[f][3], pw[3], sq[2], sq[3]);
                    } else {
#if !(MXG_ABLATE & 8)
                        post_lo_sq2(pa[f][0], pb[f][0], pw[0], pa[f][1], pb[f][1], pw[1], sq[0], sq[1]);
                        post_lo_sq2(pa[f][2], pb[f][2], pw[2], pa[f][3], pb[f][3], pw[3], sq[2], sq[3]);
#endif
                    }
                    float m[4];
                    if constexpr (TOL) {
#pragma unroll
                        for (int q = 0; q < 4; q++) m[q] = __builtin_amdgcn_sqrtf(sq[q].x + sq[q].y);
                    } else {
                        const float ss[4] = {sq[0].x + sq[0].y, sq[1].x + sq[1].y, sq[2].x + sq[2].y, sq[3].x + sq[3].y};
                        const bool slow = exact_sqrtf4_try(ss, m);  // L/fft.cpp:510-511
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {  // some lane saw 0, a tiny value, Inf or NaN
                            if constexpr (MODE == 1) {
                                // round3_s1 skipped the (1, 0) products: exact while the transform stays finite.  Every windowed
                                // sample below 2^40 keeps it finite (|X| < 2^50, squares < 2^101); a frame with a larger, infinite
                                // or NaN sample gets what the reference computes for Inf / NaN: NaN in every bin.
                                const size_t fr = f0 + j + f;
                                const unsigned fu = __builtin_amdgcn_readfirstlane((unsigned)(fr < nframes ? fr : nframes - 1));
                                const float *x = A.signal + (size_t)fu * A.frame_stride + 2 * rev6;
                                bool bad = false;
#pragma unroll
                                for (int e = 0; e < 8; e++) {
                                    const float a = x[128 * kRev3[e]] * wv[e].x, b = x[128 * kRev3[e] + 1] * wv[e].y;
                                    bad |= !(__builtin_fabsf(a) < 0x1p40f) || !(__builtin_fabsf(b) < 0x1p40f);
                                }
                                garbage[f] = __builtin_amdgcn_ballot_w64(bad) != 0;
                            }
                            if (garbage[f]) {
#pragma unroll
                                for (int q = 0; q < 4; q++) m[q] = __builtin_nanf("");
                            } else if (slow) {
#pragma unroll
                                for (int q = 0; q < 4; q++) m[q] = sqrtf(ss[q]);
                            }
                        }
                    }
                    if (A.mUncond) {
#pragma unroll
                        for (int q = 0; q < 4; q++) Mrow[1 + lane + 64 * q] = m[q];
                    } else {
                        // bins the bank never reads are not kept; lane 63's fourth pair would be bin 256: post_edge
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (1u + (unsigned)lane + 64u * q < A.mstride && (q < 3 || lane < 63)) Mrow[1 + lane + 64 * q] = m[q];
                    }
                }
                if (A.edgeBins) {
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        post_edge(X[f], j + f, f0);
                        if (garbage[f] && (lane == 0 || lane == 63)) M[(j + f) * A.mstride + zidx] = __builtin_nanf("");
                    }
                }
            } else {
#pragma unroll
                for (int f = 0; f < NF; f++) post_frame(X[f], j + f, f0);
            }
            wave_lds_sync();
          };
          body(jj, nxb[0]);
          body(jj + NF, nxb[PF - 1]);
        }
#if MXG_ABLATE & 1
        if (lane < kGroup * A.numCoeffs && f0 + lane / A.numCoeffs < nframes) A.mfcc[f0 * A.numCoeffs + lane] = (double)M[lane];
        wave_lds_sync();
        continue;
#endif
        // ---- matrix layout tail (MEL >= 1): logs in place, DCT on the matrix pipe, one coalescing-free store per coefficient quad
        auto finish_mm = [&](const MmLane &L, const double (&acc)[kMmPairs]) {
            double lg[kMmPairs];
#pragma unroll
            for (int p = 0; p < kMmPairs; p++) lg[p] = log_square(acc[p]);  // L/maxiMFCC.cpp:63
            const size_t fr = f0 + 4 * L.fh + L.ij;
            if (A.melraw || A.melbands) {
#pragma unroll
                for (int p = 0; p < kMmPairs; p++) {
                    const unsigned ff = 4 * (2 * p + L.gs) + L.k;
                    if (ff < A.numFilters && fr < nframes) {
                        if (A.melraw) A.melraw[fr * A.numFilters + ff] = acc[p];
                        if (A.melbands) A.melbands[fr * A.numFilters + ff] = lg[p];
                    }
                }
            }
            double C[kMmCoefQuads];
            dct_mfma(lg, s_d, L, C);
            if (L.gs == 0 && fr < nframes) {
#pragma unroll
                for (int q = 0; q < kMmCoefQuads; q++)
                    if (4u * q + L.k < A.numCoeffs) A.mfcc[fr * A.numCoeffs + 4 * q + L.k] = C[q] / (double)A.numCoeffs;  // L/maxiMFCC.h:108
            }
        };
        if constexpr (MEL == 1) {
            const MmLane L = mm_lane(lane);
            double acc[kMmPairs];
            mel_mfma(M, A.mstride, s_mmW, A.mm, L, acc);
            finish_mm(L, acc);
            wave_lds_sync();
            continue;
        }
        if constexpr (TOL && MEL == 0) {
            // ---- tolerance mode: the same three phases in fp32 (mel_walk_t, v_log_f32, dct_dot_rows_f) ---------------------------
            float *s_melf = reinterpret_cast<float *>(s_mel);
            for (unsigned i = lane; i < kGroup * A.nfpf; i += 64) s_melf[i] = 0.0f;  // empty filters and the row padding stay 0
            wave_lds_sync();
            mel_walk_t<kFusedSlots>(reinterpret_cast<const char *>(M + mj * A.mstride), s_melf + mj * A.nfpf, s_fs32 + ms, A.steps);
            wave_lds_sync();
            for (unsigned idx = lane; idx < kGroup * A.numFilters; idx += 64) {
                const unsigned jj = idx / A.numFilters, ff = idx - jj * A.numFilters;
                const float raw = s_melf[jj * A.nfpf + ff];
                // log(mb * mb) = 2 ln 2 * log2(mb) through the hardware's fp32 log2 (abs. error ~1e-6 on values of magnitude <= 30)
                const float lv = raw > 0.000001f ? 1.3862943611198906f * __builtin_amdgcn_logf(raw) : 0.0f;
                s_melf[jj * A.nfpf + ff] = lv;
                if (f0 + jj < nframes) {
                    if (A.melraw) A.melraw[(f0 + jj) * A.numFilters + ff] = (double)raw;
                    if (A.melbands) A.melbands[(f0 + jj) * A.numFilters + ff] = (double)lv;
                }
            }
            wave_lds_sync();
            const float rcpNC = 1.0f / (float)A.numCoeffs;
            for (unsigned p = lane; p < kGroup * A.numCoeffs; p += 64) {
                const unsigned jj = p / A.numCoeffs, i = p - jj * A.numCoeffs;
                const float c = dct_dot_rows_f(s_df + i * A.nfpf, s_melf + jj * A.nfpf, A.numFilters);
                if (f0 + jj < nframes) A.mfcc[(f0 + jj) * A.numCoeffs + i] = (double)(c * rcpNC);
            }
            wave_lds_sync();
            continue;
        }
        // ---- mel walk: lane (mj, ms) walks slot ms's filter list over frame mj's magnitudes -----------------
        for (unsigned i = lane; i < kGroup * A.nfp; i += 64) s_mel[i] = 0.0;  // filters with an empty support stay 0
        wave_lds_sync();
        mel_walk<kFusedSlots>(reinterpret_cast<const char *>(M + mj * A.mstride), s_mel + mj * A.nfp, s_fs + ms, A.steps);
        wave_lds_sync();
        if constexpr (MEL == 2) {  // the exact band sums into matrix layout (rows of nfp >= 48 doubles, zeros beyond numFilters)
            const MmLane L = mm_lane(lane);
            double acc[kMmPairs];
#pragma unroll
            for (int p = 0; p < kMmPairs; p++) acc[p] = s_mel[(4 * L.fh + L.ij) * A.nfp + 4 * (2 * p + L.gs) + L.k];
            finish_mm(L, acc);
            wave_lds_sync();
            continue;
        }
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
            const double c = dct_dot_rows(s_d + i * A.nfp, s_mel + jj * A.nfp, A.numFilters);
            if (f0 + jj < nframes) A.mfcc[(f0 + jj) * A.numCoeffs + i] = c / (double)A.numCoeffs;
        }
        wave_lds_sync();  // the next frame's first transpose overwrites the band rows
    }
}

// ---- the 16-wave form (fft_mfcc16_kernel) -----------------------------------------------------------------------------
// Same arithmetic, laid out for FOUR wavefronts per SIMD.  Issue costs measured with tools/ubench (profiles/r02_ubench.md):
// one wave per SIMD issues a VALU op every 4.35 clk, two waves 2.4 clk per SIMD, four waves 1.7 clk -- the kernel above
// (2 workgroups of 4 waves per CU: its 15.8 KB of LDS per wave and 162 VGPRs allow no more) leaves almost half of the
// issue rate unused.  Here ONE 1024-thread workgroup per CU shares the tables; a wave owns 8 KB of LDS:
//   X      the FFT transposes (4608 B), re-used for the band rows of the mel stage once the last frame's post-pass is done;
//   M      magnitude rows of FOUR frames, only the bins the bank reads (row stride = nbUsed rounded up to 8 mod 16 floats,
//          so the four rows start 8 banks apart);
// and at most 128 VGPRs (round-1 twiddles are wave-uniform: SGPRs).  The mel walk runs lane = (frame of 4, slot of 16) over
// the 16-list packing of the plan; a step is one 16-byte table entry {weight, byte offset, closing filter} and one LDS
// gather, and the loop is software-pipelined two batches deep (entries two batches ahead, gathers one) so that no LDS
// latency sits between the dependent fp64 adds.  Per-filter sums are still the reference's sequential sums (bit-identical
// to the kernel above and to mxg_mfcc_batch).  Used when no magnitudes are requested and the bank reads bins [1, 256].
constexpr int kGroup16 = 4;
constexpr int kWaves16 = 16;

struct Fused16Args {
    const float *signal;
    size_t frame_stride, nframes;
    const float *window;
    const float2 *tw, *post;
    unsigned numFilters, numCoeffs, mstride, nfp, dctPad;
    int steps;
    const mxg_fs_entry *fs;
    const double *dct;
    double *melraw, *melbands, *mfcc;
};

template <bool ALIGNED8>
__global__ __launch_bounds__(64 * kWaves16) void fft_mfcc16_kernel(const Fused16Args A) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    // [tw 512 float2][fs (steps + 2 batches) * 16 entries][dct NF*NC f64, padded to 16 B] | per wave: X (= band rows), M
    float2 *s_tw = reinterpret_cast<float2 *>(s_dyn);
    mxg_fs_entry *s_fs = reinterpret_cast<mxg_fs_entry *>(s_tw + 512);
    const int fsRows = A.steps + 2 * kMelBatch;
    double *s_d = reinterpret_cast<double *>(s_fs + (size_t)fsRows * kFusedSlots16);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t perWaveBytes = sizeof(float2) * kX1024 + sizeof(float) * kGroup16 * A.mstride;
    char *wbase = reinterpret_cast<char *>(s_d + A.dctPad) + (size_t)wave * perWaveBytes;
    float2 *X = reinterpret_cast<float2 *>(wbase);
    double *s_mel = reinterpret_cast<double *>(wbase);
    float *M = reinterpret_cast<float *>(wbase + sizeof(float2) * kX1024);
    for (int i = threadIdx.x; i < 511; i += blockDim.x) s_tw[i] = A.tw[i];
    for (int i = threadIdx.x; i < fsRows * kFusedSlots16; i += blockDim.x) s_fs[i] = A.fs[i];
    for (unsigned i = threadIdx.x; i < A.numFilters * A.numCoeffs; i += blockDim.x) s_d[i] = A.dct[i];
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
    float2 pw[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pw[q] = A.post[(1 + lane + 64 * q) < 256 ? 1 + lane + 64 * q : 255];
    asm volatile("" : "+v"(pw[0].x), "+v"(pw[0].y), "+v"(pw[1].x), "+v"(pw[1].y), "+v"(pw[2].x), "+v"(pw[2].y),
                 "+v"(pw[3].x), "+v"(pw[3].y));
    // round-1 twiddles are the same for every lane: scalar registers
    float2 ta[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float2 t = A.tw[i];
        ta[i].x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t.x)));
        ta[i].y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t.y)));
    }
    const int pa0 = pad8(1 + lane), pb0 = pad8(511 - lane);
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
    const size_t ngroups = (nframes + kGroup16 - 1) / kGroup16;
    const size_t gstep = (size_t)gridDim.x * kWaves16;
    const size_t g0 = (size_t)blockIdx.x * kWaves16 + wave;
    float2 nxt[8];
    load_frame(g0 * kGroup16, nxt);
    const int mj = lane >> 4, ms = lane & 15;  // mel walk: frame of the group, slot
    for (size_t g = g0; g < ngroups; g += gstep) {
        const size_t f0 = g * kGroup16;
#pragma unroll 1
        for (int j = 0; j < kGroup16; j++) {
            float2 v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[e].x = nxt[e].x * wv[e].x;  // calcFFT L/fft.cpp:501-503
                v[e].y = nxt[e].y * wv[e].y;
            }
            load_frame(j + 1 < kGroup16 ? f0 + j + 1 : (g + gstep) * kGroup16, nxt);
            {
                const float2 a1[2] = {ta[1], ta[2]};
                const float2 a2[4] = {ta[3], ta[4], ta[5], ta[6]};
                round3(v, ta[0], a1, a2);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) X[pad8(8 * lane + e)] = v[e];
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = X[pad8(hi * 64 + e * 8 + lo)];
            {
                const float2 b0 = s_tw[7 + lo];
                const float2 b1[2] = {s_tw[15 + lo], s_tw[15 + 8 + lo]};
                const float2 b2[4] = {s_tw[31 + lo], s_tw[31 + 8 + lo], s_tw[31 + 16 + lo], s_tw[31 + 24 + lo]};
                round3(v, b0, b1, b2);
            }
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) X[pad8(hi * 64 + e * 8 + lo)] = v[e];
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = X[pad8(e * 64 + lane)];
            {
                const float2 c0 = s_tw[63 + lane];
                const float2 c1[2] = {s_tw[127 + lane], s_tw[127 + 64 + lane]};
                const float2 c2[4] = {s_tw[255 + lane], s_tw[255 + 64 + lane], s_tw[255 + 128 + lane], s_tw[255 + 192 + lane]};
                round3(v, c0, c1, c2);
            }
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 8; e++) X[pad8(e * 64 + lane)] = v[e];
            wave_lds_sync();
            // real split post-pass (L/fft.cpp:245-275), low half only, + magnitudes (cartToPol :510-511) into row j of M
            float *Mrow = M + j * A.mstride;
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                v2f sq0, sq1;
                post_lo_sq2(as_v2f(X[pa0 + 72 * q]), as_v2f(X[pb0 - 72 * q]), as_v2f(pw[q]), as_v2f(X[pa0 + 72 * (q + 1)]),
                            as_v2f(X[pb0 - 72 * (q + 1)]), as_v2f(pw[q + 1]), sq0, sq1);
                const float m0 = exact_sqrtf(sq0.x + sq0.y), m1 = exact_sqrtf(sq1.x + sq1.y);
                if (1u + (unsigned)lane + 64u * q < A.mstride) Mrow[1 + lane + 64 * q] = m0;  // bins the bank never reads are not kept
                if (65u + (unsigned)lane + 64u * q < A.mstride) Mrow[65 + lane + 64 * q] = m1;
            }
            wave_lds_sync();
        }
        // ---- mel walk: lane (mj, ms) walks slot ms's filter list over frame mj's magnitudes -----------------
        for (unsigned i = lane; i < kGroup16 * A.nfp; i += 64) s_mel[i] = 0.0;  // filters with an empty support stay 0
        wave_lds_sync();
        mel_walk<kFusedSlots16>(reinterpret_cast<const char *>(M + mj * A.mstride), s_mel + mj * A.nfp, s_fs + ms, A.steps);
        wave_lds_sync();
        // ---- log-square (L/maxiMFCC.cpp:63), one band per lane ---------------------------------------------
        for (unsigned idx = lane; idx < kGroup16 * A.numFilters; idx += 64) {
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
        for (unsigned p = lane; p < kGroup16 * A.numCoeffs; p += 64) {
            const unsigned jj = p / A.numCoeffs, i = p - jj * A.numCoeffs;
            const double c = dct_dot(s_d + i, A.numCoeffs, s_mel + jj * A.nfp, A.numFilters);
            if (f0 + jj < nframes) A.mfcc[(f0 + jj) * A.numCoeffs + i] = c / (double)A.numCoeffs;
        }
        wave_lds_sync();  // the next frame's first transpose overwrites the band rows
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
    MXG_REQUIRE(mp->fsSteps > 0 && mp->d_fs8, "this filter bank has no fused schedule (numFilters > 64?): use the two-kernel path");
    MXG_REQUIRE(nframes < ((size_t)1 << 32), "nframes must be < 2^32");
    if (nframes == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const bool aligned8 = (((uintptr_t)d_signal) & 7) == 0 && (frame_stride & 1) == 0;
    if (!d_mags && mp->fs16Steps > 0 && mp->fsMinBin >= 1 && mp->nbUsed <= 256 && tune_get("fused_waves16")) {
        Fused16Args B;
        B.signal = d_signal; B.frame_stride = frame_stride; B.nframes = nframes;
        B.window = fp->d_window; B.tw = fp->d_tw; B.post = fp->d_post;
        B.numFilters = mp->numFilters; B.numCoeffs = mp->numCoeffs;
        B.mstride = (mp->nbUsed + 7) / 16 * 16 + 8;  // smallest stride >= nbUsed that is 8 mod 16: the four rows start 8 or 24 banks apart
        B.nfp = mp->numFilters | 1u;
        B.dctPad = (mp->numFilters * mp->numCoeffs + 1) & ~1u;
        B.steps = mp->fs16Steps; B.fs = mp->d_fs16; B.dct = mp->d_dct;
        B.melraw = d_melraw; B.melbands = d_melbands; B.mfcc = d_mfcc;
        const size_t perWave = sizeof(float2) * kX1024 + sizeof(float) * kGroup16 * B.mstride;
        const size_t lds = sizeof(float2) * 512 + sizeof(mxg_fs_entry) * (size_t)(B.steps + 2 * kMelBatch) * kFusedSlots16 +
                           sizeof(double) * B.dctPad + kWaves16 * perWave;
        if (lds <= 160 * 1024 && sizeof(double) * kGroup16 * B.nfp <= sizeof(float2) * kX1024) {
            const size_t ngroups = (nframes + kGroup16 - 1) / kGroup16;
            size_t blocks = (ngroups + kWaves16 - 1) / kWaves16;
            if (blocks > 256) blocks = 256;  // persistent: one 16-wave workgroup per CU, grid-stride over groups of 4 frames
            typedef void (*kern16_t)(const Fused16Args);
            kern16_t k = aligned8 ? fft_mfcc16_kernel<true> : fft_mfcc16_kernel<false>;
            if (lds > 64 * 1024) MXG_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            KernelTimer kt("fft_mfcc_kernel", st);
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * kWaves16), lds, st, B);
            return check_hip(hipGetLastError(), "fft_mfcc16_kernel launch");
        }
    }
    FusedArgs A;
    A.signal = d_signal; A.frame_stride = frame_stride; A.nframes = nframes;
    A.window = fp->d_window; A.tw = fp->d_tw; A.post = fp->d_post; A.tw8 = fp->d_tw8;
    A.numFilters = mp->numFilters; A.numCoeffs = mp->numCoeffs; A.nbUsed = mp->nbUsed;
    const bool tol = !tune_get("fft_exact") && fp->d_tw8 != nullptr;  // tolerance mode (reordered, fused arithmetic): opt-in
    MXG_REQUIRE(mp->nbUsed <= 257, "mel bank reaches beyond bin 256");  // binFreq = sr/numBins*bin never does
    const bool full = d_mags != nullptr;
    A.mstride = full ? 520 : 264;                 // the post-pass writes bins 0..256 (0..511 with magnitudes out) + pad, = 8 mod 32
    // a bank that reads neither bin 0 nor bin 256 keeps only the bins below nbUsed (stride 8 mod 32 floats): with the two X
    // images that is what lets two workgroups share a CU's 160 KB
    if (!full && mp->fsMinBin >= 1 && mp->nbUsed <= 256) A.mstride = (mp->nbUsed + 23) / 32 * 32 + 8;
    // row stride (doubles) of the band rows and of the coefficient-major DCT table: even (16-byte aligned rows for ds_read_b128) with
    // an odd half, so that the rows of 16 consecutive coefficients start in 16 different 16-byte bank slots
    A.nfp = (mp->numFilters + 1) & ~1u;
    if (!((A.nfp / 2) & 1)) A.nfp += 2;
    A.nfpf = (mp->numFilters + 7) & ~7u;
    if (!((A.nfpf / 4) & 1)) A.nfpf += 4;
    A.dctPad = mp->numCoeffs * A.nfp;
    if (A.dctPad < (mp->numCoeffs * A.nfpf + 1) / 2) A.dctPad = (mp->numCoeffs * A.nfpf + 1) / 2;  // (tiny banks: the fp32 rows are longer)
    A.dctPad = (A.dctPad + 1) & ~1u;
    A.steps = mp->fsSteps; A.fs = mp->d_fs8; A.dct = mp->d_dct;
    A.edgeBins = full || mp->fsMinBin < 1 || mp->nbUsed > 256;
    A.mags = d_mags; A.melraw = d_melraw; A.melbands = d_melbands; A.mfcc = d_mfcc;
    const size_t tablesBytes = sizeof(mxg_fs_entry) * (size_t)(A.steps + 2 * kMelBatch) * kFusedSlots + sizeof(double) * A.dctPad;
    // layout (knob fused_layout): 1 = two frames in flight, two 4-wave workgroups per CU; 2 = one frame in flight, one 12-wave
    // workgroup per CU (only without the full magnitude rows: twelve 8 x 520 tiles do not fit); 0 = automatic
    constexpr int kWaves1 = 12;
    auto lds_for = [&](int nf, int waves, unsigned slack) {
        return tablesBytes + (nf == 1 ? sizeof(float2) * (8 + 64) * 7 : 0) +
               waves * (nf * sizeof(float2) * kX1024 + sizeof(float) * (kGroup * A.mstride + slack));
    };
    const unsigned slackWanted = A.mstride >= 257 ? 0u : (257 - A.mstride + 3) & ~3u;  // per-wave regions stay 16-byte aligned
    int layout = (int)tune_get("fused_layout");
    if (layout == 0) layout = tol ? 2 : 1;  // measured (1 M frames): exact 1.36 / 1.34-1.38 ms, tolerance mode 1.10 / 1.06 ms for layouts 1 / 2
    if (layout == 2 && (full || lds_for(1, kWaves1, 0) > 160 * 1024)) layout = 1;
    // mel / log / DCT stage (knob fused_mel): the matrix-pipe forms need the plan's quad tables, a half-spectrum launch and a bank
    // inside bins [1, 255]; they run as one 8-wave workgroup per CU on rows of 260 floats
    int mel = (int)tune_get("fused_mel");
    // automatic: the matrix pipe for the DCT always (the DCT only ever saw the device log's values: the mfcc tolerance does not move),
    // for the mel contraction too when the caller does not ask for the band sums -- a caller who reads d_melraw / d_melbands gets the
    // sparse walk's sums, bit for bit the reference's (measured, 2^20 frames: 1.30 / 1.19 / 1.13 ms for forms 1 / 2 / 3)
    if (mel == 0) mel = (d_melraw || d_melbands) ? 2 : 3;
    const bool mmOk = mp->mmOk && mp->d_mmW && mp->d_mmD && !full && !A.edgeBins;
    const int MEL = mel == 3 && mmOk ? 1 : (mel == 2 && mmOk ? 2 : 0);
    if (MEL) {
        layout = 1;
        A.mstride = 260;
        A.mmW = mp->d_mmW; A.mmD = mp->d_mmD; A.mmBatches = mp->mmBatches;
        for (int pr = 0; pr < kMmPairs; pr++) {
            A.mm.nb[pr] = mp->mmNb[pr];
            A.mm.base[pr][0] = mp->mmBase[pr][0];
            A.mm.base[pr][1] = mp->mmBase[pr][1];
        }
        if (MEL == 2) A.nfp = 50;  // band rows of >= 48 doubles (the matrix layout reads quads up to filter 47), 16-byte aligned, odd half
    } else {
        A.mmW = A.mmD = nullptr;
        A.mmBatches = 0;
    }
    constexpr int kWavesMm = 8;
    // (four frames in flight per wavefront -- the shared image leaves the LDS for it -- were measured: 256 VGPRs do not hold them
    // without spills and the tables re-read from LDS, 1.41 ms against 1.14: profiles/r05_config4_summary.md)
    // (likewise three frames in flight in groups of six: no spills, 1.21 ms against 1.14 -- more frames per wavefront do not help, the
    // kernel is not waiting for its own LDS round trips)
    const int nf = MEL ? 2 : (layout == 2 ? 1 : 2), waves = MEL ? kWavesMm : (layout == 2 ? kWaves1 : kWavesPerBlock),
              wgPerCU = MEL || layout == 2 ? 1 : 2;
    // unconditional magnitude stores (see FusedArgs::mUncond): rows of >= 257 floats, or a slack behind the tile that still lets
    // the layout's workgroups share a CU
    A.mslack = 0;
    A.mUncond = A.mstride >= 257;
    if (!A.mUncond && wgPerCU * lds_for(nf, waves, slackWanted) <= 160 * 1024) {
        A.mslack = slackWanted;
        A.mUncond = 1;
    }
    size_t lds = lds_for(nf, waves, A.mslack);
    if (MEL)  // [mmW (batches + 1) KB | fs][mmD 6 KB] + 8 waves x (ONE X image + the 8 x 260 tile)
        lds = (MEL == 1 ? sizeof(double) * 128 * (size_t)(A.mmBatches + 1) : sizeof(mxg_fs_entry) * (size_t)(A.steps + 2 * kMelBatch) * kFusedSlots) +
              sizeof(double) * kMmCoefQuads * kMmPairs * 32 + waves * (sizeof(float2) * kX1024 + sizeof(float) * kGroup * A.mstride);
    if (MEL && (lds > 160 * 1024 || sizeof(double) * kGroup * A.nfp > sizeof(float2) * kX1024))
        return fail(MXG_ERR_INVALID, "fused_mel %d: the tables do not fit the LDS (%zu bytes)", mel, lds);
    MXG_REQUIRE(lds <= 160 * 1024, "filter bank too large for the fused kernel's LDS layout");
    const size_t ngroups = (nframes + kGroup - 1) / kGroup;
    size_t blocks = (ngroups + waves - 1) / waves;
    const size_t cap = 256 * (size_t)wgPerCU;  // persistent: grid-stride over groups of 8 frames
    if (blocks > cap) blocks = cap;
    typedef void (*kern_t)(const FusedArgs);
    kern_t k;
    const int mode = tol ? 2 : (fp->round1Trivial ? 1 : 0);
#define MXG_PICK(FULL_, WM_, NF_, W_)                                                                                          \
    (mode == 2   ? (aligned8 ? fft_mfcc_kernel<FULL_, WM_, true, 2, NF_, W_> : fft_mfcc_kernel<FULL_, WM_, false, 2, NF_, W_>) \
     : mode == 1 ? (aligned8 ? fft_mfcc_kernel<FULL_, WM_, true, 1, NF_, W_> : fft_mfcc_kernel<FULL_, WM_, false, 1, NF_, W_>) \
                 : (aligned8 ? fft_mfcc_kernel<FULL_, WM_, true, 0, NF_, W_> : fft_mfcc_kernel<FULL_, WM_, false, 0, NF_, W_>))
#define MXG_PICK_FULL(WM_)                                                                                                     \
    (tol ? (aligned8 ? fft_mfcc_kernel<true, WM_, true, 2, 2, kWavesPerBlock> : fft_mfcc_kernel<true, WM_, false, 2, 2, kWavesPerBlock>) \
         : (aligned8 ? fft_mfcc_kernel<true, WM_, true, 0, 2, kWavesPerBlock> : fft_mfcc_kernel<true, WM_, false, 0, 2, kWavesPerBlock>))
    if (d_mags)  // all 512 bins: the generic first round (MODE 1's non-finite handling lives in the half-spectrum post-pass)
        k = MXG_PICK_FULL(true);
    else if (full)
        k = MXG_PICK_FULL(false);
#define MXG_PICK_MM(NF_, MEL_)                                                                                                       \
    (mode == 2   ? (aligned8 ? fft_mfcc_kernel<false, false, true, 2, NF_, kWavesMm, MEL_> : fft_mfcc_kernel<false, false, false, 2, NF_, kWavesMm, MEL_>) \
     : mode == 1 ? (aligned8 ? fft_mfcc_kernel<false, false, true, 1, NF_, kWavesMm, MEL_> : fft_mfcc_kernel<false, false, false, 1, NF_, kWavesMm, MEL_>) \
                 : (aligned8 ? fft_mfcc_kernel<false, false, true, 0, NF_, kWavesMm, MEL_> : fft_mfcc_kernel<false, false, false, 0, NF_, kWavesMm, MEL_>))
    else if (MEL == 1)
        k = MXG_PICK_MM(2, 1);
    else if (MEL == 2)
        k = MXG_PICK_MM(2, 2);
#undef MXG_PICK_MM
    else if (layout == 2)
        k = MXG_PICK(false, false, 1, kWaves1);
    else
        k = MXG_PICK(false, false, 2, kWavesPerBlock);
#undef MXG_PICK_FULL
#undef MXG_PICK
    if (lds > 64 * 1024) MXG_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer kt("fft_mfcc_kernel", st);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * (unsigned)waves), lds, st, A);
    return check_hip(hipGetLastError(), "fft_mfcc_kernel launch");
}
