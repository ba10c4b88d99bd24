// osctab.hip -- EXTENSION (labelled as such everywhere): maxiOsc::sinebuf over PER-VOICE wavetables.
//
// The reference has ONE shared 514-entry table (`double sineBuffer[514]`, src/maximilian.cpp:63) that sinebuf interpolates
// (C:266-274); with it the voice-bank render has no read stream to speak of (the table lives in LDS, osc.hip).  SURVEY.md 8(d) row 2
// and north_star name the variant in which every voice owns its table -- [V][514] doubles, 4112 B read per voice and block = 8.03 B per
// sample at 512-sample blocks -- as the HBM-READ roofline of the wavetable path.  Same arithmetic, same expression order, per voice
//     phase += 512./(sr/(freq*chandiv)); if (phase >= 511) phase -= 512; rem = phase - floor(phase);
//     out = (1-rem)*T_v[1+(long)phase] + rem*T_v[2+(long)phase]
// so a bank whose tables all equal sineBuffer gives mxg_osc_render(sinebuf)'s bits, i.e. the reference's (tests/test_gpu_osctab.py).
//
// A table must be on chip for the whole block to be read ONCE, and 160 KB of LDS hold 38 of them: one lane per voice would leave
// a CU with 38 busy lanes.  So a block is cut along TIME as well: 16 lanes per voice, lane (u, t) renders samples [t PL, (t+1) PL) of
// voice u (PL = 16 or 32), a workgroup of 256 lanes = 16 voices per ROUND.  The phase at the start of each part comes from a small
// first kernel (lanes = voices, the recurrence without its output -- the same additions in the same order, the same bits -- leaving 16
// marks per voice: 3 % of the table traffic, written and read once).  Rounds are double-buffered: while round r is rendered from one
// half of the LDS, the 65 792 bytes of round r + 1's tables stream into the other half with global_load_lds_dwordx4 (LDS-DMA: no
// registers, 1 KiB per wave instruction; the source of a round is ONE contiguous piece of the table array), one barrier per round.
// A persistent grid of one workgroup per CU walks contiguous ranges of voice groups.
// Output: the per-voice block is optional (out[n][v]; rows of 128 contiguous bytes per store: a convenience, not a fast path); the
// measured form is the fused maxiMix::stereo mixdown (C:503-509 + the user's sum over voices): the 16 voices of a round sit in the 16
// lanes of a DPP row, a transposing butterfly (mxg_lanefold.h) turns 16 samples x 16 voices into 16 sums, and every lane keeps the
// running sum of ITS sample over all rounds in registers; the per-workgroup rows [workgroup][N][2] are added by mix_partials_kernel.
#include <map>
#include <mutex>

#include "mxg_common.h"
#include "mxg_lanefold.h"
#include "mxg_osc.h"

namespace mxg {
namespace {

#ifndef MXG_TAB_ABLATE
#define MXG_TAB_ABLATE 0  // timing experiments only (wrong results): 1 = no rendering (DMA + barriers only), 2 = no DMA inside the loop (the first pair's tables rendered again and again)
#endif
#ifndef MXG_TAB_AUX
#define MXG_TAB_AUX 2  // A/B: 0 = default cache policy
#endif
#ifndef MXG_TAB_SIDEFOLD
#define MXG_TAB_SIDEFOLD 1  // A/B: 0 = a partial mix row per workgroup AND side (round 4: twice the rows for the row sum)
#endif
constexpr int kTabLen = 514;         // doubles per voice (sineBuffer[514], C:63)
constexpr int kTabParts = 16;        // time parts per block (32 samples each), one mark per part and voice
constexpr int kTabVoices = 8;        // voices per round
constexpr int kTabHdr = kTabParts * kTabVoices + 3 * kTabVoices;  // per-round header: marks[16][8], inc[8], gl[8], gr[8] (152 doubles)
constexpr int kTabRound = kTabVoices * kTabLen + kTabHdr;         // doubles per LDS buffer: 4112 + 152 = 4264 (34 112 B)
static_assert(kTabRound * 8 == 2048 * 16 + 84 * 16, "a round is 8 full 4 KiB pieces + 84 sixteen-byte pieces");

// (An experiment that stays as an A/B form, MXG_TAB_CMPX 2.)  The wrap test ahead of the add.  C:270 tests the SUM: `phase += inc; if (phase >= 511) phase -= 512;` -- add, compare, select,
// subtract, four dependent fp64 operations per step of a chain that nothing else can hide (the marks pass IS that chain, 512 steps per
// voice).  d -> fl(d + inc) is monotone, so there is one double `thr` with  fl(ph + inc) >= 511  <=>  ph >= thr  for every ph that is
// not a NaN (a NaN fails both tests): the compare then runs BESIDE the add and the chain is add -> subtract.  thr is found by bisection
// over the doubles' order with the add itself as the oracle: no reasoning about roundings to get wrong.  Returns false where the search
// would not bracket (inc not in (0, 511): a negative, zero, absurd or non-finite frequency): the caller keeps the reference's form.
__device__ __forceinline__ long long dbl_key(double d) {  // order-preserving map double -> integer (both zeros -> 0)
    const long long b = __double_as_longlong(d);
    return b < 0 ? (long long)0x8000000000000000ull - b : b;
}
__device__ __forceinline__ double key_dbl(long long k) {
    return __longlong_as_double(k < 0 ? (long long)0x8000000000000000ull - k : k);
}
__device__ __forceinline__ bool wrap_threshold(const double inc, double &thr) {
    if (!(inc > 0.0 && inc < 511.0)) return false;
    long long lo = dbl_key(-2.0), hi = dbl_key(511.0);  // fl(-2 + inc) < 511 <= fl(511 + inc)
    while (hi - lo > 1) {
        const long long mid = lo + (hi - lo) / 2;
        if (key_dbl(mid) + inc >= 511) hi = mid; else lo = mid;
    }
    thr = key_dbl(hi);
    return true;
}

// pass 1 (lanes = voices): the recurrence without its output -- the same additions in the same order, the same bits -- leaving per
// group of 8 voices ONE contiguous header: the phase at the start of each 32-sample part, the increment, the two gains.  The main
// kernel fetches it with the group's tables, by DMA: no ordinary load in its loop (hipcc waits vmcnt(0) at the use of one while
// LDS-DMA pieces are in flight, which would drain the ring every round).
// phase_out: where the phase after the block goes -- phase_io itself, or (the pipelined form, below) the carry array.
__global__ void osctab_marks_kernel(size_t V, size_t N, const double *__restrict__ freq, const double *__restrict__ pan,
                                    const double *__restrict__ phase_io, double *__restrict__ phase_out, double *__restrict__ hdr, double sr) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double *h = hdr + (v / kTabVoices) * kTabHdr;
    const int u = (int)(v % kTabVoices);
    double ph = phase_io[v];
    const double inc = 512. / (sr / (freq[v] * kChandiv));  // C:269
    h[kTabParts * kTabVoices + u] = inc;
    double x = pan ? pan[v] : 0.0;
    if (x > 1) x = 1;  // C:504
    if (x < 0) x = 0;  // C:505
    h[kTabParts * kTabVoices + kTabVoices + u] = sqrt(1.0 - x);  // two[0] = input*sqrt(1.0-x)   C:506
    h[kTabParts * kTabVoices + 2 * kTabVoices + u] = sqrt(x);    // two[1] = input*sqrt(x)       C:507
    const int whole = (int)(N / 32);  // parts that lie inside the block completely
#ifndef MXG_TAB_CMPX
#define MXG_TAB_CMPX 1  // A/B: 0 = compare + select + subtract (round 4), 2 = the wrap test ahead of the add (wrap_threshold)
#endif
    // The pass is bound by VALU issue, not by the chain's latency: measured, 131 072 voices (two wavefronts per SIMD) take twice the
    // time of 65 536 (one), and a chain with the compare taken off the critical path (MXG_TAB_CMPX 2: five instructions per step instead
    // of four + a wait state) takes the same 17-18 us (profiles/r05_k1t.md).  So the step is cut to THREE vector instructions: the add,
    // a compare that writes the EXEC mask itself (v_cmpx), the subtraction under that mask; the mask is restored on the scalar unit,
    // which issues beside the vector ALU.  `ph + (-512.0)` is `ph - 512` bit for bit; a NaN fails the compare as it fails `>=`.
    double thr = 0.0;
    const bool ahead = MXG_TAB_CMPX == 2 && __all(wrap_threshold(inc, thr));  // (wave-uniform choice of the loop form)
    const unsigned long long live = __builtin_amdgcn_read_exec();
    const double c511 = 511.0, cm512 = -512.0;
#pragma unroll 1
    for (int t = 0; t < kTabParts; t++) {
        h[t * kTabVoices + u] = ph;
        if (t < whole) {
            if (MXG_TAB_CMPX == 1) {
#pragma unroll
                for (int k = 0; k < 32; k += 4)
                    asm volatile("v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4\n\t"
                                 "v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4\n\t"
                                 "v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4\n\t"
                                 "v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4"
                                 : "+v"(ph)
                                 : "v"(inc), "s"(c511), "s"(cm512), "s"(live)
                                 : "vcc");
            } else if (ahead) {
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    const double sum = ph + inc;                                             // C:269
                    const double sub = __hiloint2double(ph >= thr ? 0x40800000 : 0, 0);      // 512.0 where C:270's test holds, else +0.0
                    ph = sum - sub;                                                          // (x - 0.0 is x for every x)
                }
            } else {
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    ph += inc;
                    // `if (ph >= 511) ph -= 512;` (C:270) as compare + ONE select + subtract: the subtrahend is 512.0 or +0.0, assembled from
                    // its high word (x - 0.0 is x for every x)
                    ph = ph - __hiloint2double(ph >= 511 ? 0x40800000 : 0, 0);
                }
            }
        } else {
            for (size_t n = (size_t)t * 32; n < N; n++) {
                ph += inc;
                if (ph >= 511) ph -= 512;
            }
        }
    }
    phase_out[v] = ph;
}

// ---- the marks pass ONE BLOCK AHEAD, inside the render kernel (round 6) ---------------------------------------------------------------
// Block k + 1's marks need only block k's: they do not have to sit in front of their own render.  In the pipelined form
// (mxg_osc_render_tables_ex, flag MXG_TABLES_AHEAD) the render kernel of block k ALSO runs the recurrence of block k + 1: the workgroup
// that renders voice groups [g0, g1) has 512 lanes and at most 512 voices, so lane j walks voice 8 g0 + j -- 16 steps per pair of
// rounds, at the end of the iteration, where the wavefront would otherwise wait for the next pair's DMA -- and leaves that voice's
// marks, increment and gains in the OTHER header buffer.  No second kernel, no second stream, no event.  The caller's promise: the next
// call comes with the same V, N, frequencies and pans and has not touched d_phase in between; a call that does not match (or comes
// without the flag) finds d_phase holding the phase after the last RENDERED block and starts over with the marks kernel above.
// carry[v]: in = the phase block k + 1 starts from (left by the previous launch's duty, or by the marks kernel), out = where it ends.
struct TabAhead {
    const double *freq, *pan;
    double *carry, *phase_io, *hdr_next;
    double sr;
};
// The row sum inside the render kernel (mix != null): the workgroup that finishes LAST among those whose index is w mod 16 adds their
// rows in index order -- mix_partials_kernel's chain w (mxg_lanefold.h) -- and the last of those sixteen adds the chains left to right:
// the additions of mxg_mix_rows_sum in its order, the same bits, without a second launch.  tickets: 17 counters, zero between launches
// (the workgroup that completes a count resets it); chains: [16][N][2].
struct TabSum {
    double *mix, *chains;
    int *tickets;
};
// sixteen steps of C:269-270 (the marks kernel's three-instruction step; `live` = the EXEC mask to restore)
__device__ __forceinline__ void marks_steps16(double &ph, const double inc, const unsigned long long live) {
    const double c511 = 511.0, cm512 = -512.0;
#pragma unroll
    for (int k = 0; k < 16; k += 4)
        asm volatile("v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4\n\t"
                     "v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4\n\t"
                     "v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4\n\t"
                     "v_add_f64 %0, %0, %1\n\tv_cmpx_le_f64 %2, %0\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, %4"
                     : "+v"(ph)
                     : "v"(inc), "s"(c511), "s"(cm512), "s"(live)
                     : "vcc");
}

// 8 lanes (one sample group of 8 voices) -> per lane the sum over the 8 voices of ONE of 8 samples: the last three levels of
// mxg_lanefold.h's transposing butterfly (row_half_mirror, quad xor 2, quad xor 1)
template <typename T>
__device__ __forceinline__ T fold8(const T (&v)[8], int lane) {
    T l2[4], l3[2];
#pragma unroll
    for (int j = 0; j < 4; j++) l2[j] = fold_dpp<kDppRowHalfMirror, 0xA>(v[2 * j], v[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 2; j++) l3[j] = fold_quad<kDppQuadXor2>(l2[2 * j], l2[2 * j + 1], (lane & 2) != 0);
    return fold_quad<kDppQuadXor1>(l3[0], l3[1], (lane & 1) != 0);
}

// round g (its 8 tables + its header) -> one LDS buffer, 16-byte pieces, lane-linear, by all 512 threads: 4 pieces per lane + a fifth
// for threads 0..83 (the last 128 B of the tables and the 1216 B of the header)
template <int NT>  // threads of the workgroup (512: 4 pieces per lane, 256: 8) + a last one for threads 0..83
__device__ __forceinline__ void round_issue(const double *__restrict__ tables, const double *__restrict__ hdr, size_t g, size_t V,
                                            double *buf) {
    const size_t first = g * kTabVoices;
    const size_t nv = V - first < (size_t)kTabVoices ? V - first : (size_t)kTabVoices;
    const unsigned tbytes = (unsigned)(nv * kTabLen * sizeof(double));  // (the bank's last group may be short)
    const char *src = reinterpret_cast<const char *>(tables + first * kTabLen);
    const char *hsrc = reinterpret_cast<const char *>(hdr + g * kTabHdr);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *dst = reinterpret_cast<char *>(buf) + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2048 / NT; i++) {
        unsigned off = (unsigned)(i * NT + (int)threadIdx.x) * 16u;
        if (off >= tbytes) off = 0;  // (a short last group: re-read its first bytes)
        // (aux = 2: non-temporal -- every table byte is read once by one CU; MI355X_MICROARCH.md "nt-weights")
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                         (__attribute__((address_space(3))) void *)(dst + i * (NT * 16)), 16, 0, MXG_TAB_AUX);
    }
    if (threadIdx.x < 84) {  // bytes [32768, 34112) of the buffer
        const unsigned off = 32768u + threadIdx.x * 16u;
        const char *p = off < 8u * kTabLen * 8u ? (off < tbytes ? src + off : src) : hsrc + (off - 8u * kTabLen * 8u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                         (__attribute__((address_space(3))) void *)(dst + 32768), 16, 0, 0);
    }
}

// One lane = (voice u of the round, time part t, half h): 16 samples [32 t + 16 h, 32 t + 16 h + 16) of voice u; the second half first
// advances the recurrence over the first half's 16 samples.  256 lanes = 8 voices x 16 parts x 2 halves = one round; the workgroup's
// 512 lanes render TWO rounds side by side (lanes 0-255 the even rounds of its range, lanes 256-511 the odd ones): two wavefronts per
// SIMD, so that one round's LDS latencies and bank conflicts hide behind the other's arithmetic.  Inside a round the table reads of all
// 16 samples are in flight at once (osc_pipe_*, mxg_osc.h).  Ring of four buffers: two rounds in use, the next two arriving by DMA.
// SIDES (round 6): 2 = a workgroup of 512 lanes renders two rounds side by side, one barrier for both (rounds 4-5); 1 = a workgroup of 256
// lanes renders one round at a time with a ring of two buffers, TWO such workgroups per CU (the same 136 KB of LDS, the same two wavefronts
// per SIMD).  The two workgroups of a CU drift apart, so one of them is always waiting for ITS next round while the other computes: the CU
// has a DMA in flight all the time, where the lockstep pair issued 67 KB at once and then drained it to nothing before issuing again --
// the kernel was bound by the latency of that burst, not by its arithmetic (the kernel without rendering took 101-109 us of its 112-119,
// the kernel without DMA 87: profiles/r06_k1t.md).
template <bool STORE, bool MIX, bool AHEAD, int SIDES>
__global__ __launch_bounds__(256 * SIDES) void osctab_kernel(size_t V, size_t N, const double *__restrict__ tables,
                                                     const double *__restrict__ hdr, double *__restrict__ hold_io,
                                                     double *__restrict__ out, double *__restrict__ rows, TabAhead AH, TabSum SM) {
    constexpr int kRing = 2 * SIDES;  // LDS buffers: the SIDES rounds in use + the next SIDES arriving by DMA
    constexpr int NT = 256 * SIDES;
    __shared__ __attribute__((aligned(16))) double s_buf[kRing * kTabRound];
    const int lane = threadIdx.x & 63;
    const int side = SIDES == 2 ? threadIdx.x >> 8 : 0, tid8 = threadIdx.x & 255;
    const int u = tid8 & 7, h = (tid8 >> 3) & 1, t = tid8 >> 4;
    const size_t groups = (V + kTabVoices - 1) / kTabVoices;
    const size_t per = (groups + gridDim.x - 1) / gridDim.x;
    const size_t g0 = (size_t)blockIdx.x * per;
    const size_t g1 = g0 + per < groups ? g0 + per : groups;
    // the mixdown: every lane keeps the running sums of ITS 16 samples over the voices it meets (voice u of every other round of this
    // workgroup) -- two multiply-adds per sample and round, nothing crosses lanes inside the loop; the lanes of a sample group are
    // folded once, after the last round
    double accL[kMixChunk], accR[kMixChunk];
#pragma unroll
    for (int i = 0; i < kMixChunk; i++) accL[i] = accR[i] = 0.0;
    const size_t nb = (size_t)t * 32 + (size_t)h * 16;  // this lane's first sample
    const bool has_last = N - 1 >= nb && N - 1 < nb + 16;
    const int i_last = (int)((N - 1) & 15);
    // the next block's marks (AHEAD): this lane's voice, its duty in units of 16 steps (unit q: part q / 2, half q % 2; a mark is left
    // at the start of every part), `upi` units per pair of rounds.  A mark waits in a register until the NEXT iteration has issued its DMA,
    // so that the store is long complete when that iteration ends on vmcnt(0).
    const size_t mv = g0 * kTabVoices + threadIdx.x;
    const size_t mvend = g1 * kTabVoices < V ? g1 * kTabVoices : V;
    const bool mlive = AHEAD && g0 < g1 && mv < mvend;
    double mph = 0.0, minc = 0.0, mpend = 0.0;
    int mq = 0, mpend_t = -1, upi = 0;
    double *mh = nullptr;
    if constexpr (AHEAD) {
        const size_t vv = mlive ? mv : 0;
        mph = AH.carry[vv];
        const double f = AH.freq[vv];
        double x = AH.pan ? AH.pan[vv] : 0.0;
        mh = AH.hdr_next + (vv / kTabVoices) * kTabHdr + (vv % kTabVoices);
        minc = 512. / (AH.sr / (f * kChandiv));  // C:269
        if (x > 1) x = 1;  // C:504
        if (x < 0) x = 0;  // C:505
        if (mlive) {
            AH.phase_io[vv] = mph;  // the phase after the block THIS launch renders: what d_phase holds when the call has completed
            mh[kTabParts * kTabVoices] = minc;
            mh[kTabParts * kTabVoices + kTabVoices] = sqrt(1.0 - x);  // two[0] = input*sqrt(1.0-x)   C:506
            mh[kTabParts * kTabVoices + 2 * kTabVoices] = sqrt(x);    // two[1] = input*sqrt(x)       C:507
        }
        const size_t npairs = (g1 - g0 + SIDES - 1) / SIDES;  // iterations of the loop below
        upi = npairs ? (int)((2 * kTabParts + npairs - 1) / npairs) : 2 * kTabParts;
    }
    auto marks_units = [&](int units) {  // (wave-uniform control flow; every lane of the wavefront is active here)
        if constexpr (AHEAD) {
            const unsigned long long live = __builtin_amdgcn_read_exec();
            for (int i = 0; i < units && mq < 2 * kTabParts; i++, mq++) {
                if ((mq & 1) == 0) {
                    if (mpend_t >= 0 && mlive) mh[mpend_t * kTabVoices] = mpend;  // (more than one mark per iteration: small banks only)
                    mpend = mph;
                    mpend_t = mq >> 1;
                }
                const size_t n0 = (size_t)mq * 16;
                if (n0 + 16 <= N) {
                    marks_steps16(mph, minc, live);
                } else {
                    for (size_t n = n0; n < N; n++) {
                        mph += minc;
                        if (mph >= 511) mph -= 512;
                    }
                }
            }
        }
    };
    auto marks_flush = [&]() {
        if constexpr (AHEAD) {
            if (mpend_t >= 0 && mlive) mh[mpend_t * kTabVoices] = mpend;
            mpend_t = -1;
        }
    };
    for (int k = 0; k < SIDES; k++)
        if (g0 + k < g1) round_issue<NT>(tables, hdr, g0 + k, V, s_buf + k * kTabRound);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (size_t gp = g0; gp < g1; gp += SIDES) {  // the rounds (gp, ..., gp + SIDES - 1)
        const int b0 = (int)((gp - g0) % kRing);
        __builtin_amdgcn_s_barrier();  // everybody's pieces of these rounds have landed; everybody is done with the previous rounds' buffers ...
        asm volatile("" ::: "memory");
#if !(MXG_TAB_ABLATE & 2)
        for (int k = 0; k < SIDES; k++)    // ... which the next rounds may now overwrite
            if (gp + SIDES + k < g1) round_issue<NT>(tables, hdr, gp + SIDES + k, V, s_buf + ((b0 + SIDES + k) % kRing) * kTabRound);
#endif
        marks_flush();  // (the mark of the previous iteration: its store travels with this iteration's DMA)
        const size_t g = gp + side;
        if (g < g1 && !(MXG_TAB_ABLATE & 1)) {
        const int b = (MXG_TAB_ABLATE & 2) ? side : (b0 + side) % kRing;
        const double *B = s_buf + b * kTabRound;
        const double *H = B + kTabVoices * kTabLen;
        const size_t first = g * kTabVoices;
        const bool live = first + u < V;
        const int uu = live ? u : 0;  // a surplus lane shadows the group's first voice (a table that IS in the buffer), gain 0
        const double *T = B + uu * kTabLen - 1;  // T[i + 1] == table[i]: the layout osc_pipe_* index
        double ph = H[t * kTabVoices + uu];
        OscPre q;
        q.inc = H[kTabParts * kTabVoices + uu]; q.k = 0.0; q.p1 = 0.0; q.p2 = 0.0;
        double gl = 0.0, gr = 0.0;
        if constexpr (MIX) {
            gl = live ? H[kTabParts * kTabVoices + kTabVoices + uu] : 0.0;
            gr = live ? H[kTabParts * kTabVoices + 2 * kTabVoices + uu] : 0.0;
        }
        if (h) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                ph += q.inc;
                if (ph >= 511) ph -= 512;
            }
        }
        double hd = 0.0, r[kMixChunk];
        OscPipe<kMixChunk> P;
        osc_pipe_phase<MXG_OSC_SINEBUF, kMixChunk>(ph, q, P);
        __builtin_amdgcn_sched_barrier(0);
#ifndef MXG_TAB_SPLIT_READS
#define MXG_TAB_SPLIT_READS 1  // A/B (tools/build_ab.sh): 0 = the two table values of a sample as one ds_read2_b64
#endif
        osc_pipe_fetch<MXG_OSC_SINEBUF, kMixChunk, MXG_TAB_SPLIT_READS ? 4 : 0>(P, T);  // (a voice's own table: no copy to read from)
        __builtin_amdgcn_sched_barrier(0);
        osc_pipe_finish<MXG_OSC_SINEBUF, kMixChunk>(P, r, hd);
        if constexpr (STORE) {
            if (live) {
#pragma unroll
                for (int i = 0; i < kMixChunk; i++)
                    if (nb + i < N) out[(nb + i) * V + first + u] = r[i];
            }
        }
        if (has_last && live) {  // the member `output` after the block: sample N - 1
            double hv = r[0];
#pragma unroll
            for (int i = 1; i < kMixChunk; i++) hv = i == i_last ? r[i] : hv;
            hold_io[first + u] = hv;
        }
        if constexpr (MIX) {  // (samples at or beyond N are summed too and never written out)
#pragma unroll
            for (int i = 0; i < kMixChunk; i++) {
                accL[i] += r[i] * gl;
                accR[i] += r[i] * gr;
            }
        }
        }  // (g < g1)
        marks_units(upi);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's pieces of the next pair have landed
    }
    if constexpr (AHEAD) {
        marks_units(2 * kTabParts);  // (whatever is left: nothing when the duty divided evenly)
        marks_flush();
        if (mlive) AH.carry[mv] = mph;
    }
    if constexpr (MIX) {
        int idx[8];
#pragma unroll
        for (int i = 0; i < 8; i++) idx[i] = i;
        const int slot = fold8<int>(idx, lane);  // which of 8 samples this lane's sums belong to
        double2v pr[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            double L[8], R[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                L[i] = accL[8 * j + i];
                R[i] = accR[8 * j + i];
            }
            pr[j] = double2v{fold8<double>(L, lane), fold8<double>(R, lane)};
        }
        if constexpr (SIDES == 1) {  // one round at a time: this workgroup's row as it is
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const size_t n = nb + 8 * j + slot;
                if (slot >= 0 && n < N) *reinterpret_cast<double2v *>(rows + ((size_t)blockIdx.x * N + n) * 2) = pr[j];
            }
            return;
        }
#if MXG_TAB_SIDEFOLD
        // the two sides of the workgroup (even / odd rounds of its range) meet in LDS -- the table ring is free after the last round
        // -- and ONE row per workgroup leaves, side 0's sum + side 1's: half the rows for the row sum to read.  Every sample n of the
        // block belongs to exactly one lane of each side (n = nb + 8 j + slot), so the meeting place is indexed by n.
        double2v *meet = reinterpret_cast<double2v *>(s_buf);
        __syncthreads();  // every wavefront is out of the ring
        if (side == 1 && slot >= 0) {
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (nb + 8 * j + slot < N) meet[nb + 8 * j + slot] = pr[j];
        }
        __syncthreads();
        if (side == 0 && slot >= 0) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const size_t n = nb + 8 * j + slot;
                if (n < N) {
                    const double2v o = meet[n];
                    double *rp = rows + ((size_t)blockIdx.x * N + n) * 2;
                    if (SM.mix) {  // (read by a workgroup of another XCD in this launch: write-through, like the lists of grains.hip)
                        __hip_atomic_store(rp, pr[j].x + o.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(rp + 1, pr[j].y + o.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        *reinterpret_cast<double2v *>(rp) = double2v{pr[j].x + o.x, pr[j].y + o.y};
                    }
                }
            }
        }
        if (SM.mix) {
            // Cross-XCD hand-off WITHOUT fences (an agent-scope fence writes back and invalidates the XCD's whole L2: measured +55 us per
            // block with one per workgroup): the rows and chains travel as relaxed agent-scope atomics -- write-through stores, loads that
            // never take a stale L2 line -- and a ticket is drawn only after this wavefront's stores have been acknowledged (vmcnt(0)).
            __shared__ int s_last;
            const unsigned G = gridDim.x, w = blockIdx.x % kPartWaves;
            const size_t count = N * 2;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const int members = (int)((G - w + kPartWaves - 1) / kPartWaves);
                s_last = __hip_atomic_fetch_add(SM.tickets + w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
            }
            __syncthreads();
            if (s_last) {
                for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
                    double c = 0.0;
                    for (size_t gb = w; gb < G; gb += 16 * kPartWaves) {  // sixteen rows in flight at a time (a load per addition would be a round trip each)
                        double v[16];
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const size_t g = gb + (size_t)k * kPartWaves;
                            v[k] = __hip_atomic_load(rows + (g < G ? g : (size_t)w) * count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (int k = 0; k < 16; k++)
                            if (gb + (size_t)k * kPartWaves < G) c += v[k];
                    }
                    __hip_atomic_store(SM.chains + (size_t)w * count + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    __hip_atomic_store(SM.tickets + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int nchains = G < (unsigned)kPartWaves ? (int)G : kPartWaves;
                    s_last = __hip_atomic_fetch_add(SM.tickets + kPartWaves, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nchains - 1;
                }
                __syncthreads();
                if (s_last) {
                    for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
                        // (chains of indices no workgroup has are 0.0, as the wavefronts of mix_partials_kernel that meet no row leave them)
                        double v[kPartWaves];
#pragma unroll
                        for (unsigned k = 0; k < (unsigned)kPartWaves; k++)
                            v[k] = __hip_atomic_load(SM.chains + (size_t)(k < G ? k : 0) * count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        double t2 = v[0];
#pragma unroll
                        for (unsigned k = 1; k < (unsigned)kPartWaves; k++) t2 += k < G ? v[k] : 0.0;
                        SM.mix[i] = t2;
                    }
                    if (threadIdx.x == 0) __hip_atomic_store(SM.tickets + kPartWaves, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
#else
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const size_t n = nb + 8 * j + slot;
            if (slot >= 0 && n < N) *reinterpret_cast<double2v *>(rows + (((size_t)blockIdx.x * 2 + side) * N + n) * 2) = pr[j];
        }
#endif
    }
}

}  // namespace
}  // namespace mxg

using namespace mxg;

// knob tab_sides: 0 automatic = 1; 1 = workgroups of 256 lanes, one round at a time, two per CU; 2 = workgroups of 512 lanes, two rounds
// side by side (rounds 4-5)
static int tables_sides() {
    const int k = tune_get("tab_sides");
    return k == 2 ? 2 : 1;
}
static size_t tables_grid(size_t V, int sides) {  // persistent: 136 KB of LDS per CU either way
    const size_t groups = (V + kTabVoices - 1) / kTabVoices;
    const size_t cus = (size_t)device_cus();
    if (sides == 1) return groups < 2 * cus ? groups : 2 * cus;
    const size_t pairs = (groups + 1) / 2;
    return pairs < cus ? pairs : cus;
}
extern "C" size_t mxg_osc_tables_groups(size_t V) {
    if (ensure_init_only()) return 0;
    const int sides = tables_sides();
    return ((MXG_TAB_SIDEFOLD || sides == 1) ? 1 : 2) * tables_grid(V, sides);  // a partial mix row per workgroup
}

namespace {
// the pipelined form's memory, per stream: which block the second header buffer was prepared for
struct TabAheadCtx {
    bool valid = false;
    size_t V = 0, N = 0;
    const double *freq = nullptr, *pan = nullptr;
    double *phase = nullptr;
    int parity = 0;  // the header buffer the NEXT call renders from
};
std::mutex g_tab_mu;
std::map<hipStream_t, TabAheadCtx> g_tab_ctx;
}  // namespace

extern "C" int mxg_osc_render_tables_ex(size_t V, size_t N, const double *d_freq, const double *d_tables, double *d_phase,
                                        double *d_outhold, double *d_out, const double *d_pan, double *d_rows, double *d_mix, int flags,
                                        void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_freq && d_tables && d_phase && d_outhold, "null device pointer");
    MXG_REQUIRE(d_out || d_pan, "nothing to produce: give d_out (the per-voice block), d_pan + d_rows (the mixdown), or both");
    MXG_REQUIRE(!d_pan || d_rows, "the mixdown needs d_rows");
    MXG_REQUIRE(!d_mix || d_pan, "d_mix is the sum of the mixdown's rows: it needs d_pan and d_rows");
    MXG_REQUIRE(N <= (size_t)kTabParts * 32, "blocks of at most 512 samples (16 time parts of 32 samples per voice)");
    MXG_REQUIRE(!(((uintptr_t)d_tables) & 15), "d_tables must be 16-byte aligned");
    MXG_REQUIRE((flags & ~MXG_TABLES_AHEAD) == 0, "unknown flag");
    if (V == 0 || N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const size_t groups = (V + kTabVoices - 1) / kTabVoices;
    int sides = tables_sides();
    if (d_mix) sides = 2;  // (the row sum inside the kernel is written for the side-by-side form)
    const size_t grid = tables_grid(V, sides);
    const double sr = (double)settings().sampleRate;
    // the pipelined form: a workgroup's lanes walk the marks of its own voices -- every workgroup must have at most as many voices as lanes
    const size_t per = (groups + grid - 1) / grid;
    const bool ahead = (flags & MXG_TABLES_AHEAD) && per * kTabVoices <= (size_t)256 * sides;
    double *hdr = nullptr, *hdr2 = nullptr, *carry = nullptr;  // [groups][152] per-stream scratch: marks, increments, gains
    bool fresh = false, fresh2 = false, fresh3 = false;
    if (int s = scratch_get(SCR_OSCTAB_MARKS, st, sizeof(double) * kTabHdr * groups, (void **)&hdr, &fresh)) return s;
    TabAhead AH{nullptr, nullptr, nullptr, nullptr, nullptr, sr};
    std::lock_guard<std::mutex> lock(g_tab_mu);
    TabAheadCtx &C = g_tab_ctx[st];
    if (!ahead) {
        C.valid = false;  // (d_phase holds the phase after the last rendered block: start over from it)
        KernelTimer kt("osctab_marks_kernel", st);
        hipLaunchKernelGGL(osctab_marks_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, N, d_freq, d_pan, d_phase, d_phase, hdr, sr);
    } else {
        if (int s = scratch_get(SCR_OSCTAB_MARKS2, st, sizeof(double) * kTabHdr * groups, (void **)&hdr2, &fresh2)) return s;
        if (int s = scratch_get(SCR_OSCTAB_CARRY, st, sizeof(double) * V, (void **)&carry, &fresh3)) return s;
        const bool match = C.valid && !fresh && !fresh2 && !fresh3 && C.V == V && C.N == N && C.freq == d_freq && C.pan == d_pan && C.phase == d_phase;
        if (!match) {  // nothing prepared for this block: its marks now, the phase after it into the carry array
            C.parity = 0;
            KernelTimer kt("osctab_marks_kernel", st);
            hipLaunchKernelGGL(osctab_marks_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, N, d_freq, d_pan, d_phase, carry, hdr, sr);
        }
        double *cur = C.parity ? hdr2 : hdr, *nxt = C.parity ? hdr : hdr2;
        AH = TabAhead{d_freq, d_pan, carry, d_phase, nxt, sr};
        hdr = cur;
        C.valid = true; C.V = V; C.N = N; C.freq = d_freq; C.pan = d_pan; C.phase = d_phase;
        C.parity ^= 1;
    }
    TabSum SM{nullptr, nullptr, nullptr};
    if (d_mix) {
        char *sumscr = nullptr;  // 17 tickets (in the first 128 bytes: a fixed place whatever N) + [16][N][2] chains
        bool fresh4 = false;
        const size_t cb = sizeof(double) * kPartWaves * N * 2;
        if (int s = scratch_get(SCR_OSCTAB_SUM, st, 128 + cb, (void **)&sumscr, &fresh4)) return s;
        int *tickets = reinterpret_cast<int *>(sumscr);
        if (fresh4) MXG_HIP(hipMemsetAsync(tickets, 0, 128, st));
        SM = TabSum{d_mix, reinterpret_cast<double *>(sumscr + 128), tickets};
    }
    KernelTimer kt("osctab_kernel", st);
#define MXG_TAB_LAUNCH(S, M, A)                                                                                                            \
    if (sides == 2)                                                                                                                        \
        hipLaunchKernelGGL((osctab_kernel<S, M, A, 2>), dim3((unsigned)grid), dim3(512), 0, st, V, N, d_tables, hdr, d_outhold, d_out, d_rows, AH, SM); \
    else                                                                                                                                   \
        hipLaunchKernelGGL((osctab_kernel<S, M, A, 1>), dim3((unsigned)grid), dim3(256), 0, st, V, N, d_tables, hdr, d_outhold, d_out, d_rows, AH, SM)
    if (ahead) {
        if (d_out && d_pan) { MXG_TAB_LAUNCH(true, true, true); }
        else if (d_out) { MXG_TAB_LAUNCH(true, false, true); }
        else { MXG_TAB_LAUNCH(false, true, true); }
    } else {
        if (d_out && d_pan) { MXG_TAB_LAUNCH(true, true, false); }
        else if (d_out) { MXG_TAB_LAUNCH(true, false, false); }
        else { MXG_TAB_LAUNCH(false, true, false); }
    }
#undef MXG_TAB_LAUNCH
    return check_hip(hipGetLastError(), "osctab_kernel launch");
}

extern "C" int mxg_osc_render_tables(size_t V, size_t N, const double *d_freq, const double *d_tables, double *d_phase,
                                     double *d_outhold, double *d_out, const double *d_pan, double *d_rows, void *stream) {
    return mxg_osc_render_tables_ex(V, N, d_freq, d_tables, d_phase, d_outhold, d_out, d_pan, d_rows, nullptr, 0, stream);
}
