// sample.hip -- maxiDelayline and maxiSample (play family) voice banks on gfx950.
//
// Path (reference src/maximilian.cpp, cited as C:line):
//   maxiDelayline::dl C:420-429, dlFromPosition C:431-439;
//   maxiSample::play C:740-747, playOnce C:982-991, playLoop C:960-967, playUntil C:969-978,
//   playAtSpeed C:1060-1075, playOnceAtSpeed C:994-1003, playUntilAtSpeed C:1047-1058,
//   play4 C:884-956, playAtSpeedBetweenPoints C:823-880.
// One lane owns one voice (one delay line / one play head).  All arithmetic is + - * / floor
// and integer indexing => bit-exact.
//
// HBM layout.  Delay memory of a bank is slot-major: mem[slot*V + v], slot < cap.  Voices
// that share (size, start phase) -- the normal case for a bank -- then touch one contiguous
// 512-B row per wavefront per sample (coalesced read-modify-write); voices with different
// phases degrade to a gather but stay correct.  Algorithmic traffic: 8 B in + 8 B ring read +
// 8 B ring write + 8 B out = 32 B/sample (K4, HBM-bound).  The reference embeds a fixed
// 88200*8-slot array in every object (H:273); here the ring is `cap` slots, any size <= cap.
// maxiSample: one shared, read-only sample buffer per bank (L2/Infinity-Cache resident for the
// sizes of the configs); per sample a lane gathers 1-4 neighbouring doubles (K5, 8 B out +
// gather).  The buffer must be valid on [-4, len+5] with zero guards (mxg_sample_upload's layout, mxg_smp.h).
#include "mxg_common.h"
#include "mxg_pace.h"
#include "mxg_advance.h"
#include "mxg_smp.h"

namespace mxg {
namespace {

// PX (dl only): 0 = 8-byte input / output streams; 1 / 2 / 3 = the whole chunks take their input and leave their output as 16-byte pair
// rows (pair_rows_swap / store_pair_rows, mxg_common.h) with plain / write-through / non-temporal stores -- V even, both blocks 16-byte
// aligned (round 4).  The ring keeps its slot-major 8-byte accesses: its rows are per-voice phases, not sample numbers.
template <int MODE, int PX>
__global__ void __launch_bounds__(256) delay_kernel(size_t V, size_t N, const double *__restrict__ in,
                             const int32_t *__restrict__ size, const double *__restrict__ feedback,
                             const int32_t *__restrict__ position, double *__restrict__ mem, int cap,
                             int32_t *__restrict__ phase_io, double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int ph = phase_io[v];
    // size as the ring sees it: the reference's `phase >= size` tests (C:421, C:432-433) with size <= 0 reset the phase
    // on every sample (size 0 here), and a size beyond the bank's capacity -- the reference's own array is 705 600 slots,
    // a bank's is `cap` -- is held to it.  The tests are unsigned so that an uploaded negative phase or position, which
    // the reference would use as a negative array index, restarts at slot 0 instead of writing outside the ring.
    int sz = size[v];
    sz = sz < 0 ? 0 : (sz > cap ? cap : sz);
    const double fb = feedback[v];
    int pos = 0;
    if constexpr (MODE == 1) {
        pos = position[v];
        if ((unsigned)pos >= (unsigned)sz) pos = 0;  // C:433
    }
    const double *ip = in + v;
    double *op = out + v;
    double *m = mem + v;
    size_t n = 0;
    if constexpr (MODE == 0) {
        // Pipelined form: the input and ring reads of chunk k+1 are requested before chunk k's two
        // store streams are issued, so waiting for them is a counted vmcnt and the stores are
        // never drained.  A ring read may run ahead of the ring writes of the chunk before it
        // only if they cannot touch the same slot: every line of the wavefront needs
        // size >= 2U (otherwise the plain loop below renders the block).
        constexpr int U = 8;
        const size_t nfull = N / U;
        if (nfull > 0 && __all(sz >= 2 * U)) {
            double xi0[U], xi1[U], c0[U], c1[U];
            int s0[U], s1[U];
            int ph_prev = ph;
            constexpr int PST = PX == 2 ? 2 : (PX == 3 ? 1 : 0);
            const size_t odd = threadIdx.x & 1, vp = v & ~(size_t)1;
            const double *ipp = in + vp;             // pair rows: the lane reads 16 bytes of row n + (lane & 1) ...
            double *opp = out + odd * V + vp;        // ... and writes 16 bytes of it
            auto request = [&](size_t k, double(&xi)[U], double(&c)[U], int(&sl)[U]) {
                ph_prev = ph;
                if constexpr (PX != 0) {  // (xi[2j], xi[2j+1] hold the RAW 16 bytes until retire swaps them)
#pragma unroll
                    for (int j = 0; j < U / 2; j++) {
                        const size_t nn = k * U + 2 * j + odd;
                        const size_t mm = (nn < N) ? nn : N - 1;  // clamped: no branch
                        const double2v raw = *reinterpret_cast<const double2v *>(ipp + mm * V);
                        xi[2 * j] = raw.x;
                        xi[2 * j + 1] = raw.y;
                    }
                }
#pragma unroll
                for (int i = 0; i < U; i++) {
                    const size_t nn = k * U + i;
                    const size_t mm = (nn < N) ? nn : N - 1;  // clamped: no branch
                    if constexpr (PX == 0) xi[i] = ip[mm * V];
                    if ((unsigned)ph >= (unsigned)sz) ph = 0;  // C:421
                    sl[i] = ph;
                    c[i] = m[(size_t)ph * V];
                    ph += 1;
                }
            };
            auto retire = [&](double(&xi)[U], double(&c)[U], int(&sl)[U]) {
                if constexpr (PX != 0) {
#pragma unroll
                    for (int j = 0; j < U / 2; j++) {
                        const double2v raw = {xi[2 * j], xi[2 * j + 1]};
                        pair_rows_swap(raw, xi[2 * j], xi[2 * j + 1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < U; i++) {
                    m[(size_t)sl[i] * V] = (c[i] * fb) + (xi[i] * fb) * 0.5;  // C:425
                    if constexpr (PX == 0) {
                        *op = c[i];                                            // C:424
                        op += V;
                    }
                }
                if constexpr (PX != 0) {
#pragma unroll
                    for (int j = 0; j < U / 2; j++) {
                        store_pair_rows<PST>(opp, c[2 * j], c[2 * j + 1]);
                        opp += 2 * V;
                    }
                    op += (size_t)U * V;
                }
            };
            request(0, xi0, c0, s0);
#pragma unroll
            for (int i = 0; i < U; i++) {  // keep the prologue's loads out of the loop header's wait
                asm volatile("" : "+v"(xi0[i]));
                asm volatile("" : "+v"(c0[i]));
            }
            size_t k = 0;
            for (; k + 1 < nfull; k += 2) {
                request(k + 1, xi1, c1, s1);
                retire(xi0, c0, s0);
                request(k + 2, xi0, c0, s0);
                retire(xi1, c1, s1);
            }
            if (k < nfull) {
                request(k + 1, xi1, c1, s1);
                retire(xi0, c0, s0);
            }
            ph = ph_prev;  // the last request ran ahead speculatively (loads only)
            n = nfull * U;
            ip += n * V;
        }
    }
    for (; n < N; n++) {
        double input = *ip;
        if ((unsigned)ph >= (unsigned)sz) ph = 0;  // C:421 / C:432
        double *slot = m + (size_t)ph * V;
        double cur = *slot;
        double o;
        if constexpr (MODE == 0) {
            o = cur;                                 // C:424
            *slot = (cur * fb) + (input * fb) * 0.5;  // C:425
        } else {
            // the read slot may alias the slot written earlier in this block by this voice:
            // plain loads/stores of one lane to one address stay ordered.
            o = m[(size_t)pos * V];                       // C:434
            *slot = (cur * fb) + (input * fb) * kChandiv;  // C:435
        }
        ph += 1;
        *op = o;
        ip += V;
        op += V;
    }
    phase_io[v] = ph;
}

struct SmpArgs {
    size_t V, N;
    const double *amp;
    size_t len;
    double step_div, sr;
    const double *a;      // speed / frequency, [V] or [N][V] (XMOD)
    const double *trig;   // [N][V], modes 9-14
    const double *start, *end;  // [V] or null; modes 11-13: p0 / p1
    double *position;     // [V] in/out (unused by mode 14)
    double *tprev;        // [V] in/out, modes 9-14
    int32_t *tfirst;      // [V] in/out, modes 9-14
    double *out;
    int px_store;         // pair-row store flavour (smp_emit), 0 = 8-byte stores
    unsigned pace_arg;    // the paced schedule (mxg_pace.h): the starting / fixed period in ticks of 10 ns per chunk, 0 = not paced
    unsigned *pace_ctl;   // ... and its controller's words (play() on whole heads), or null
};

// PX: the whole chunks leave as 16-byte pair rows (emit_chunk, mxg_common.h) -- V even, out 16-byte aligned (round 4; same values, same
// order per voice); SmpArgs::px_store: 1 / 2 / 3 = plain / write-through / non-temporal stores.
template <int MODE, bool XMOD, bool PX>
__global__ void __launch_bounds__(256) sample_kernel(SmpArgs A) {
    const size_t V = A.V, N = A.N;
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    constexpr bool TRIG = MODE >= 9;
    Smp s = {A.amp, A.len, (MODE == 14) ? 0.0 : A.position[v], A.step_div, 0.0, false, 0.0, 0.0};
    double st = 0.0, en = 1.0;
    if constexpr (TRIG) {
        s.tprev = A.tprev[v];
        s.tfirst = A.tfirst[v] != 0;
        s.p0 = A.start ? A.start[v] : 0.0;
        s.p1 = A.end ? A.end[v] : 0.0;
        if constexpr (MODE == 12) en = s.p0 + s.p1;  // playUntilAtSpeed(offset+length, speed) C:1034
    } else {
        st = A.start ? A.start[v] : 0.0;
        en = A.end ? A.end[v] : 1.0;
    }
    const double *amp = A.amp;
    const double sr = A.sr;
    const double x0 = A.a ? A.a[v] : 1.0;
    const double *ap = A.a ? A.a + v : nullptr;
    const double *tp = TRIG ? A.trig + v : nullptr;
    double *op = A.out + v;
    using Req = SmpReq<MODE>;
    constexpr int L = Req::L;
    constexpr int U = (L == 4) ? 4 : 8;
    constexpr bool xmod = XMOD;  // per-sample speed input
    const size_t nfull = N / U;

    if constexpr (MODE == 0) {
        // play() (C:740-747) with no wrap inside this block for any line of the wavefront: the head just walks
        // pos, pos+1, ..., so a line's 8 samples of a chunk are 64 contiguous bytes.  Four 16-B loads per chunk
        // (instead of eight 8-B gathers) request every cache line once instead of relying on the 32 KB L1 to
        // keep 64 lines per wavefront alive between samples.  Same values, same head afterwards.
        // (round 6: heads BETWEEN two elements too -- what playAtSpeed leaves behind.  The index is (long)pos and pos grows by exactly 1.0
        // per sample, so sample k reads element p0 + k as long as no addition rounds the head up to the next integer: below 2^31 an ulp is
        // at most 2^-22, and a fraction of at most 1 - 2^-20 survives every binade crossing of the block.  The head after the block is the
        // exact multi-step sum of mxg_advance.h, the reference's one addition per sample.)
        const long long p0 = (long long)s.pos;
        const double frac0 = s.pos - (double)p0;
        const bool straight = s.pos >= 0.0 && frac0 <= 1.0 - 0x1p-20 && (size_t)p0 + N + 1 < A.len && A.len < ((size_t)1 << 31);
        if (nfull > 0 && __all(straight)) {
            const double *src = amp + p0;
            double2v a0[4], a1[4];
            auto request = [&](size_t k, double2v(&d)[4]) {
                const size_t kk = (k < nfull) ? k : nfull - 1;  // clamped: the run-ahead re-reads the last chunk
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const double *q = src + kk * U + 2 * j;  // 8-B aligned only: two 8-B halves of one 16-B request
                    d[j] = *reinterpret_cast<const double2v *>(q);
                }
            };
            // the paced schedule (mxg_pace.h): this path is a pure store stream (its loads hit the L2) -- 49.7 us free-running, 38.6 us
            // on a period of 56 ticks at 65 536 voices (profiles/r06_pace.md)
            Pace pc;
            pc.start(A.pace_ctl, A.pace_arg);
            auto retire = [&](double2v(&d)[4]) {
                const double o[8] = {d[0].x, d[0].y, d[1].x, d[1].y, d[2].x, d[2].y, d[3].x, d[3].y};
                pc.wait(true);
                emit_chunk<PX>(op, V, o, A.px_store);
                __builtin_amdgcn_sched_barrier(0);
            };
            request(0, a0);
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(a0[j]));
            size_t k = 0;
            for (; k + 1 < nfull; k += 2) {
                request(k + 1, a1);
                retire(a0);
                request(k + 2, a0);
                retire(a1);
            }
            if (k < nfull) retire(a0);
            if (__all(frac0 == 0.0)) {
                s.pos = s.pos + (double)(nfull * U);  // exact: integers below 2^53, one addition per sample in the reference
            } else {
                size_t left = nfull * U;
                while (left > 0) {
                    bool crossed;
                    left -= (size_t)advance_until(s.pos, 1.0, HUGE_VAL, true, left > (size_t)(1 << 30) ? (1 << 30) : (int)left, crossed);
                }
            }
            for (size_t n = nfull * U; n < N; n++) {
                *op = amp[(long long)s.pos];
                op += V;
                s.pos += 1.0;
            }
            A.position[v] = s.pos;
            if (threadIdx.x == 0) pc.finish(A.pace_ctl, A.pace_arg, blockIdx.x, gridDim.x);
            return;
        }
        if (threadIdx.x == 0 && A.pace_ctl) {  // (a workgroup that takes the general path still hands in its ticket: not a store-bound launch)
            Pace none;
            none.start(A.pace_ctl, A.pace_arg);
            none.finish(A.pace_ctl, A.pace_arg, blockIdx.x, gridDim.x);
        }
    }
    if (nfull > 0) {
        // Two register sets used alternately (the loop is unrolled by two) so that no loaded value
        // is ever copied: a copy would be a use, and a use is a wait.  x and trig are requested
        // two chunks ahead (the head must advance through chunk k+1 before its gathers can be
        // issued), the gathers one chunk ahead.
        double x0s[U], x1s[U], t0s[U], t1s[U];
        Req r0[U], r1[U];
        double v0[U][L], v1[U][L];
#pragma unroll
        for (int i = 0; i < U; i++) {
            const size_t m = ((size_t)(U + i) < N) ? (size_t)(U + i) : N - 1;
            x0s[i] = xmod ? ap[(size_t)i * V] : x0;
            x1s[i] = xmod ? ap[m * V] : x0;
            t0s[i] = TRIG ? tp[(size_t)i * V] : 0.0;
            t1s[i] = TRIG ? tp[m * V] : 0.0;
        }
        // consume the prologue's loads here, so that the loop header does not inherit a pending
        // load it would have to wait for with vmcnt(0) on every iteration
#pragma unroll
        for (int i = 0; i < U; i++) {
            if constexpr (xmod) asm volatile("" : "+v"(x1s[i]));
            if constexpr (TRIG) asm volatile("" : "+v"(t1s[i]));
        }
        double pos_prev = s.pos, tprev_prev = s.tprev;
        bool tfirst_prev = s.tfirst;
#pragma unroll
        for (int i = 0; i < U; i++) {
            smp_gen<MODE>(s, x0s[i], t0s[i], st, en, sr, r0[i]);
#pragma unroll
            for (int l = 0; l < L; l++) v0[i][l] = amp[r0[i].idx[l]];
        }
#pragma unroll
        for (int i = 0; i < U; i++)
#pragma unroll
            for (int l = 0; l < L; l++) asm volatile("" : "+v"(v0[i][l]));
        auto stage = [&](size_t k, double(&xuse)[U], double(&xload)[U], double(&tuse)[U],
                         double(&tload)[U], Req(&rcur)[U], double(&vcur)[U][L], Req(&rnext)[U],
                         double(&vnext)[U][L]) {
            if constexpr (xmod || TRIG) {  // inputs of chunk k+2 (clamped index: no branch)
#pragma unroll
                for (int i = 0; i < U; i++) {
                    const size_t n = (k + 2) * U + i;
                    const size_t m = (n < N) ? n : N - 1;
                    if constexpr (xmod) xload[i] = ap[m * V];
                    if constexpr (TRIG) tload[i] = tp[m * V];
                }
            }
            // advance the head through chunk k+1 and request its gathers.  After the last full
            // chunk this runs ahead speculatively: *_prev keep the state to carry.
            pos_prev = s.pos;
            tprev_prev = s.tprev;
            tfirst_prev = s.tfirst;
#pragma unroll
            for (int i = 0; i < U; i++) {
                smp_gen<MODE>(s, xuse[i], tuse[i], st, en, sr, rnext[i]);
#pragma unroll
                for (int l = 0; l < L; l++) vnext[i][l] = amp[rnext[i].idx[l]];
            }
            double o[U];
#pragma unroll
            for (int i = 0; i < U; i++) o[i] = smp_eval<MODE>(rcur[i], vcur[i]);
            emit_chunk<PX>(op, V, o, A.px_store);
            // keep the next stage's arithmetic on this stage's input loads from being scheduled
            // up here, in front of the stores (it would have to wait for loads just issued)
            __builtin_amdgcn_sched_barrier(0);
        };
        size_t k = 0;
        for (; k + 1 < nfull; k += 2) {
            stage(k, x1s, x0s, t1s, t0s, r0, v0, r1, v1);
            stage(k + 1, x0s, x1s, t0s, t1s, r1, v1, r0, v0);
        }
        if (k < nfull) stage(k, x1s, x0s, t1s, t0s, r0, v0, r1, v1);
        s.pos = pos_prev;
        s.tprev = tprev_prev;
        s.tfirst = tfirst_prev;
    }
    // ragged tail (< U samples): one sample at a time
    for (size_t n = nfull * U; n < N; n++) {
        const double x = xmod ? ap[n * V] : x0;
        const double t = TRIG ? tp[n * V] : 0.0;
        Req q;
        double val[L];
        smp_gen<MODE>(s, x, t, st, en, sr, q);
#pragma unroll
        for (int l = 0; l < L; l++) val[l] = amp[q.idx[l]];
        *op = smp_eval<MODE>(q, val);
        op += V;
    }
    if constexpr (MODE != 14) A.position[v] = s.pos;
    if constexpr (TRIG) {
        A.tprev[v] = s.tprev;
        A.tfirst[v] = s.tfirst ? 1 : 0;
    }
}

// ---- time parts for the *AtSpeed players ---------------------------------------------------------------------------------
// 65 536 voices are one wavefront per SIMD, and the speed players spend ~35 VALU instructions per sample on the head
// (floor, remainder, bounds, wrap) plus two gathers: with nothing else resident every latency and every issue slot is
// exposed (87 us per 65 536 x 512 block against 42 us for the stores; staging the gathers through LDS windows was measured
// and changed nothing -- the texture path is not the limit).  So a block is cut into gridDim.y time parts, like K1's
// sinewave: part p first moves the head over the samples before it WITHOUT rendering them.  The head's recurrence is
// pos <- fl(pos + step) with, for playAtSpeed, pos -= len once it reaches len (C:1071-1073); advance_until (mxg_advance.h,
// the exact multi-step form written for the grain schedulers and fuzzed against the one-step recurrence on the host)
// advances it by hundreds of samples at a time without changing a bit.  It needs step > 0 and pos >= 0: every part tests
// the same initial state, and a wavefront with any other voice is rendered whole by part 0.  The last part (or part 0)
// stores the head.  Block-constant speed only; per-sample speed inputs take sample_kernel.
//
// More resident wavefronts alone made it SLOWER (96 -> 125 us): every lane keeps its own cache line alive for ~16 samples, and
// 16 wavefronts x 64 lines are four times the 32 KB L1.  So the parts do not gather through the L1 at all when they can avoid
// it: the indices of one voice over a chunk of 8 samples are local (for |speed| < 1.85 they fall inside 16 consecutive
// doubles), and the wavefront fetches each voice's window cooperatively -- 8 lanes x 16 bytes cover one voice's 128 bytes, one
// instruction covers 8 voices (8-16 lines instead of 64), 8 instructions the wavefront; the pieces go to LDS (144-byte rows)
// and every lane picks its own values with ds_read.  Whether a chunk qualifies (every lane local, no wrap inside the chunk) is
// one ballot per chunk; the others take per-lane gathers.  Same values either way.  (The window path alone, at one wavefront
// per SIMD, had changed nothing either: 87 us.  It takes both.)
constexpr int kRowDoubles = kSmpWindow + 2;  // 144-byte rows: 16-byte aligned pieces, consecutive voices 4 banks apart

__device__ __forceinline__ void smp_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int B>
__device__ __forceinline__ void smp_skip(double &pos, const double step, const double dlen, size_t n) {
    while (n > 0) {
        bool crossed;
        const int kmax = n > (size_t)(1 << 30) ? (1 << 30) : (int)n;
        n -= (size_t)advance_until(pos, step, B == 4 ? dlen : HUGE_VAL, true, kmax, crossed);
        if (B == 4 && crossed) pos -= dlen;  // C:1072-1073
    }
}

#ifndef MXG_SMP_WPE
#define MXG_SMP_WPE 3  // A/B (tools/build_ab.sh): wavefronts per SIMD the time-part kernel is compiled for
#endif
template <int MODE, bool PIPE, bool PX, bool RING>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RING ? 2 : MXG_SMP_WPE))) sample_parts_kernel(SmpArgs A, const size_t part_len, PartSync psync) {
    __shared__ double s_win[4 * 64 * kRowDoubles];
    const size_t V = A.V, N = A.N;
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    constexpr int B = smp_base(MODE);
    static_assert(B >= 4 && B <= 6, "the *AtSpeed players");
    Smp s = {A.amp, A.len, A.position[v], A.step_div, 0.0, false, 0.0, 0.0};
    const double st = A.start ? A.start[v] : 0.0, en = A.end ? A.end[v] : 1.0;
    const double x0 = A.a[v], sr = A.sr;
    const double step = (x0 * kChandiv) / s.step_div;  // the increment of smp_gen (C:1070)
    const bool can_skip = __all(step > 0.0 && step < HUGE_VAL && s.pos >= 0.0 && s.pos < HUGE_VAL);
    Pace pc;
    pc.start(nullptr, A.pace_arg);
    size_t n0 = 0, n1 = N;
    // the writer of the head is ALWAYS the last part (the last dispatched: it never waits for work that is not resident yet) --
    // where the wavefront cannot skip, the last part renders it whole; the other parts tell it when they have read the head
    // (part_signal / part_wait, mxg_common.h), also the parts that only looked at it to decide can_skip
    const bool writer = blockIdx.y + 1 == gridDim.y;
    int *const part_ctr = gridDim.y > 1 ? part_counter(psync) : nullptr;
    if (!writer) part_signal(part_ctr);
    if (can_skip) {
        n0 = (size_t)blockIdx.y * part_len;
        n1 = n0 + part_len < N ? n0 + part_len : N;
        smp_skip<B>(s.pos, step, (double)s.len, n0);
    } else if (!writer) {
        return;
    }
    const double *amp = A.amp;
    double *op = A.out + n0 * V + v;
    using Req = SmpReq<MODE>;
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    double *win = s_win + (threadIdx.x >> 6) * (64 * kRowDoubles);
    double *row = win + lane * kRowDoubles;
    const bool can_stage = __popcll(__ballot(true)) == 64 && A.len < ((size_t)1 << 30);  // full wavefront, 32-bit indices
    // One chunk of U samples, in the two halves a software pipeline needs (loads and stores retire in order on ONE counter, so
    // a wait for a load also waits for every store issued before it -- and a store takes microseconds when HBM's write queues are
    // full; the loads of chunk k+1 are therefore issued BEFORE the stores of chunk k and waited for after them):
    //   fetch    advance the head over the chunk (smp_gen), decide window / gathers, issue the loads into c[];
    //   land     put c[] into this wavefront's LDS rows (the wait for the loads is here);
    //   render   every lane reads its taps from its row, interpolates and stores.
    // A chunk whose indices do not fit a window is gathered per lane and parked in the lane's own row (pairs at 2i, 2i+1), so
    // `render` is the same code either way.
    struct Chunk {
        double rem[U];
        int tap[U];   // offset of the first tap in the lane's row; the second is the next double (ring: the next double modulo 16)
        unsigned slots;  // ring chunks, loader view: nibble j = 8 | the ring slot (0..7) piece j of this lane goes to, 0 = nothing to load
        unsigned ok;  // bit i: the reference's bounds test of sample i
        bool staged;
        bool ring;    // wave-uniform: the rows are used as rings (below)
        bool allok;   // wave-uniform: every sample of every lane passed it (no select needed)
    };
    // Round 6 -- the row as a RING of eight 16-byte pieces (absolute piece p = sample index / 2 lives in slot p & 7, so sample index i is
    // at double i & 15 of the row): consecutive chunks of a forward-moving head overlap by half a window, and re-fetching the overlap
    // is harmless while the sample sits in L2 but DOUBLES the HBM traffic of a bank whose heads are spread over gigabytes (every
    // XCD streams its L2's worth of other voices' windows between two chunks of one wavefront: FETCH_SIZE 563 MB per 268 MB block,
    // profiles/r06_sample_bank.md).  A smooth chunk (forward, no wrap, every lane's span within eight pieces) therefore loads only the
    // pieces behind the last one it holds: `ring_end` = one past that piece, per voice; a chunk of any other kind uses the rows the
    // old way and forgets the ring.  The (first, end) pair of pieces a voice wants travels to its eight loader lanes through the two
    // padding doubles of its own row.
    int ring_end = -(1 << 30);
    double2v c[U];  // the loads in flight: only ever one chunk's (fetched after the previous one has landed)
    // The head in 32 bits.  A part that skipped (step > 0, head >= 0) with the sample and the head below 2^30 needs none of the
    // 64-bit conversions of smp_gen: (long long)pos == (int)pos, the bounds tests compare ints, and playAtSpeed's wrap
    // `if ((size_t)(long long)pos >= len) pos -= len` (C:1072-1073) has pos in [len, 2 len) where it fires (step < len), so the
    // subtraction is exact (Sterbenz) and the new integer part is the old one minus len: ONE v_cvt_i32_f64 per sample, carried.
    const int len32 = (int)(A.len < ((size_t)1 << 30) ? A.len : 0);
    const double dlen = (double)A.len;
    bool fast = can_skip && len32 > 0;
    if constexpr (B == 4) fast = fast && __all(step < dlen && s.pos + step < 2.0 * dlen);
    else fast = fast && __all(s.pos + step * (double)(n1 - n0 + 2) < 1073741824.0);
    const double endc = en > 1.0 ? 1.0 : en;  // C:1050
    int icur = fast ? (int)s.pos : 0;
    auto fetch = [&](Chunk &C) {
        Req r[U];
        // Most chunks cross nothing: the head only moves forward, so if the LAST sample of the chunk still passes the bounds
        // test and the head has not reached `len` after it, no sample of the chunk wrapped or failed.  Such a chunk is the bare
        // recurrence (5 instructions per sample instead of 17); it is computed first, on copies, and kept if every lane agrees.
        bool smooth = false;
        if (fast) {
            double p = s.pos;
            int ic = icur, ilast = 0;
#pragma unroll
            for (int i = 0; i < U; i++) {
                ilast = ic;
                r[i].rem = p - (double)ic;
                r[i].idx[0] = (B == 5) ? ic : ic + 1;
                r[i].ok = true;
                p = p + step;
                ic = (int)p;
            }
            bool good;
            if constexpr (B == 4) good = ic < len32;
            else if constexpr (B == 5) good = ilast + 1 < len32;
            else good = (double)ilast < dlen * endc;
            smooth = __all(good);
            if (smooth) {
                s.pos = p;
                icur = ic;
            }
        }
        if (smooth) {
#pragma unroll
            for (int i = 0; i < U; i++) r[i].idx[1] = r[i].idx[0] + 1;
        } else if (fast) {
#pragma unroll
            for (int i = 0; i < U; i++) {
                const int ip = icur;
                const double di = (double)ip;
                r[i].rem = s.pos - di;
                int first = ip + 1;
                if constexpr (B == 4) r[i].ok = ip < len32;
                if constexpr (B == 5) {
                    r[i].ok = ip + 1 < len32;
                    first = ip;
                }
                if constexpr (B == 6) {
                    r[i].ok = di < dlen * endc;
                    first = 1 + (ip < len32 + 2 ? ip : len32 + 2);  // 1 + (long long)smp_safe_head(pos, len) for pos >= 0
                }
                const int i0 = r[i].ok ? first : 0;
                r[i].idx[0] = i0;
                r[i].idx[1] = i0 + 1;
                s.pos = s.pos + step;
                icur = (int)s.pos;
                if constexpr (B == 4) {
                    const bool wrap = icur >= len32;
                    s.pos = wrap ? s.pos - dlen : s.pos;
                    icur = wrap ? icur - len32 : icur;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < U; i++) smp_gen<MODE>(s, x0, 0.0, st, en, sr, r[i]);
        }
        C.ok = 0;
#pragma unroll
        for (int i = 0; i < U; i++) {
            C.rem[i] = r[i].rem;
            C.ok |= r[i].ok ? (1u << i) : 0u;
        }
        C.staged = false;
        C.ring = false;
        if (RING && can_stage && smooth) {
            const int plo = (int)r[0].idx[0] >> 1, phi = ((int)r[U - 1].idx[0] + 1) >> 1;  // pieces of the first / the last tap
            C.ring = __all(phi - plo < 8) && A.len < ((size_t)1 << 30);
            if (C.ring) {
                // at least the last piece is (re)loaded, so that the count fits three bits: first piece * 8 + (pieces - 1) in one word
                int first = ring_end > plo ? ring_end : plo;
                first = first < phi ? first : phi;
                const unsigned want = ((unsigned)first << 3) | (unsigned)(phi - first);
                ring_end = phi + 1;
                C.slots = 0;
#pragma unroll
                for (int j = 0; j < U; j++) {  // piece (lane & 7) of what voice 8j + lane/8 still needs
                    const unsigned w = (unsigned)__shfl((int)want, 8 * j + (lane >> 3));
                    const bool need = (unsigned)(lane & 7) <= (w & 7u);
                    // (a lane with nothing to fetch repeats the voice's last piece -- the line its neighbour requests anyway -- so that c[]
                    // is overwritten as a whole: a predicated load would keep the previous chunk's 32 registers alive through this function)
                    const unsigned pc = (w >> 3) + (need ? (unsigned)(lane & 7) : (w & 7u));
                    C.slots |= need ? (8u | (pc & 7u)) << (4 * j) : 0u;
                    c[j] = *reinterpret_cast<const double2v *>(amp + 2 * (size_t)pc);
                    C.tap[j] = (int)r[j].idx[0] & (kSmpWindow - 1);
                }
                C.staged = true;
                C.allok = true;
                return;
            }
        }
        ring_end = -(1 << 30);
        int base = 0;
        if (can_stage) {
            int lo = (int)r[0].idx[0], hi = (int)r[U - 1].idx[0];  // idx[1] = idx[0] + 1 for these players (mxg_smp.h)
            if (!smooth) {  // (a smooth chunk's indices ascend)
                hi = lo;
#pragma unroll
                for (int i = 1; i < U; i++) {
                    const int ix = (int)r[i].idx[0];
                    lo = ix < lo ? ix : lo;
                    hi = ix > hi ? ix : hi;
                }
            }
            C.staged = __all(hi + 1 - lo < kSmpWindow);
            base = lo;
        }
        C.allok = smooth;
        if (C.staged) {
#pragma unroll
            for (int j = 0; j < U; j++) {  // piece j: 16 bytes of the window of voice 8j + lane/8
                const int b = __shfl(base, 8 * j + (lane >> 3));
                c[j] = *reinterpret_cast<const double2v *>(amp + b + 2 * (lane & 7));
                C.tap[j] = (int)r[j].idx[0] - base;
            }
        } else {
#pragma unroll
            for (int i = 0; i < U; i++) {
                c[i].x = amp[r[i].idx[0]];
                c[i].y = amp[r[i].idx[1]];
                C.tap[i] = 2 * i;
            }
        }
    };
    auto land = [&](const Chunk &C) {
        if (RING && C.ring) {
#pragma unroll
            for (int j = 0; j < U; j++) {
                const unsigned nb = (C.slots >> (4 * j)) & 15u;
                if (nb) *reinterpret_cast<double2v *>(win + (8 * j + (lane >> 3)) * kRowDoubles + 2 * (nb & 7u)) = c[j];
            }
            smp_lds_sync();
            return;
        }
        if (C.staged) {
#pragma unroll
            for (int j = 0; j < U; j++)
                *reinterpret_cast<double2v *>(win + (8 * j + (lane >> 3)) * kRowDoubles + 2 * (lane & 7)) = c[j];
        } else {
#pragma unroll
            for (int i = 0; i < U; i++) *reinterpret_cast<double2v *>(row + 2 * i) = c[i];
        }
        smp_lds_sync();
    };
    auto render = [&](const Chunk &C) {
        Req q;
        double o[U];
        if (C.allok) {
            const int wrap = (RING && C.ring) ? kSmpWindow - 1 : 2 * kSmpWindow - 1;  // (a ring's second tap wraps at 16; a window's never reaches it)
#pragma unroll
            for (int i = 0; i < U; i++) {
                const double val[2] = {row[C.tap[i]], row[(C.tap[i] + 1) & wrap]};
                q.rem = C.rem[i];
                q.ok = true;
                o[i] = smp_eval<MODE>(q, val);
            }
        } else {
#pragma unroll
            for (int i = 0; i < U; i++) {
                const double val[2] = {row[C.tap[i]], row[C.tap[i] + 1]};
                q.rem = C.rem[i];
                q.ok = (C.ok >> i) & 1u;
                o[i] = smp_eval<MODE>(q, val);
            }
        }
        pc.wait(true);
        emit_chunk<PX>(op, V, o, A.px_store);
        smp_lds_sync();  // the rows are free again
    };
    size_t n = n0;
    const size_t nchunks = (n1 - n0) / U;
    bool lean_done = false;
    if constexpr (RING && PIPE) {
        // ---- the lean part (round 6, the HBM-resident bank): TWO chunks of window loads in flight -------------------------------------------
        // Without its window loads this kernel takes 64 of its 127 us on the 8.6 GB bank, and 59 % of a resident wavefront's cycles are
        // spent parked on s_waitcnt (profiles/r06_sample_bank.md): with one chunk in flight per wavefront it is bound by the LATENCY of its
        // loads.  A second chunk in flight needs a second set of load registers and a third set of per-chunk metadata -- 256 registers
        // and spills with the general chunk (rem[8], tap[8], flags per chunk).  So a part whose chunks are ALL smooth ring chunks for every
        // lane of the wavefront -- known up front: forward, step <= 1.7 (at most eight pieces per chunk), no bounds test fails and nothing
        // wraps before the part's last whole chunk -- carries per chunk only the head it started from: render replays the eight additions
        // (the same operations: the same remainders and indices) instead of reading them back.
        bool lean = fast && can_stage && nchunks >= 3;
        if (lean) {
            const double pend = s.pos + step * (double)(nchunks * U + 1);  // beyond the head after the part's last whole chunk
            bool ok = step <= 1.7;
            if constexpr (B == 4) ok = ok && pend + 2.0 < dlen;
            if constexpr (B == 5) ok = ok && pend + 3.0 < dlen;
            if constexpr (B == 6) ok = ok && pend + 2.0 < dlen * endc && pend + 4.0 < dlen;
            lean = __all(ok);
        }
        if (lean) {
            struct Lean {
                double p0;       // the head at the chunk's first sample
                int ic0;         // its integer part
                unsigned slots;  // loader view, as Chunk::slots
            };
            double2v cA[U], cB[U], cC[U];
            auto lfetch = [&](Lean &Lc, double2v (&cc)[U]) {
                Lc.p0 = s.pos;
                Lc.ic0 = icur;
                double ph = s.pos;
                int ic = icur, ilast = icur;
#pragma unroll
                for (int i = 0; i < U; i++) {
                    ilast = ic;
                    ph = ph + step;
                    ic = (int)ph;
                }
                const int i_first = (B == 5) ? icur : icur + 1;             // first tap of the first sample (mxg_smp.h, B = 4 / 5 / 6)
                const int i_last = ((B == 5) ? ilast : ilast + 1) + 1;      // second tap of the last
                s.pos = ph;
                icur = ic;
                const int plo = i_first >> 1, phi = i_last >> 1;
                int first = ring_end > plo ? ring_end : plo;
                first = first < phi ? first : phi;
                const unsigned want = ((unsigned)first << 3) | (unsigned)(phi - first);
                ring_end = phi + 1;
                Lc.slots = 0;
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const unsigned w = (unsigned)__shfl((int)want, 8 * j + (lane >> 3));
                    const bool need = (unsigned)(lane & 7) <= (w & 7u);
                    const unsigned pc = (w >> 3) + (need ? (unsigned)(lane & 7) : (w & 7u));
                    Lc.slots |= need ? (8u | (pc & 7u)) << (4 * j) : 0u;
                    cc[j] = *reinterpret_cast<const double2v *>(amp + 2 * (size_t)pc);
                }
            };
            auto lland = [&](const Lean &Lc, const double2v (&cc)[U]) {
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const unsigned nb = (Lc.slots >> (4 * j)) & 15u;
                    if (nb) *reinterpret_cast<double2v *>(win + (8 * j + (lane >> 3)) * kRowDoubles + 2 * (nb & 7u)) = cc[j];
                }
                smp_lds_sync();
            };
            auto lrender = [&](const Lean &Lc) {
                double ph = Lc.p0;
                int ic = Lc.ic0;
                double o[U];
                Req q;
#pragma unroll
                for (int i = 0; i < U; i++) {
                    const int t0 = ((B == 5) ? ic : ic + 1) & (kSmpWindow - 1);
                    const double val[2] = {row[t0], row[(t0 + 1) & (kSmpWindow - 1)]};
                    q.rem = ph - (double)ic;
                    q.ok = true;
                    o[i] = smp_eval<MODE>(q, val);
                    ph = ph + step;
                    ic = (int)ph;
                }
                pc.wait(true);
                emit_chunk<PX>(op, V, o, A.px_store);
                smp_lds_sync();  // the rows are free again
            };
#ifndef MXG_SMP_LEAN_DEPTH
#define MXG_SMP_LEAN_DEPTH 2  // chunks of window loads in flight per wavefront in a lean part (A/B: 3 -- 246 registers, measured the same gain as 2)
#endif
#if MXG_SMP_LEAN_DEPTH == 3
            double2v cD[U];
            Lean L0, L1, L2, L3;
            lfetch(L0, cA);
            lland(L0, cA);
            lfetch(L1, cB);
            lfetch(L2, cC);  // (nchunks >= 3)
            size_t k = 0;  // L0 holds chunk k (landed); L1's and L2's loads are in flight
#define MXG_LEAN_STEP(LN, CN, LR, LL, CL)                 \
    if (k + 3 < nchunks) lfetch(LN, CN);                  \
    lrender(LR);                                          \
    if (k + 1 >= nchunks) break;                          \
    lland(LL, CL);                                        \
    k++;
            while (true) {
                MXG_LEAN_STEP(L3, cD, L0, L1, cB)
                MXG_LEAN_STEP(L0, cA, L1, L2, cC)
                MXG_LEAN_STEP(L1, cB, L2, L3, cD)
                MXG_LEAN_STEP(L2, cC, L3, L0, cA)
            }
#undef MXG_LEAN_STEP
#else
            Lean L0, L1, L2;
            lfetch(L0, cA);
            lland(L0, cA);
            lfetch(L1, cB);
            size_t k = 0;  // L0 holds chunk k (landed), L1's loads are in flight
            while (true) {
                if (k + 2 < nchunks) lfetch(L2, cC);
                lrender(L0);
                if (k + 1 >= nchunks) break;
                lland(L1, cB);
                k++;
                if (k + 2 < nchunks) lfetch(L0, cA);
                lrender(L1);
                if (k + 1 >= nchunks) break;
                lland(L2, cC);
                k++;
                if (k + 2 < nchunks) lfetch(L1, cB);
                lrender(L2);
                if (k + 1 >= nchunks) break;
                lland(L0, cA);
                k++;
            }
#endif
            n = n0 + nchunks * U;
            lean_done = true;
        }
    }
    if (lean_done) {
    } else if (nchunks && !PIPE) {
        for (size_t k = 0; k < nchunks; k++) {
            Chunk C;
            fetch(C);
            land(C);
            render(C);
        }
        n = n0 + nchunks * U;
    } else if (nchunks) {
        Chunk Ca, Cb;
        fetch(Ca);
        land(Ca);
        size_t k = 0;  // Ca has landed chunk k
        while (true) {
            const bool more_b = k + 1 < nchunks;
            if (more_b) fetch(Cb);
            render(Ca);
            if (!more_b) break;
            land(Cb);
            k++;
            const bool more_a = k + 1 < nchunks;
            if (more_a) fetch(Ca);
            render(Cb);
            if (!more_a) break;
            land(Ca);
            k++;
        }
        n = n0 + nchunks * U;
    }
    for (; n < n1; n++) {
        Req q;
        double val[2];
        smp_gen<MODE>(s, x0, 0.0, st, en, sr, q);
        val[0] = amp[q.idx[0]];
        val[1] = amp[q.idx[1]];
        *op = smp_eval<MODE>(q, val);
        op += V;
    }
    if (writer && part_wait(part_ctr, psync)) A.position[v] = s.pos;
}

inline dim3 grid_for(size_t V, int block) { return dim3((unsigned)((V + block - 1) / block)); }

// time parts for a block-constant *AtSpeed launch: the kernel keeps three wavefronts per SIMD resident (~160 VGPRs); two rounds
// of them measured best at 65 536 voices (69 us with 6 parts against 76 with 4 and 84 with 3), each part >= 32 samples
inline int speed_parts(size_t V, size_t N, size_t sample_len, size_t *part_len) {
    int split = tune_get("smp_split");
    if (split == 0) {
        const size_t waves = (V + 63) / 64;
        split = waves >= 6144 ? 1 : (int)(6144 / (waves ? waves : 1));
        if (split > 8) split = 8;
        // a sample far beyond the 256 MB Infinity Cache: the heads read HBM, and every extra part is one more place per voice that is being
        // read at the same time -- two parts measured best there (65 536 heads over an 8.6 GB sample: 1 / 2 / 3 / 4 / 6 / 8 parts = 190 /
        // 137 / 151 / 152 / 152 / 147 us, profiles/r06_sample_bank.md)
        if (sample_len * sizeof(double) > ((size_t)1 << 30) && split > 2) split = 2;
    }
    while (split > 1 && N / (size_t)split < 32) split--;
    size_t len = ((N + split - 1) / split + 7) / 8 * 8;
    while (split > 1 && (size_t)(split - 1) * len >= N) split--;  // every part renders at least one sample
    *part_len = len;
    return split;
}

template <int M>
void launch_sample(bool xmod, dim3 grid, dim3 block, hipStream_t st, const SmpArgs &A) {
    constexpr bool kHasSpeed = smp_base(M) >= 4 && M != 14;
    KernelTimer kt("sample_kernel", st);
    if (kHasSpeed && xmod) {
        if (A.px_store) hipLaunchKernelGGL((sample_kernel<M, kHasSpeed, true>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((sample_kernel<M, kHasSpeed, false>), grid, block, 0, st, A);
    } else {
        if (A.px_store) hipLaunchKernelGGL((sample_kernel<M, false, true>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((sample_kernel<M, false, false>), grid, block, 0, st, A);
    }
}

// maxiSample::playAtSpeedBetweenPointsFromPos (C:826-880) with the caller's `pos`: a pure function of its arguments (the
// member `position` is neither read nor written), so a block is one lane per (sample, voice)
__global__ __launch_bounds__(256) void sample_frompos_kernel(size_t V, size_t N, const double *__restrict__ amp, size_t len,
                                                             const double *__restrict__ freq, int fps,
                                                             const double *__restrict__ start, const double *__restrict__ end,
                                                             const double *__restrict__ pos, double *__restrict__ out, double sr) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * V) return;
    const size_t v = t % V;
    Smp s;
    s.amp = amp;
    s.len = len;
    s.pos = pos[t];
    s.step_div = 1.0;
    s.tprev = 0.0;
    s.tfirst = false;
    s.p0 = s.p1 = 0.0;
    SmpReq<8> q;
    smp_gen<8>(s, fps ? freq[t] : freq[v], 0.0, start[v], end[v], sr, q);
    double val[2];
    val[0] = amp[q.idx[0]];
    val[1] = amp[q.idx[1]];
    out[t] = smp_eval<8>(q, val);
}

}  // namespace
}  // namespace mxg

using namespace mxg;

namespace mxg {
namespace {
__global__ void widen_i32_kernel(int64_t *dst, const int32_t *src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (int64_t)src[i];
}
__global__ void narrow_i64_kernel(int32_t *dst, const int64_t *src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (int32_t)src[i];
}
}  // namespace
}  // namespace mxg

extern "C" {

// State plumbing for hosts that keep every integer member as int64 (the per-sample engine of include/maximilian.h): the flag
// arrays of the trigger-driven players (maxiTrigger::firstTrigger, phasorFirst) are int32 on this ABI.
int mxg_i64_from_i32(int64_t *d_dst, const int32_t *d_src, size_t n, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_dst && d_src, "null device pointer");
    if (n) hipLaunchKernelGGL(widen_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, resolve_stream(stream), d_dst, d_src, n);
    return check_hip(hipGetLastError(), "widen_i32_kernel launch");
}
int mxg_i32_from_i64(int32_t *d_dst, const int64_t *d_src, size_t n, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_dst && d_src, "null device pointer");
    if (n) hipLaunchKernelGGL(narrow_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, resolve_stream(stream), d_dst, d_src, n);
    return check_hip(hipGetLastError(), "narrow_i64_kernel launch");
}

int mxg_sample_render_frompos(size_t V, size_t N, const double *d_samples, size_t len, const double *d_freq, int fps,
                              const double *d_start, const double *d_end, const double *d_pos, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_samples && d_freq && d_start && d_end && d_pos && d_out, "null device pointer");
    MXG_REQUIRE(len > 0, "empty sample");
    if (V == 0 || N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const size_t n = V * N;
    KernelTimer kt("sample_frompos_kernel", st);
    hipLaunchKernelGGL(sample_frompos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, V, N, d_samples, len, d_freq,
                       fps, d_start, d_end, d_pos, d_out, (double)settings().sampleRate);
    return check_hip(hipGetLastError(), "sample_frompos_kernel launch");
}

int mxg_delay_render(int mode, size_t V, size_t N, const double *d_in, const int32_t *d_size,
                     const double *d_feedback, const int32_t *d_position, double *d_mem, size_t cap,
                     int32_t *d_phase, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (dl) or 1 (dlFromPosition)");
    MXG_REQUIRE(d_in && d_size && d_feedback && d_mem && d_phase && d_out, "null device pointer");
    MXG_REQUIRE(mode == 0 || d_position, "dlFromPosition needs d_position");
    MXG_REQUIRE(cap > 0 && cap <= 0x7fffffff, "cap must be in 1 .. 2^31-1");
    if (V == 0 || N == 0) return MXG_OK;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;  // delay_kernel is compiled for <= 256 lanes per workgroup
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("delay_kernel", st);
    // 16-byte pair-row input / output streams for dl (knob rw_store, rw_store_choice in mxg_common.h: automatic = non-temporal stores for
    // blocks from 64 MB)
    const bool pairs_ok = mode == 0 && !(((uintptr_t)d_in) & 15);
    const int rw = pairs_ok ? 1 + rw_store_choice(V, N, d_out, RW_WRITE_ONLY) : 1;  // (1 = 8-byte streams, 2 / 3 / 4 as the knob)
#define MXG_DL(M, X)                                                                                              \
    hipLaunchKernelGGL((delay_kernel<M, X>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_size, d_feedback, \
                       d_position, d_mem, (int)cap, d_phase, d_out)
    if (mode == 1) MXG_DL(1, 0);
    else if (!pairs_ok || rw < 2) MXG_DL(0, 0);
    else if (rw == 2) MXG_DL(0, 1);
    else if (rw == 3) MXG_DL(0, 2);
    else MXG_DL(0, 3);
#undef MXG_DL
    return check_hip(hipGetLastError(), "delay_kernel launch");
}

double *mxg_sample_upload(const double *h_samples, size_t len) {
    if (ensure_init_only()) return nullptr;
    if (!h_samples && len) {
        fail(MXG_ERR_INVALID, "mxg_sample_upload: null samples");
        return nullptr;
    }
    double *base = nullptr;
    const size_t total = len + kSmpGuardLo + kSmpGuardHi;  // layout: mxg_smp.h
    if (check_hip(hipMalloc(&base, total * sizeof(double)), "hipMalloc(sample)")) return nullptr;
    if (check_hip(hipMemset(base, 0, total * sizeof(double)), "hipMemset(sample)") ||
        (len && check_hip(hipMemcpy(base + kSmpGuardLo, h_samples, len * sizeof(double), hipMemcpyHostToDevice),
                          "hipMemcpy(sample)"))) {
        (void)hipFree(base);  // (the message of the failed call stays in mxg_last_error)
        return nullptr;
    }
    return base + kSmpGuardLo;
}

int mxg_sample_free(double *d_samples) {
    if (!d_samples) return MXG_OK;
    MXG_HIP(hipFree(d_samples - kSmpGuardLo));
    return MXG_OK;
}

int mxg_sample_render(int mode, size_t V, size_t N, const double *d_samples, size_t len,
                      int mySampleRate, const double *d_a, int aps, const double *d_start,
                      const double *d_end, double *d_position, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode >= 0 && mode <= 8, "unknown maxiSample mode");
    MXG_REQUIRE(d_samples && d_position && d_out, "null device pointer");
    MXG_REQUIRE(len > 0, "empty sample");
    MXG_REQUIRE(mySampleRate > 0, "mySampleRate must be > 0");
    MXG_REQUIRE(mode < 4 || d_a, "speed/frequency modes need d_a");
    MXG_REQUIRE(!(mode == 2 || mode == 7 || mode == 8) || (d_start && d_end), "mode needs d_start/d_end");
    MXG_REQUIRE(!(mode == 3 || mode == 6) || d_end, "mode needs d_end");
    if (V == 0 || N == 0) return MXG_OK;
    const size_t q = settings().sampleRate / (size_t)mySampleRate;  // integer division, C:1070
    int block = tune_get("voice_block");
    if (block > 256) block = 256;  // sample_kernel is compiled for <= 256 lanes per workgroup
    hipStream_t st = resolve_stream(stream);
    SmpArgs A = {V, N, d_samples, len, (double)q, (double)settings().sampleRate, d_a, nullptr,
                       d_start, d_end, d_position, nullptr, nullptr, d_out, rw_store_choice(V, N, d_out, RW_WRITE_ONLY), 0u, nullptr};
    const bool xmod = mode >= 4 && aps;
    const dim3 grid = grid_for(V, block);
    {
        // the paced schedule (mxg_pace.h; knob smp_pace: 0 automatic, 1 never, >= 2 a fixed period in ticks of 10 ns per 8 samples and
        // wavefront): play() -- whose whole-chunk path is a pure store stream -- at the store-bound bank sizes, on the controller
        const int knob = tune_get("smp_pace");
        if (knob >= 2) {
            A.pace_arg = (unsigned)knob;
        } else if (knob == 0 && mode == 0 && V >= 45056 && V <= 229375) {
            A.pace_arg = pace_start_period(V * 8 * 8);
            A.pace_ctl = A.pace_arg ? pace_words(SCR_SMP_PACE, st, kPaceWords) : nullptr;
            if (!A.pace_ctl) A.pace_arg = 0;  // (inside a graph capture before the first eager launch, or a device whose counter's rate is unknown: not paced)
        }
    }
    if (mode >= 4 && mode <= 6 && !xmod) {
        size_t part_len = N;
        const int split = speed_parts(V, N, len, &part_len);
        if (split > 1) {
            PartSync part_ctrs;
            const dim3 pgrid(grid.x, (unsigned)split);
            if (int e = part_sync_get(st, (size_t)grid.x * ((block + 63) / 64), split, &part_ctrs)) return e;
            const bool pipe = tune_get("smp_pipe") != 0;
            KernelTimer kt("sample_parts_kernel", st);
            // rows as rings (only the pieces behind the last one held are fetched): for samples beyond the Infinity Cache, where re-fetching
            // the overlap of consecutive windows doubled the HBM traffic (knob smp_ring: 0 automatic, 1 off, 2 on); measured, 65 536 heads:
            // 8.6 GB sample 142 -> 128 us, the shared 3.5 MB sample 72 -> 81 us (profiles/r06_sample_bank.md)
            const int ring_knob = tune_get("smp_ring");
            const bool ring = ring_knob == 2 || (ring_knob == 0 && len * sizeof(double) > ((size_t)1 << 30));
#define MXG_PARTS2(M, P, R)                                                                                                          \
    if (A.px_store) hipLaunchKernelGGL((sample_parts_kernel<M, P, true, R>), pgrid, dim3(block), 0, st, A, part_len, part_ctrs);      \
    else hipLaunchKernelGGL((sample_parts_kernel<M, P, false, R>), pgrid, dim3(block), 0, st, A, part_len, part_ctrs);
#define MXG_PARTS(M)                                                \
    if (pipe) {                                                     \
        if (ring) { MXG_PARTS2(M, true, true) } else { MXG_PARTS2(M, true, false) }   \
    } else {                                                        \
        if (ring) { MXG_PARTS2(M, false, true) } else { MXG_PARTS2(M, false, false) } \
    }
            if (mode == 4) { MXG_PARTS(4) }
            if (mode == 5) { MXG_PARTS(5) }
            if (mode == 6) { MXG_PARTS(6) }
#undef MXG_PARTS
#undef MXG_PARTS2
            return check_hip(hipGetLastError(), "sample_parts_kernel launch");
        }
    }
    switch (mode) {
        case 0: launch_sample<0>(xmod, grid, dim3(block), st, A); break;
        case 1: launch_sample<1>(xmod, grid, dim3(block), st, A); break;
        case 2: launch_sample<2>(xmod, grid, dim3(block), st, A); break;
        case 3: launch_sample<3>(xmod, grid, dim3(block), st, A); break;
        case 4: launch_sample<4>(xmod, grid, dim3(block), st, A); break;
        case 5: launch_sample<5>(xmod, grid, dim3(block), st, A); break;
        case 6: launch_sample<6>(xmod, grid, dim3(block), st, A); break;
        case 7: launch_sample<7>(xmod, grid, dim3(block), st, A); break;
        case 8: launch_sample<8>(xmod, grid, dim3(block), st, A); break;
    }
    return check_hip(hipGetLastError(), "sample_kernel launch");
}

int mxg_sample_render_trig(int mode, size_t V, size_t N, const double *d_samples, size_t len,
                           int mySampleRate, const double *d_trig, const double *d_a, int aps,
                           const double *d_p0, const double *d_p1, double *d_position,
                           double *d_tprev, int32_t *d_tfirst, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode >= 9 && mode <= 14, "unknown trigger-driven maxiSample mode");
    MXG_REQUIRE(d_samples && d_trig && d_tprev && d_tfirst && d_out, "null device pointer");
    MXG_REQUIRE(mode == 14 || d_position, "mode needs d_position");
    MXG_REQUIRE(len > 0, "empty sample");
    MXG_REQUIRE(mySampleRate > 0, "mySampleRate must be > 0");
    MXG_REQUIRE(!(mode >= 10 && mode <= 12) || d_a, "AtSpeed modes need d_a");
    MXG_REQUIRE(!(mode >= 11 && mode <= 13) || d_p0, "mode needs d_p0 (offset / pos)");
    MXG_REQUIRE(mode != 12 || d_p1, "playOnZXAtSpeedBetweenPoints needs d_p1 (length)");
    if (V == 0 || N == 0) return MXG_OK;
    const size_t q = settings().sampleRate / (size_t)mySampleRate;  // integer division, C:1070
    int block = tune_get("voice_block");
    if (block > 256) block = 256;
    hipStream_t st = resolve_stream(stream);
    const SmpArgs A = {V, N, d_samples, len, (double)q, (double)settings().sampleRate, d_a, d_trig,
                       d_p0, d_p1, d_position, d_tprev, d_tfirst, d_out, rw_store_choice(V, N, d_out, RW_WRITE_ONLY), 0u, nullptr};
    const bool xmod = aps != 0;
    const dim3 grid = grid_for(V, block);
    switch (mode) {
        case 9: launch_sample<9>(xmod, grid, dim3(block), st, A); break;
        case 10: launch_sample<10>(xmod, grid, dim3(block), st, A); break;
        case 11: launch_sample<11>(xmod, grid, dim3(block), st, A); break;
        case 12: launch_sample<12>(xmod, grid, dim3(block), st, A); break;
        case 13: launch_sample<13>(xmod, grid, dim3(block), st, A); break;
        case 14: launch_sample<14>(xmod, grid, dim3(block), st, A); break;
    }
    return check_hip(hipGetLastError(), "sample_kernel launch");
}

}  // extern "C"
