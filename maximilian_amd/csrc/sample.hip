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
// gather).  The buffer must be valid on [-1, len+1] with zero guards (see maxigpu.h).
#include "mxg_common.h"

namespace mxg {
namespace {

template <int MODE>
__global__ void __launch_bounds__(256) delay_kernel(size_t V, size_t N, const double *__restrict__ in,
                             const int32_t *__restrict__ size, const double *__restrict__ feedback,
                             const int32_t *__restrict__ position, double *__restrict__ mem,
                             int32_t *__restrict__ phase_io, double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int ph = phase_io[v];
    const int sz = size[v];
    const double fb = feedback[v];
    int pos = 0;
    if constexpr (MODE == 1) {
        pos = position[v];
        if (pos >= sz) pos = 0;  // C:433
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
            auto request = [&](size_t k, double(&xi)[U], double(&c)[U], int(&sl)[U]) {
                ph_prev = ph;
#pragma unroll
                for (int i = 0; i < U; i++) {
                    const size_t nn = k * U + i;
                    const size_t mm = (nn < N) ? nn : N - 1;  // clamped: no branch
                    xi[i] = ip[mm * V];
                    if (ph >= sz) ph = 0;  // C:421
                    sl[i] = ph;
                    c[i] = m[(size_t)ph * V];
                    ph += 1;
                }
            };
            auto retire = [&](double(&xi)[U], double(&c)[U], int(&sl)[U]) {
#pragma unroll
                for (int i = 0; i < U; i++) {
                    m[(size_t)sl[i] * V] = (c[i] * fb) + (xi[i] * fb) * 0.5;  // C:425
                    *op = c[i];                                                // C:424
                    op += V;
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
        if (ph >= sz) ph = 0;  // C:421 / C:432
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

struct Smp {
    const double *amp;
    size_t len;
    double pos;
    double step_div;  // (double)(sampleRate / mySampleRate), the INTEGER quotient of C:1070
    // trigger-driven modes (9-14): maxiTrigger::previousValue/firstTrigger (H:593-594), or
    // phasorPrev/phasorFirst (H:731-732) for playWithPhasor; p0/p1 = the per-voice offset/length/pos
    double tprev;
    bool tfirst;
    double p0, p1;
};

// Modes 9-13 are a maxiTrigger::onZX test (H:569-579) in front of one of the plain players.
__host__ __device__ constexpr int smp_base(int mode) {
    return mode == 9 ? 1 : (mode == 10 || mode == 11) ? 5 : mode == 12 ? 6 : mode == 13 ? 0 : mode;
}
__host__ __device__ constexpr int smp_loads(int mode) {
    return mode == 14 ? 2 : (smp_base(mode) <= 3 ? 1 : (smp_base(mode) == 7 ? 4 : 2));
}

// ---- pipelined form ------------------------------------------------------------------------
// Every player is split in two: smp_gen advances the play head and emits the gather indices (it
// never needs a loaded sample value), smp_eval turns the gathered values into the output.  The
// kernel issues the gathers of chunk k+1 before the stores of chunk k, so waiting for them is a
// counted vmcnt and the store stream is never drained (loads and stores retire in order on one
// counter).  Guarded reads of the reference (`cond ? A[i] : 0`) become a read of a clamped index
// plus a select, which loads the same value whenever the reference loads at all.
template <int MODE>
struct SmpReq {
    static constexpr int L = smp_loads(MODE);
    long long idx[L];
    double rem;
    bool ok;   // modes 1,3,4,5,6: the reference's bounds test; modes 7,8: "backward" branch
};

template <int MODE>
__device__ __forceinline__ void smp_gen(Smp &s, double x, double t, double start, double end,
                                        double sr, SmpReq<MODE> &q) {
    constexpr int B = smp_base(MODE);
    q.rem = 0.0;
    q.ok = true;
    if constexpr (MODE >= 9 && MODE <= 13) {  // C:1006-1042
        const bool zx = (s.tprev <= 0.0 || s.tfirst) && t > 0;  // H:572
        s.tprev = t;
        s.tfirst = false;
        if (zx) {
            if constexpr (MODE == 13) {  // setPosition(pos) C:749-751, maxiMap::clamp H:843-854
                double c = s.p0;
                if (c > 1.0) c = 1.0;
                else if (c < 0.0) c = 0.0;
                s.pos = c * (double)s.len;
            } else {
                s.pos = 0;  // trigger() C:597-600
                if constexpr (MODE == 11 || MODE == 12) s.pos = s.p0 * (double)s.len;  // C:1024, C:1032
            }
        }
    }
    if constexpr (MODE == 14) {  // playWithPhasor C:753-816 (pos1/pos2 are size_t there)
        const unsigned long long amplen = s.len;
        double pha = t;
        if (pha > 1) pha = 1;
        if (pha < 0) pha = 0;
        const double pos = pha * (double)amplen * 0.99999999999999;
        if (s.tfirst) {
            s.tfirst = false;
            s.tprev = pos;
        }
        unsigned long long pos1 = (unsigned long long)(round(s.tprev));
        unsigned long long pos2 = (unsigned long long)(round(pos));
        if (pos1 == pos2) {
            if (pos >= s.tprev) pos2++;
            else pos1--;  // 0 wraps to 2^64-1 and is caught by the next test, as in the reference
        }
        if (pos2 >= amplen) pos2 = 0;
        if (pos1 >= amplen) pos1 = 0;
        double q1;
        if (pos2 > pos1) {
            const double dist = (double)(pos2 - pos1);
            q1 = (dist == 0) ? 0 : (pos - (double)pos1) / dist;
        } else {
            const double dist = (double)((amplen - pos1) + pos2);
            if (dist == 0) q1 = 0;
            else if (pos > (double)pos1) q1 = (pos - (double)pos1) / dist;
            else q1 = ((double)(amplen - pos1) + pos) / dist;
        }
        q.rem = q1;
        q.idx[0] = (long long)pos1;
        q.idx[1] = (long long)pos2;
        s.tprev = pos;
    } else if constexpr (B == 0) {  // C:740-747
        q.idx[0] = (long long)s.pos;
        s.pos += 1.0;
        if ((size_t)(long long)s.pos >= s.len) s.pos = 0;
    } else if constexpr (B == 1) {  // C:982-991
        q.ok = (size_t)(long long)s.pos < s.len;
        q.idx[0] = q.ok ? (long long)s.pos : 0;
        s.pos += 1.0;
    } else if constexpr (B == 2) {  // C:960-967
        s.pos += 1.0;
        double lo = (double)s.len * start;
        if (s.pos < lo) s.pos = lo;
        if ((double)(long long)s.pos >= (double)s.len * end) s.pos = lo;
        q.idx[0] = (long long)s.pos;
    } else if constexpr (B == 3) {  // C:969-978
        s.pos += 1.0;
        if (end > 1.0) end = 1.0;
        q.ok = (double)(long long)s.pos < (double)s.len * end;
        q.idx[0] = q.ok ? (long long)s.pos : 0;
    } else if constexpr (B == 4 || B == 5 || B == 6) {  // C:1060-1075, C:994-1003, C:1047-1058
        long long i = (long long)s.pos;
        q.rem = s.pos - (double)i;
        if constexpr (B == 4) q.ok = (size_t)i < s.len;
        if constexpr (B == 5) q.ok = (size_t)(i + 1) < s.len;
        if constexpr (B == 6) {
            if (end > 1.0) end = 1.0;
            q.ok = (double)i < (double)s.len * end;
        }
        const long long first = (B == 5) ? i : 1 + i;
        q.idx[0] = q.ok ? first : 0;
        q.idx[1] = q.idx[0] + 1;
        s.pos = s.pos + ((x * kChandiv) / s.step_div);
        if constexpr (B == 4)
            if ((size_t)(long long)s.pos >= s.len) s.pos -= (double)s.len;
    } else if constexpr (B == 7) {  // C:884-956; idx = {a, b, c, d}
        double frequency = x;
        if (frequency > 0.) {
            if (s.pos < start) s.pos = start;
            if (s.pos >= end) s.pos = start;
            s.pos += ((end - start) / (sr / (frequency * kChandiv)));
            q.rem = s.pos - floor(s.pos);
            q.idx[0] = (s.pos > 0) ? (long long)((int)(floor(s.pos)) - 1) : 0;
            q.idx[1] = (long long)s.pos;
            q.idx[2] = (s.pos < end - 2) ? (long long)s.pos + 1 : 0;
            q.idx[3] = (s.pos < end - 3) ? (long long)s.pos + 2 : 0;
            q.ok = false;
        } else {
            frequency *= -1.;
            if (s.pos <= start) s.pos = end;
            s.pos -= ((end - start) / (sr / (frequency * kChandiv)));
            q.rem = s.pos - floor(s.pos);
            q.idx[0] = (s.pos > start && s.pos < end - 1) ? (long long)s.pos + 1 : 0;
            q.idx[1] = (long long)s.pos;
            q.idx[2] = (s.pos > start) ? (long long)s.pos - 1 : 0;
            q.idx[3] = (s.pos > start + 1) ? (long long)s.pos - 2 : 0;
            q.ok = true;
        }
    } else {  // C:823-880: `position` is a by-value parameter there, the head never advances
        double frequency = x, pos = s.pos;
        const size_t amplen = s.len;
        if (end >= (double)amplen) end = (double)(amplen - 1);
        if (frequency > 0.) {
            if (pos < start) pos = start;
            if (pos >= end) pos = start;
            pos += ((end - start) / ((sr) / (frequency * kChandiv)));
            q.rem = pos - floor(pos);
            long long posl = (long long)floor(pos);
            q.idx[0] = ((size_t)(posl + 1) < amplen) ? posl + 1 : posl - 1;
            q.idx[1] = ((size_t)(posl + 2) < amplen) ? posl + 2 : (long long)amplen - 1;
            q.ok = false;
        } else {
            frequency *= -1.;
            if (pos <= start) pos = end;
            pos -= ((end - start) / (sr / (frequency * kChandiv)));
            q.rem = pos - floor(pos);
            long long posl = (long long)floor(pos);
            q.idx[0] = (posl - 1 >= 0) ? posl - 1 : 0;
            q.idx[1] = (posl - 2 >= 0) ? posl - 2 : 0;
            q.ok = true;
        }
    }
}

template <int MODE>
__device__ __forceinline__ double smp_eval(const SmpReq<MODE> &q, const double *val) {
    constexpr int B = smp_base(MODE);
    if constexpr (MODE == 14) {
        const double q2 = 1 - q.rem;
        return (q.rem * val[0] + q2 * val[1]);  // C:810-811
    } else if constexpr (B == 0 || B == 2) {
        return val[0];
    } else if constexpr (B == 1 || B == 3) {
        return q.ok ? val[0] : 0.0;
    } else if constexpr (B == 4 || B == 5 || B == 6) {
        double o = ((1 - q.rem) * val[0] + q.rem * val[1]);
        return q.ok ? o : 0.0;
    } else if constexpr (B == 7) {
        const double a = val[0], b = val[1], c = val[2], d = val[3];
        double a1 = 0.5 * (c - a);
        double a2 = a - 2.5 * b + 2. * c - 0.5 * d;
        double a3 = 0.5 * (d - a) + 1.5 * (b - c);
        const double m = q.ok ? -q.rem : q.rem;  // C:950 multiplies by -remainder going backwards
        return (((a3 * q.rem + a2) * m + a1) * m + b);
    } else {
        const double w = q.ok ? (-1 - q.rem) : (1 - q.rem);  // C:872 / C:850
        return (w * val[0] + q.rem * val[1]);
    }
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
};

template <int MODE, bool XMOD>
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
#pragma unroll
            for (int i = 0; i < U; i++) {
                *op = smp_eval<MODE>(rcur[i], vcur[i]);
                op += V;
            }
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

inline dim3 grid_for(size_t V, int block) { return dim3((unsigned)((V + block - 1) / block)); }

template <int M>
void launch_sample(bool xmod, dim3 grid, dim3 block, hipStream_t st, const SmpArgs &A) {
    constexpr bool kHasSpeed = smp_base(M) >= 4 && M != 14;
    if (kHasSpeed && xmod)
        hipLaunchKernelGGL((sample_kernel<M, kHasSpeed>), grid, block, 0, st, A);
    else
        hipLaunchKernelGGL((sample_kernel<M, false>), grid, block, 0, st, A);
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

int mxg_delay_render(int mode, size_t V, size_t N, const double *d_in, const int32_t *d_size,
                     const double *d_feedback, const int32_t *d_position, double *d_mem, size_t cap,
                     int32_t *d_phase, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (dl) or 1 (dlFromPosition)");
    MXG_REQUIRE(d_in && d_size && d_feedback && d_mem && d_phase && d_out, "null device pointer");
    MXG_REQUIRE(mode == 0 || d_position, "dlFromPosition needs d_position");
    MXG_REQUIRE(cap > 0, "cap must be > 0");
    if (V == 0 || N == 0) return MXG_OK;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;  // delay_kernel is compiled for <= 256 lanes per workgroup
    hipStream_t st = resolve_stream(stream);
    if (mode == 0)
        hipLaunchKernelGGL((delay_kernel<0>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_size,
                           d_feedback, d_position, d_mem, d_phase, d_out);
    else
        hipLaunchKernelGGL((delay_kernel<1>), grid_for(V, block), dim3(block), 0, st, V, N, d_in, d_size,
                           d_feedback, d_position, d_mem, d_phase, d_out);
    return check_hip(hipGetLastError(), "delay_kernel launch");
}

double *mxg_sample_upload(const double *h_samples, size_t len) {
    if (ensure_init()) return nullptr;
    if (!h_samples && len) {
        fail(MXG_ERR_INVALID, "mxg_sample_upload: null samples");
        return nullptr;
    }
    double *base = nullptr;
    if (check_hip(hipMalloc(&base, (len + 3) * sizeof(double)), "hipMalloc(sample)")) return nullptr;
    if (check_hip(hipMemset(base, 0, (len + 3) * sizeof(double)), "hipMemset(sample)")) return nullptr;
    if (len && check_hip(hipMemcpy(base + 1, h_samples, len * sizeof(double), hipMemcpyHostToDevice),
                         "hipMemcpy(sample)"))
        return nullptr;
    return base + 1;
}

int mxg_sample_free(double *d_samples) {
    if (!d_samples) return MXG_OK;
    MXG_HIP(hipFree(d_samples - 1));
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
    const SmpArgs A = {V, N, d_samples, len, (double)q, (double)settings().sampleRate, d_a, nullptr,
                       d_start, d_end, d_position, nullptr, nullptr, d_out};
    const bool xmod = mode >= 4 && aps;
    const dim3 grid = grid_for(V, block);
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
                       d_p0, d_p1, d_position, d_tprev, d_tfirst, d_out};
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
