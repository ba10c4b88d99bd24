// grains.hip -- maxiTimeStretch / maxiStretch granular banks on gfx950.
//
// Path (reference L/ = src/libs/maxiGrains.h): window functors :18-90 and
// maxiGrainWindowCache::getWindow :112-120 (host, plan creation); maxiGrain ctor :160-181 and
// maxiGrain::play :216-245 (the non-MAXIGRAINFAST linear path); maxiGrainPlayer::play :270-283;
// maxiTimeStretch::play :341-355; maxiStretch::play :512-530.
//
// One stream = one maxiTimeStretch/maxiStretch object: a scheduler (position, looper,
// randomOffset) that spawns a grain roughly every grainLength*sr/overlaps samples, and a list of
// live grains whose outputs are summed IN CREATION ORDER.  Everything is + - * / floor on fp64
// plus integer indexing, so a stream is bit-exact as long as each grain's position recurrence
// (pos += inc, wrapped) and the per-sample sum order are kept -- both are, see the kernels.
// `rand() % 10` (L/maxiGrains.h:352) is a process-wide libc stream in the reference and cannot
// be reproduced for a bank: the caller supplies the drawn values per stream (d_rnd) or none (0).
//
// K8 `granular_kernel`: one lane per stream walks the T samples; up to 8 live grains sit in
// registers in creation order (FIFO: equal durations inside a plan).  The sample buffer is shared
// by all streams (35 MB in config 5: L2 / Infinity-Cache resident), the window table is staged in
// LDS.  Output is [T][S] sample-major like every other bank.
#include <math.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "mxg_common.h"
#include "mxg_sched.h"
#include "mxg_lanefold.h"

struct mxg_grain_plan {
    int window_kind, mySampleRate;
    double grainLength;
    unsigned long sampleDur;
    double *d_window;  // [sampleDur]
    std::vector<double> h_window;
};

namespace mxg {
namespace {

constexpr int kSlots = 8;

double window_value(int kind, unsigned long windowLength, unsigned long windowPos) {  // L/maxiGrains.h:18-97
    switch (kind) {
        case 0: return 0.5 * (1.0 - cos((2.0 * MXG_PI * windowPos) / (windowLength - 1)));
        case 1: return 0.54 - (0.46 * cos((2.0 * MXG_PI * windowPos) / (windowLength - 1)));
        case 2: return sin((MXG_PI * windowPos) / (windowLength - 1));
        case 3: return 1;
        case 4:
            return (2.0 / (windowLength - 1.0)) *
                   (((windowLength - 1.0) / 2.0) - fabs(windowPos - ((windowLength - 1.0) / 2.0)));
        case 5:
            return (2.0 / windowLength) * ((windowLength / 2.0) - fabs(windowPos - ((windowLength - 1.0) / 2.0)));
        case 6:
            return 0.35875 - (0.48829 * cos((2 * MXG_PI * windowPos) / (windowLength - 1))) +
                   (0.14128 * cos((4 * MXG_PI * windowPos) / (windowLength - 1))) +
                   (0.01168 * cos((6 * MXG_PI * windowPos) / (windowLength - 1)));
        case 7:
            return 0.3635819 - (0.4891775 * cos((2 * MXG_PI * windowPos) / (windowLength - 1))) +
                   (0.1365995 * cos((4 * MXG_PI * windowPos) / (windowLength - 1))) +
                   (0.0106411 * cos((6 * MXG_PI * windowPos) / (windowLength - 1)));
        default: {
            const double gausDivisor = (-2.0 * 0.3 * 0.3);
            const double phase = ((windowPos / (double)windowLength) - 0.5) * 2.0;
            return exp((phase * phase) / gausDivisor);
        }
    }
}

// A live grain handed in through d_gst must be one this plan could have made: the plan's duration (a grain keeps its own
// window in the reference; a bank has one window table per plan, so grains of another grain length have to finish -- or
// d_gst be cleared -- before the plan changes), a window index inside it, a position inside the sample and a finite
// step.  Anything else (NaN fails every comparison) is refused with MXG_ERR_INVALID instead of being used as an index.
__device__ __forceinline__ bool carried_grain_ok(double pos, double inc, double idx, double dur, double dlen,
                                                 int sampleDur) {
    return dur == (double)sampleDur && idx >= 0.0 && idx < dur && pos >= 0.0 && pos <= dlen && fabs(inc) <= dlen;
}

struct GrainArgs {
    size_t S, T, len, R;
    const double *amp, *window, *a, *b, *posMod;
    const int32_t *rnd;
    double *st, *gst, *out;
    int *err;
    double sr, cycleLength, grainLength;
    int sampleDur, mySampleRate, winInLds;
};

template <int MODE>
__global__ __launch_bounds__(64) void granular_kernel(GrainArgs A) {
    extern __shared__ double s_win[];
    const double *win = A.window;
    if (A.winInLds) {
        for (int i = threadIdx.x; i < A.sampleDur; i += blockDim.x) s_win[i] = A.window[i];
        __syncthreads();
        win = s_win;
    }
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.S) return;
    const size_t S = A.S;
    const double dlen = (double)A.len;
    SchedState q = {A.st[s], A.st[S + s], A.st[2 * S + s], 0.0, (size_t)A.st[3 * S + s]};
    const SchedConst sc = sched_const<MODE>(s, S, A.len, A.R, A.a, A.b, A.posMod, A.rnd, A.cycleLength,
                                            A.grainLength, A.sr, A.sampleDur);
    q.thr = A.cycleLength + q.randomOffset;
    double gpos[kSlots], ginc[kSlots];
    int gidx[kSlots], gdur[kSlots];
    int tail = 0;
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        gpos[k] = A.gst[(0 * kSlots + k) * S + s];
        ginc[k] = A.gst[(1 * kSlots + k) * S + s];
        gidx[k] = (int)A.gst[(2 * kSlots + k) * S + s];
        gdur[k] = (int)A.gst[(3 * kSlots + k) * S + s];
        if (gdur[k]) tail = k + 1;
    }
    int failed = 0;
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        const double dur = A.gst[(3 * kSlots + k) * S + s];
        if (dur != 0.0 && !carried_grain_ok(gpos[k], ginc[k], A.gst[(2 * kSlots + k) * S + s], dur, (double)A.len,
                                            A.sampleDur)) {
            failed = 4;
            gdur[k] = 0;
        }
    }
    double *op = A.out + s;
    for (size_t n = 0; n < A.T; n++) {
        // ---- scheduler (sched_step: the reference's play() up to the addGrain)
        double bornPos, bornInc;
        if (sched_step<MODE>(q, sc, n, bornPos, bornInc, failed)) {
            if (tail == kSlots) {  // compact the holes (only non-FIFO deaths leave any)
#pragma unroll
                for (int pass = 0; pass < kSlots - 1; pass++)
#pragma unroll
                    for (int k = 0; k < kSlots - 1; k++)
                        if (gdur[k] == 0) {
                            gpos[k] = gpos[k + 1]; ginc[k] = ginc[k + 1]; gidx[k] = gidx[k + 1];
                            gdur[k] = gdur[k + 1]; gdur[k + 1] = 0;
                        }
                tail = 0;
#pragma unroll
                for (int k = 0; k < kSlots; k++)
                    if (gdur[k]) tail = k + 1;
            }
            if (tail == kSlots) {
                failed = 1;  // more than 8 grains alive: reported through *err, stream keeps running
            } else {
#pragma unroll
                for (int k = 0; k < kSlots; k++)
                    if (k == tail) {
                        gpos[k] = bornPos; ginc[k] = bornInc; gidx[k] = 0; gdur[k] = A.sampleDur;
                    }
                tail++;
            }
        }
        // ---- maxiGrainPlayer::play :270-283: sum the live grains in creation order
        double total = 0.0;
#pragma unroll
        for (int k = 0; k < kSlots; k++) {
            if (gdur[k] > 0) {  // maxiGrain::play :216-245
                const double envValue = win[gidx[k]];
                double pos = gpos[k] + ginc[k];
                if (pos >= dlen)
                    pos -= dlen;
                else if (pos < 0)
                    pos += dlen;
                gpos[k] = pos;
                const double fl = floor(pos);
                const double remainder = pos - fl;
                const long long ia = (long long)fl;
                long long ib = ia + 1;
                if ((size_t)ib >= A.len) ib = 0;
                double o = ((1 - remainder) * A.amp[ia] + remainder * A.amp[ib]);
                o *= envValue;
                total += o;
                gidx[k]++;
                if (gidx[k] == gdur[k]) gdur[k] = 0;  // finished: erased from the list
            }
        }
        if (tail > 0 && gdur[0] == 0) {  // FIFO: the oldest grain is gone, close ranks
#pragma unroll
            for (int k = 0; k < kSlots - 1; k++) {
                gpos[k] = gpos[k + 1]; ginc[k] = ginc[k + 1]; gidx[k] = gidx[k + 1]; gdur[k] = gdur[k + 1];
            }
            gdur[kSlots - 1] = 0;
            tail--;
        }
        *op = total;
        op += S;
    }
    if (failed) atomicMax(A.err, failed);
    // state out, compacted (live grains first, creation order)
    A.st[s] = q.position;
    A.st[S + s] = q.looper;
    A.st[2 * S + s] = q.randomOffset;
    A.st[3 * S + s] = (double)q.cursor;
#pragma unroll
    for (int pass = 0; pass < kSlots - 1; pass++)
#pragma unroll
        for (int k = 0; k < kSlots - 1; k++)
            if (gdur[k] == 0) {
                gpos[k] = gpos[k + 1]; ginc[k] = ginc[k + 1]; gidx[k] = gidx[k + 1];
                gdur[k] = gdur[k + 1]; gdur[k + 1] = 0;
            }
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        const bool live = gdur[k] > 0;
        A.gst[(0 * kSlots + k) * S + s] = live ? gpos[k] : 0.0;
        A.gst[(1 * kSlots + k) * S + s] = live ? ginc[k] : 0.0;
        A.gst[(2 * kSlots + k) * S + s] = live ? (double)gidx[k] : 0.0;
        A.gst[(3 * kSlots + k) * S + s] = live ? (double)gdur[k] : 0.0;
    }
}

// ---- K8a/K8b: time-sharded version ------------------------------------------------------------
// The only truly serial part of a stream is its scheduler (position += rate with wraps, looper,
// spawn decisions): ~10 flops per sample.  K8a runs just that, one lane per stream, and records
// every spawn (sample index, grain start position) plus, per time chunk, how many spawns precede
// it.  A grain's own recurrence (pos += inc, wrapped) restarts at its birth, so K8b can render any
// chunk [n0, n1) of any stream independently: one lane per (stream, chunk) first re-creates the
// grains alive at n0 -- carried-in grains and earlier spawns, in creation order -- by replaying
// their position recurrence for the n0 - birth elapsed samples (<= one grain length; the same
// additions in the same order, so bit-identical), then runs the chunk exactly like K8.
// Parallelism S x chunks instead of S; redundant warm-up work < 10 % for chunks >= half a grain.
struct SchedArgs {
    size_t S, T, len, R, G, Tc, C;
    const double *a, *b, *posMod;
    const int32_t *rnd;
    double *st;
    int32_t *spawn_n;    // [G][S]
    double *spawn_pos;   // [G][S]
    double *spawn_inc;   // [G][S]
    int32_t *chunk_first;  // [C+1][S]: spawns before sample c*Tc
    int *err;
    double sr, cycleLength, grainLength;
    int sampleDur, fast;
    // Time slices (mxg_granular_render pipelines the scheduler against the render): this launch covers samples
    // [n_base, n_base + T) of the call; carry = [2][S] spawn count | next chunk-table row, kept between slices
    // (null: a single launch).  c_end = last chunk-table row this launch completes.
    int n_base;
    int32_t *carry;
    size_t c_end;
};

// Loads and stores of the scheduler's lists.  COH = the streamed form (the renders of the same launch read what the scheduler lanes
// write while both run, possibly on another XCD with its own L2): relaxed atomics at agent scope, i.e. write-through stores and
// loads that are not served from a stale line; ordering comes from the progress word (sched_walk / granular_unit_kernel).
template <bool COH, typename T>
__device__ __forceinline__ T ld_list(const T *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, typename T>
__device__ __forceinline__ void st_list(T *p, T v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// K8a for stream s over A.T samples.  STREAMED: one launch schedules the whole call while the tile renders of the SAME launch
// consume the lists; prog[s] = number of chunk-table rows of this stream that are complete (rows 0 .. prog - 1, and with them
// every spawn they count).  A row is published one event late -- at the next spawn, after an `s_waitcnt vmcnt(0)` that by then
// only finds stores issued a spawn ago -- so the walk never waits for its own stores; the sample-by-sample walk publishes at
// every tile boundary it passes.
template <int MODE, bool STREAMED>
__device__ __forceinline__ void sched_walk(const SchedArgs &A, const size_t s, int *prog) {
    const size_t S = A.S;
    SchedState q = {A.st[s], A.st[S + s], A.st[2 * S + s], 0.0, (size_t)A.st[3 * S + s]};
    const SchedConst sc = sched_const<MODE>(s, S, A.len, A.R, A.a, A.b, A.posMod, A.rnd, A.cycleLength,
                                            A.grainLength, A.sr, A.sampleDur);
    int count = (A.carry && A.n_base) ? A.carry[s] : 0;
    int failed = 0;
    // The serial part of a stream: keep this loop to the recurrences themselves (the chunk table is
    // derived from the spawn list afterwards).  thr = cycleLength + randomOffset changes only at a spawn.
    q.thr = A.cycleLength + q.randomOffset;
    const int Tn = (int)A.T;
    // chunk_first[c] = number of spawns before sample c*Tc.  Spawns arrive in increasing n, so the table is
    // filled as they are recorded -- stores only (a pass over the finished list would chain one dependent
    // global load per chunk: 1100 chunks x ~0.5 us was most of this kernel's time).
    size_t cnext = (A.carry && A.n_base) ? (size_t)A.carry[S + s] : 0;
    auto publish = [&]() {
        if constexpr (STREAMED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(prog + s, (int)cnext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto rows_to = [&](size_t clast, int stored) {  // rows cnext .. clast say `stored`
        for (; cnext <= clast && cnext <= A.C; cnext++) st_list<STREAMED>(A.chunk_first + cnext * S + s, (int32_t)stored);
    };
    auto record = [&](int n, double pos0, double inc) {
        n += A.n_base;  // sample index within the call
        const int stored = (size_t)count < A.G ? count : (int)A.G;
        if constexpr (STREAMED) {  // what the previous event stored is complete by now: publish it before storing more
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(prog + s, (int)cnext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        rows_to((size_t)n / A.Tc, stored);  // chunks starting at or before n: this spawn's predecessors only
        if ((size_t)count < A.G) {
            st_list<STREAMED>(A.spawn_n + (size_t)count * S + s, (int32_t)n);
            st_list<STREAMED>(A.spawn_pos + (size_t)count * S + s, pos0);
            st_list<STREAMED>(A.spawn_inc + (size_t)count * S + s, inc);
        } else {
            failed = 3;
        }
        count++;
    };
    const int nstart = sched_run_events<MODE>(q, sc, Tn, A.fast != 0, failed, record);
    for (int n = nstart; n < Tn; n++) {
        double pos0, inc;
        if (sched_step<MODE>(q, sc, (size_t)n, pos0, inc, failed)) record(n, pos0, inc);
        if constexpr (STREAMED) {
            // the slow walk passes a tile boundary (STREAMED: Tc = 64, n_base = 0): rows up to the tile that starts at n + 1 are final
            if (((n + 1) & 63) == 0) {
                rows_to((size_t)(n + 1) >> 6, (size_t)count < A.G ? count : (int)A.G);
                publish();
            }
        }
    }
    {   // chunks after the last spawn (of this slice: up to the row the next slice starts in)
        const int stored = (size_t)count < A.G ? count : (int)A.G;
        for (; cnext <= A.c_end; cnext++) st_list<STREAMED>(A.chunk_first + cnext * S + s, (int32_t)stored);
    }
    publish();
    if (A.carry) {
        A.carry[s] = count;
        A.carry[S + s] = (int32_t)cnext;
    }
    if (failed) atomicMax(A.err, failed);
    A.st[s] = q.position;
    A.st[S + s] = q.looper;
    A.st[2 * S + s] = q.randomOffset;
    A.st[3 * S + s] = (double)q.cursor;
}

template <int MODE>
__global__ __launch_bounds__(64) void granular_sched_kernel(SchedArgs A) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.S) return;
    sched_walk<MODE, false>(A, s, nullptr);
}

struct RenderArgs {
    size_t S, T, len, G, Tc, C;
    const double *amp, *window;
    const int32_t *spawn_n;
    const double *spawn_pos, *spawn_inc;
    const int32_t *chunk_first;
    const double *gst_in;  // carried-in grains (state before the launch)
    double *gst_out;       // grains alive after sample T-1 (written by the last chunk's lanes)
    double *out;
    int *err;
    int sampleDur, winInLds;
};

// one maxiGrain::play step (L/maxiGrains.h:216-245) on (pos, idx); returns the windowed sample
__device__ __forceinline__ double grain_step(double &pos, int &idx, const double inc, const double dlen,
                                             const size_t len, const double *amp, const double *win) {
    const double envValue = win[idx];
    double p = pos + inc;
    if (p >= dlen)
        p -= dlen;
    else if (p < 0)
        p += dlen;
    pos = p;
    const double fl = floor(p);
    const double remainder = p - fl;
    const long long ia = (long long)fl;
    long long ib = ia + 1;
    if ((size_t)ib >= len) ib = 0;
    double o = ((1 - remainder) * amp[ia] + remainder * amp[ib]);
    o *= envValue;
    idx++;
    return o;
}

// the position recurrence alone, `steps` times (warm-up to a chunk boundary)
__device__ __forceinline__ double grain_advance(double pos, const double inc, const double dlen, int steps) {
    for (int i = 0; i < steps; i++) {
        double p = pos + inc;
        if (p >= dlen)
            p -= dlen;
        else if (p < 0)
            p += dlen;
        pos = p;
    }
    return pos;
}

__global__ __launch_bounds__(64) void granular_render_kernel(RenderArgs A) {
    extern __shared__ double s_win[];
    const double *win = A.window;
    if (A.winInLds) {
        for (int i = threadIdx.x; i < A.sampleDur; i += blockDim.x) s_win[i] = A.window[i];
        __syncthreads();
        win = s_win;
    }
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t S = A.S;
    if (gid >= S * A.C) return;
    const size_t s = gid % S, c = gid / S;  // consecutive lanes = consecutive streams of one chunk
    const double dlen = (double)A.len;
    const size_t n0 = c * A.Tc, n1 = (n0 + A.Tc < A.T) ? n0 + A.Tc : A.T;
    double gpos[kSlots], ginc[kSlots];
    int gidx[kSlots], gdur[kSlots];
    int tail = 0, failed = 0;
#pragma unroll
    for (int k = 0; k < kSlots; k++) { gpos[k] = 0; ginc[k] = 0; gidx[k] = 0; gdur[k] = 0; }
    auto push = [&](double pos, double inc, int idx, int dur) {
        if (tail == kSlots) { failed = 1; return; }
#pragma unroll
        for (int k = 0; k < kSlots; k++)
            if (k == tail) { gpos[k] = pos; ginc[k] = inc; gidx[k] = idx; gdur[k] = dur; }
        tail++;
    };
    // (1) carried-in grains still alive at n0, oldest first
    for (int k = 0; k < kSlots; k++) {
        const double ddur = A.gst_in[(3 * kSlots + k) * S + s];
        if (ddur == 0.0) continue;
        const double inc = A.gst_in[(1 * kSlots + k) * S + s];
        if (!carried_grain_ok(A.gst_in[(0 * kSlots + k) * S + s], inc, A.gst_in[(2 * kSlots + k) * S + s], ddur, dlen,
                              A.sampleDur)) {
            failed = 4;
            continue;
        }
        const int dur = (int)ddur;
        const int idx = (int)A.gst_in[(2 * kSlots + k) * S + s];
        if ((long long)idx + (long long)n0 >= dur) continue;  // finished before this chunk
        const double pos = grain_advance(A.gst_in[(0 * kSlots + k) * S + s], inc, dlen, (int)n0);
        push(pos, inc, idx + (int)n0, dur);
    }
    // (2) grains spawned in earlier chunks of this launch and still alive at n0, in spawn order
    const int first = A.chunk_first[c * S + s];
    int j0 = first;
    while (j0 > 0 && (long long)A.spawn_n[(size_t)(j0 - 1) * S + s] + A.sampleDur > (long long)n0) j0--;
    for (int j = j0; j < first; j++) {
        const int born = A.spawn_n[(size_t)j * S + s];
        const int steps = (int)n0 - born;
        const double inc = A.spawn_inc[(size_t)j * S + s];
        push(grain_advance(A.spawn_pos[(size_t)j * S + s], inc, dlen, steps), inc, steps, A.sampleDur);
    }
    // (3) the chunk itself
    int jn = first;
    const int jend = A.chunk_first[(c + 1) * S + s];
    int nextSpawn = jn < jend ? A.spawn_n[(size_t)jn * S + s] : -1;
    double *op = A.out + n0 * S + s;
    for (size_t n = n0; n < n1; n++) {
        if ((int)n == nextSpawn) {
            if (tail == kSlots) {  // compact holes (non-FIFO deaths only)
#pragma unroll
                for (int pass = 0; pass < kSlots - 1; pass++)
#pragma unroll
                    for (int k = 0; k < kSlots - 1; k++)
                        if (gdur[k] == 0) {
                            gpos[k] = gpos[k + 1]; ginc[k] = ginc[k + 1]; gidx[k] = gidx[k + 1];
                            gdur[k] = gdur[k + 1]; gdur[k + 1] = 0;
                        }
                tail = 0;
#pragma unroll
                for (int k = 0; k < kSlots; k++)
                    if (gdur[k]) tail = k + 1;
            }
            push(A.spawn_pos[(size_t)jn * S + s], A.spawn_inc[(size_t)jn * S + s], 0, A.sampleDur);
            jn++;
            nextSpawn = jn < jend ? A.spawn_n[(size_t)jn * S + s] : -1;
        }
        // Branch-free over the slots so that the gathers of all live grains are in flight together
        // (one L2 round trip per sample instead of one per grain); dead slots read element 0 and are
        // not added.  Slots >= 5 are only walked when some lane of the wave has that many grains.
        double total = 0.0;
        double va[kSlots], vb[kSlots], ve[kSlots], vr[kSlots];
        const bool deep = __any(tail > 5);
#pragma unroll
        for (int k = 0; k < kSlots; k++) {
            if (k < 5 || deep) {
                const bool alive = gdur[k] > 0;
                double p = gpos[k] + ginc[k];  // maxiGrain::play :222-226
                if (p >= dlen)
                    p -= dlen;
                else if (p < 0)
                    p += dlen;
                gpos[k] = alive ? p : gpos[k];
                const double fl = floor(p);
                vr[k] = p - fl;
                long long ia = alive ? (long long)fl : 0;
                // buffer[a] and buffer[a+1] in ONE 16-byte request (8-byte aligned; the uploaded buffer is readable at
                // [len], mxg_sample_upload's guard): half the gather requests of two 8-byte loads.  The wrap b = 0
                // (:231-233) is patched in by a second load that only lanes sitting on the last sample issue.
                const double2v pr = *reinterpret_cast<const double2v *>(A.amp + ia);
                va[k] = pr.x;
                vb[k] = pr.y;
                if ((size_t)(ia + 1) >= A.len) vb[k] = A.amp[0];
                ve[k] = win[alive ? gidx[k] : 0];
            }
        }
#pragma unroll
        for (int k = 0; k < kSlots; k++) {
            if (k < 5 || deep) {
                if (gdur[k] > 0) {
                    double o = ((1 - vr[k]) * va[k] + vr[k] * vb[k]);  // :236-238
                    o *= ve[k];
                    total += o;
                    gidx[k]++;
                    if (gidx[k] == gdur[k]) gdur[k] = 0;
                }
            }
        }
        if (tail > 0 && gdur[0] == 0) {
#pragma unroll
            for (int k = 0; k < kSlots - 1; k++) {
                gpos[k] = gpos[k + 1]; ginc[k] = ginc[k + 1]; gidx[k] = gidx[k + 1]; gdur[k] = gdur[k + 1];
            }
            gdur[kSlots - 1] = 0;
            tail--;
        }
        *op = total;
        op += S;
    }
    if (failed) atomicMax(A.err, failed);
    if (c == A.C - 1) {  // grains alive after the last sample: the state handed to the next launch
#pragma unroll
        for (int pass = 0; pass < kSlots - 1; pass++)
#pragma unroll
            for (int k = 0; k < kSlots - 1; k++)
                if (gdur[k] == 0) {
                    gpos[k] = gpos[k + 1]; ginc[k] = ginc[k + 1]; gidx[k] = gidx[k + 1];
                    gdur[k] = gdur[k + 1]; gdur[k + 1] = 0;
                }
#pragma unroll
        for (int k = 0; k < kSlots; k++) {
            const bool live = gdur[k] > 0;
            A.gst_out[(0 * kSlots + k) * S + s] = live ? gpos[k] : 0.0;
            A.gst_out[(1 * kSlots + k) * S + s] = live ? ginc[k] : 0.0;
            A.gst_out[(2 * kSlots + k) * S + s] = live ? (double)gidx[k] : 0.0;
            A.gst_out[(3 * kSlots + k) * S + s] = live ? (double)gdur[k] : 0.0;
        }
    }
}

// ---- K8c: coalesced render for unit-increment grains -------------------------------------------------
// maxiTimeStretch spawns every grain with speed +-1 (L/maxiGrains.h:350); when
// inc = sampleDur/(sr/(1/grainLength)) is exactly 1.0 (e.g. grainLength 0.05 at 44.1 kHz) a grain
// reads integer positions pos0 +- 1, +- 2, ...: `pos += inc` is exact, the wrap is a modulo and the
// interpolation remainder is exactly 0, so sample k of a grain is a closed form of k -- no
// recurrence left.  Then the lanes of a wavefront can be 64 CONSECUTIVE SAMPLES of one stream:
// sample-buffer and window reads are contiguous (coalesced) instead of 64 scattered gathers, and
// each output is still the creation-ordered sum of the same products ((1-0)*buf[a] + 0*buf[b])*win[k].
// A 256-thread workgroup renders a 64-stream x 64-sample tile (each wave 16 streams, one after the
// other) into LDS and writes it out transposed, so the [T][S] stores are coalesced too.
struct UnitArgs {
    size_t S, T, len, G, C;  // C = number of 64-sample chunks; chunk_first is [C+1][S]
    const double *amp, *window;
    const int32_t *spawn_n;
    const double *spawn_pos, *spawn_inc;  // spawn_inc is +1.0 or -1.0 here: the grain's direction
    const int32_t *chunk_first;
    const double *gst_in;
    double *gst_out, *out;
    int *err;
    int sampleDur;
    unsigned c0;  // first tile of this launch (time slices)
    unsigned ltiles, cend;  // K8d: consecutive tiles per workgroup, one past the last tile of this launch
    // fused maxiMix::stereo mixdown of the tile (NULL = off): pan [S], mixpart [stream tiles][T][2]
    const double *pan;
    double *mixpart;
    // which tile renderer a call takes is decided ON THE DEVICE when it depends on device state (the carried-in grains of a
    // maxiTimeStretch on the integer grid: granular_unit_check_kernel writes *sel = 0 if K8c applies): both renderers are
    // enqueued and the one whose `want` differs from *sel returns at once -- no read-back, no host synchronisation
    const int *sel;
    int want;
    // streamed form: the stream's retry word (set by a tile render that gave up polling) and the polling budget
    int *retry;
    int spin_limit;
};
__device__ __forceinline__ bool unit_args_skip(const UnitArgs &A) { return A.sel && (*A.sel != 0) != (A.want != 0); }

__device__ __forceinline__ long long unit_index(long long t, long long len) {  // t mod len, result in [0,len)
    while (t >= len) t -= len;
    while (t < 0) t += len;
    return t;
}

// sample of a grain that has made `steps` position steps from pos0 and reads window index widx
__device__ __forceinline__ double unit_sample(const double *amp, const double *win, long long len, long long pos0,
                                              long long sgn, long long steps, long long widx) {
    const long long ia = unit_index(pos0 + steps * sgn, len);  // maxiGrain::play :222-228
    long long ib = ia + 1;
    if (ib >= len) ib = 0;  // :231-233
    const double remainder = 0.0;
    double o = ((1 - remainder) * amp[ia] + remainder * amp[ib]);  // :236-237, literally
    o *= win[widx];
    return o;
}

#ifndef MXG_UNIT_GROUP
#define MXG_UNIT_GROUP 1
#endif
constexpr int kUnitGroup = MXG_UNIT_GROUP;  // streams rendered together by a wavefront (divides 16)
#ifndef MXG_UNIT_BATCH
#define MXG_UNIT_BATCH 8
#endif
#ifndef MXG_UNIT_DPP
#define MXG_UNIT_DPP 0  // A/B: 1 = an interior pair's buffer[a] as an 8-byte load, buffer[a + 1] from the neighbour lane (wave_shl / wave_shr)
                        // + one single-lane load: same bits, half the address cycles -- and 829 -> 1008 us per config-5 step (round 5):
                        // the kernel is bound by issue and latency, not by the texture addresser; 0 = one 16-byte request per lane
#endif
constexpr int kUnitBatch = MXG_UNIT_BATCH;  // interior (grain, tile) pairs in flight per wavefront
constexpr int kInteriorFlag = 1 << 29;      // in s_cnt: every candidate of the stream is interior to the tile
constexpr int kFlatFlag = 1 << 30;          // in s_cnt: the stream-tile goes through the flattened phase 2a (1..kSlots candidates)
constexpr int kCand = 12;  // candidates per stream and tile: <= 8 alive at the tile's start + spawns inside the tile
// Candidate metadata is packed into two ints (sampleDur < sr/2 <= 2^15 by the window-cache rule): with the 33 KB
// transpose tile this keeps a workgroup under 40 KB of LDS, i.e. four workgroups per CU instead of three.

// The finished [64 samples][64 streams] tile (LDS, row stride 65): coalesced [T][S] stores, 16 sample rows per wave, and
// -- when a pan is given -- the maxiMix::stereo partial sums of the tile's 64 streams (shared by K8c and K8d).
__device__ __forceinline__ void tile_epilogue(const UnitArgs &A, const double *s_tile, const size_t s0, const size_t n0,
                                              const int lane, const int wave, const unsigned stile) {
    const size_t S = A.S;
    for (int r = wave * 16; r < wave * 16 + 16; r++) {
        const size_t nn = n0 + r, s = s0 + lane;
        if (nn < A.T && s < S) A.out[nn * S + s] = s_tile[r * 65 + lane];
    }
    if (A.pan) {  // maxiMix::stereo (C:503-509) of the tile's 64 streams, 16 sample rows per wave: the butterfly of K1m
        const size_t s = s0 + lane;
        double x = s < S ? A.pan[s] : 0.0;
        if (x > 1) x = 1;
        if (x < 0) x = 0;
        const double gl = sqrt(1.0 - x), gr = sqrt(x);
        double L[kMixChunk], R[kMixChunk];
        int idx[kMixChunk];
#pragma unroll
        for (int i = 0; i < kMixChunk; i++) {
            const int r = wave * 16 + i;
            const double v = (s < S && n0 + r < A.T) ? s_tile[r * 65 + lane] : 0.0;
            L[i] = v * gl;
            R[i] = v * gr;
            idx[i] = i;
        }
        const int slot = fold_chunk_swap<int>(idx);
        const double sl = quad_sum(fold_chunk_swap<double>(L)), sr = quad_sum(fold_chunk_swap<double>(R));
        const size_t nn = n0 + wave * 16 + (size_t)(slot < 0 ? 0 : slot);
        if ((lane & 3) == 0 && slot >= 0 && nn < A.T) {
            double *dst = A.mixpart + ((size_t)stile * A.T + nn) * 2;
            dst[0] = sl;
            dst[1] = sr;
        }
    }
}

// grains alive after sample T-1, creation order, closed form (one lane per stream)
template <bool COH>
__device__ __forceinline__ void unit_state_lane(const UnitArgs &A, const size_t s) {
    const size_t S = A.S;
    const long long len = (long long)A.len, T = (long long)A.T;
    double gp[kSlots], gi[kSlots], gx[kSlots], gd[kSlots];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kSlots; k++) { gp[k] = gi[k] = gx[k] = gd[k] = 0.0; }
    auto push = [&](double p, double i, double x, double d) {
#pragma unroll
        for (int k = 0; k < kSlots; k++)
            if (k == cnt) { gp[k] = p; gi[k] = i; gx[k] = x; gd[k] = d; }
        cnt++;
    };
    for (int k = 0; k < kSlots; k++) {
        const long long dur = (long long)A.gst_in[(3 * kSlots + k) * S + s];
        if (!dur) continue;
        const long long idx0 = (long long)A.gst_in[(2 * kSlots + k) * S + s];
        if (idx0 + T >= dur) continue;
        const long long pos = (long long)A.gst_in[(0 * kSlots + k) * S + s];
        const long long inc = (long long)A.gst_in[(1 * kSlots + k) * S + s];
        if (cnt < kSlots) push((double)unit_index(pos + T * inc, len), (double)inc, (double)(idx0 + T), (double)dur);
    }
    const int count = ld_list<COH>(A.chunk_first + A.C * S + s);
    int j0 = count;
    while (j0 > 0 && (long long)ld_list<COH>(A.spawn_n + (size_t)(j0 - 1) * S + s) + A.sampleDur > T) j0--;
    for (int j = j0; j < count; j++) {
        const long long born = ld_list<COH>(A.spawn_n + (size_t)j * S + s);
        const long long pos0 = (long long)ld_list<COH>(A.spawn_pos + (size_t)j * S + s);
        const long long steps = T - born;
        const long long sgn = ld_list<COH>(A.spawn_inc + (size_t)j * S + s) > 0 ? 1 : -1;
        if (cnt < kSlots) push((double)unit_index(pos0 + steps * sgn, len), (double)sgn, (double)steps, (double)A.sampleDur);
    }
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        A.gst_out[(0 * kSlots + k) * S + s] = gp[k];
        A.gst_out[(1 * kSlots + k) * S + s] = gi[k];
        A.gst_out[(2 * kSlots + k) * S + s] = gx[k];
        A.gst_out[(3 * kSlots + k) * S + s] = gd[k];
    }
}
__global__ __launch_bounds__(64) void granular_unit_state_kernel(UnitArgs A) {
    if (unit_args_skip(A)) return;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.S) return;
    unit_state_lane<false>(A, s);
}

// SMODE < 0: the tile renderer alone (grid = stream tiles x tiles of the slice; the lists are complete when it starts).
// SMODE = 0 / 2 (maxiTimeStretch::play / playAtPosition), the STREAMED form: one launch for the whole call, 1-D grid.  The first
// `nsched` workgroups are the scheduler (one lane per stream, 256 streams a workgroup: sched_walk<SMODE, true>, then the grains alive
// after the call); every other workgroup renders one tile, tiles in dispatch order, and starts by waiting until the 64 streams of
// its tile have published the two chunk-table rows it reads (Q.prog).  Workgroups are dispatched in index order, so the scheduler
// wavefronts are resident before the first renderer polls; the scheduler needs about half the time the renders need (0.41 us
// against 0.79 us per tile row on config 5), so after the first few rows nobody waits.  A renderer that sees no progress for
// its polling budget gives up (its tile stays silent for the moment) and raises the retry word: see below.
// (Before: four time slices on two streams -- the cross-queue event hops, the first slice's scheduler and the launch gaps between
// the slices were ~10 % of the config-5 step, profiles/r05_config5_timeline.md.)
// Round 6: a time-out no longer ends in silence.  The tile that gave up raises the stream's RETRY word (A.retry) instead of an error; when
// the launch has ended the lists are complete whatever happened (the scheduler workgroups got their turn at the latest once the
// renderers had given up), and the retry kernel that follows every streamed launch (granular_retry_kernel: a small persistent grid that
// returns at once while the word is zero) renders the call's tiles again with the renderer-alone form -- the sliced form's kernels, the
// same bits.  Knob "grain_spin_limit" lowers the polling budget (tests force the path with it); mxg_granular_retries() counts them.
constexpr int kStreamSpin = 1 << 22;
template <int SMODE>
__device__ __forceinline__ void granular_unit_body(const UnitArgs &A, const SchedArgs &Q, int *prog, const unsigned nsched, const unsigned bxi,
                                                   const unsigned byi) {
    constexpr bool COH = SMODE >= 0;
    unsigned bx = bxi, by = byi;
    if constexpr (COH) {
        if (bxi < nsched) {
            const size_t s = (size_t)bxi * 256 + threadIdx.x;
            if (s >= A.S) return;
            if (unit_args_skip(A)) return;  // (both renderers are enqueued and the other one was picked: it schedules, too)
            __builtin_amdgcn_s_setprio(3);
            sched_walk<(SMODE >= 0 ? SMODE : 0), true>(Q, s, prog);
            unit_state_lane<true>(A, s);
            return;
        }
        const unsigned idx = bxi - nsched, stiles = (unsigned)((A.S + 63) / 64);
        bx = idx % stiles;
        by = idx / stiles;
    }
    if (unit_args_skip(A)) return;
    __shared__ double s_tile[64 * 65];
    __shared__ int s_base[64 * kCand];  // buffer index the grain reads at the tile's first sample (mod len)
    __shared__ int s_kd[64 * kCand];    // bits 0-15: k0 + 64 (window index at the tile's first sample; k0 > -64),
                                        // bits 16-30: duration, bit 31: direction (1 = backwards)
    __shared__ int s_cnt[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t S = A.S;
    const size_t s0 = (size_t)bx * 64, c = (size_t)by + A.c0, n0 = c * 64;
    const long long len = (long long)A.len;
    bool listed = true;  // the rows and spawns this tile reads exist
    if constexpr (COH) {
        if (threadIdx.x < 64) {
            const size_t s = s0 + threadIdx.x;
            const int need = (int)c + 2;  // rows c and c + 1
            int spins = 0, seen = -1;
            for (;;) {
                const int have = s < S ? __hip_atomic_load(prog + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__all(have >= need)) break;
                if (__any(have != seen)) spins = 0;  // some stream moved: the count is of polls WITHOUT progress
                seen = have;
                if (++spins > A.spin_limit) {
                    listed = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            if (!listed && threadIdx.x == 0) __hip_atomic_store(A.retry, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- phase 1: one lane per stream collects that stream's candidate grains (creation order) in LDS,
    //      so the dependent metadata loads of 64 streams overlap; all 64-bit arithmetic happens here,
    //      once per grain and tile
    if (threadIdx.x < 64) {
        const size_t s = s0 + threadIdx.x;
        int cnt = 0;
        // interior candidates (bit 8+q of ibits): alive on all 64 samples of the tile, neither buffer index wraps in it
        const bool fullTile1 = (long long)n0 + 64 <= (long long)A.T;
        int ibits = 0;
        if (s < S && listed) {
            auto add = [&](long long born, long long dur, long long pos0, long long sg) {
                // sample k of the grain reads index (pos0 + (k+1)*sg) mod len; at the tile start k = n0 - born
                const long long k0 = (long long)n0 - born;
                const long long base = unit_index(pos0 + (k0 + 1) * sg, len);
                if (fullTile1 && k0 >= 0 && k0 + 63 < dur && (sg > 0 ? base + 64 < len : (base >= 63 && base + 1 < len)))
                    ibits |= 1 << (8 + cnt);
                s_base[threadIdx.x * kCand + cnt] = (int)base;
                s_kd[threadIdx.x * kCand + cnt] =
                    (int)(((unsigned)(k0 + 64) & 0xffffu) | ((unsigned)dur << 16) | (sg < 0 ? 0x80000000u : 0u));
                cnt++;
            };
            if (n0 < 32768) {  // carried-in grains can only be alive during the first <= sr/2 samples
                for (int k = 0; k < kSlots; k++) {
                    const long long dur = (long long)A.gst_in[(3 * kSlots + k) * S + s];
                    if (!dur) continue;
                    const long long idx0 = (long long)A.gst_in[(2 * kSlots + k) * S + s];
                    if (idx0 + (long long)n0 >= dur) continue;
                    const long long pos = (long long)A.gst_in[(0 * kSlots + k) * S + s];
                    const long long inc = (long long)A.gst_in[(1 * kSlots + k) * S + s];
                    // as if born at sample -idx0 from position pos - idx0*inc
                    add(-idx0, dur, pos - idx0 * inc, inc);
                }
            }
            const int first = ld_list<COH>(A.chunk_first + c * S + s), next = ld_list<COH>(A.chunk_first + (c + 1) * S + s);
            // spawns of earlier tiles still alive at n0: births increase, so the live ones are a suffix of
            // [0, first) and at most kSlots long.  Their births/positions are fetched with independent
            // loads (a walk-back loop would chain one dependent load per grain) and filtered afterwards.
            int bornB[kSlots];
            double posB[kSlots], incB[kSlots];
#pragma unroll
            for (int u = 0; u < kSlots; u++) {
                const int j = first - kSlots + u;
                const int jc = j < 0 ? 0 : j;
                bornB[u] = ld_list<COH>(A.spawn_n + (size_t)jc * S + s);
                posB[u] = ld_list<COH>(A.spawn_pos + (size_t)jc * S + s);
                incB[u] = ld_list<COH>(A.spawn_inc + (size_t)jc * S + s);
            }
#pragma unroll
            for (int u = 0; u < kSlots; u++) {
                const int j = first - kSlots + u;
                if (j >= 0 && (long long)bornB[u] + A.sampleDur > (long long)n0) {
                    if (cnt >= kCand) { atomicMax(A.err, 1); break; }
                    add(bornB[u], A.sampleDur, (long long)posB[u], incB[u] > 0 ? 1 : -1);
                }
            }
            if (first > kSlots && (long long)ld_list<COH>(A.spawn_n + (size_t)(first - kSlots - 1) * S + s) + A.sampleDur > (long long)n0)
                atomicMax(A.err, 1);  // more than kSlots earlier spawns alive: the capacity rule of every kernel
            for (int j = first; j < next; j++) {
                if (cnt >= kCand) {
                    atomicMax(A.err, 1);
                    break;
                }
                add(ld_list<COH>(A.spawn_n + (size_t)j * S + s), A.sampleDur, (long long)ld_list<COH>(A.spawn_pos + (size_t)j * S + s),
                    ld_list<COH>(A.spawn_inc + (size_t)j * S + s) > 0 ? 1 : -1);
            }
        }
        s_cnt[threadIdx.x] = cnt | ((cnt > 0 && cnt <= kSlots) ? kFlatFlag : 0) |
                             ((ibits >> 8) == (1 << cnt) - 1 ? kInteriorFlag : 0);
    }
    __syncthreads();
    // ---- phase 2: lanes = 64 consecutive samples of one stream; contiguous sample/window reads, 32-bit math
    const int ilen = (int)A.len;
    const bool inT = (long long)n0 + lane < (long long)A.T;
    // kUnitGroup streams can be rendered together (all their gathers issued before any is consumed).  Measured
    // on config 5: groups of 1 / 2 / 4 -> 2.11 / 2.49 / 3.18 ms end to end, so more loads in flight per wave
    // do not help (occupancy does); the default stays 1.
    // ---- phase 2a: the (grain, tile) pairs of a wavefront's streams, flattened in (stream, creation) order: lane L
    //      stands for candidate L&7 of stream L>>3 of a half (8 streams), the valid ones are walked through a ballot
    //      mask, and kUnitBatch of them are requested back to back before the first is consumed -- one memory
    //      latency per batch instead of per stream (the kernel is latency-, not bandwidth-bound).
    //      Pass 1 takes the stream-tiles whose candidates are all interior (about three in four: a tile sees a birth
    //      or a death every sampleDur/overlaps samples): every lane reads base +- lane from scalar bases, buffer[a]
    //      and buffer[a+1] as one 16-byte request, no per-lane index arithmetic, no predicate.  Pass 2 takes the
    //      others (births, deaths, wraps, the ragged last tile) in the general per-lane form.  Two homogeneous loops:
    //      a per-pair branch inside one loop measured 1.2-1.4 ms against 0.9.  Same products, same creation order.
    double amp0 = A.amp[0];
    asm volatile("" : "+v"(amp0));
    const unsigned voff_f = (unsigned)lane * 8u, voff_b = (unsigned)(63 - lane) * 8u;
    auto flat_pass = [&](auto interior_pass) {
        constexpr bool INTERIOR = decltype(interior_pass)::value;
        for (int h = 0; h < 2; h++) {
            const int st0 = wave * 16 + h * 8;
            const int cw = s_cnt[st0 + (lane >> 3)];
            const bool mine = (cw & kFlatFlag) && (((cw & kInteriorFlag) != 0) == INTERIOR);
            const bool valid = mine && (lane & 7) < (cw & 0xff);
            const int at = (st0 + (lane >> 3)) * kCand + (lane & 7);
            // lanes that stand for no pair keep harmless metadata (k0 = 0, duration 0, forwards, base 0): an exhausted
            // mask makes the rest of a batch read lane 63's
            const int mkd = valid ? s_kd[at] : 64, mbase = valid ? s_base[at] : 0;
            unsigned long long mask = __ballot(valid);
            int cur = -1, alive = 0;
            double total = 0.0;
            auto flush = [&]() {
                if constexpr (!INTERIOR)
                    if (alive > kSlots) atomicMax(A.err, 1);  // same capacity rule as the register-slot kernels
                s_tile[lane * 65 + st0 + cur] = total;
            };
            while (mask) {
                int slot[kUnitBatch];
                double2v vab[kUnitBatch];
                double ve[kUnitBatch];
                bool ok[kUnitBatch], wrap[kUnitBatch];
#pragma unroll
                for (int u = 0; u < kUnitBatch; u++) {
                    slot[u] = (int)__ffsll((long long)mask) - 1;  // wave-uniform; -1 once the mask is exhausted
                    mask &= mask - 1;                              // 0 stays 0
                    const unsigned kd = (unsigned)__builtin_amdgcn_readlane(mkd, slot[u] & 63);
                    const int gbase = __builtin_amdgcn_readlane(mbase, slot[u] & 63);
                    const int gk0 = (int)(kd & 0xffffu) - 64;
                    const bool back = (kd >> 31) != 0;
                    if constexpr (INTERIOR) {
                        // scalar bases + a non-negative per-lane byte offset: base + lane (forwards) or base - lane (backwards)
                        const char *pa = reinterpret_cast<const char *>(A.amp + (back ? gbase - 63 : gbase));
                        const char *pw = reinterpret_cast<const char *>(A.window + gk0);
                        const unsigned off = back ? voff_b : voff_f;
#if MXG_UNIT_DPP
                        // buffer[a] alone (8 bytes a lane: half the address cycles of the 16-byte form); buffer[a + 1] IS the next
                        // lane's buffer[a] (forwards: lane + 1, backwards: lane - 1) and comes by a wavefront shift when the pair is
                        // consumed -- except for the one lane at the end, which loads its neighbour element itself
                        vab[u].x = *reinterpret_cast<const double *>(pa + off);
                        vab[u].y = 0.0;  // (the end lane's neighbour element is 8 bytes after its own in either direction: ONE lane loads it)
                        if (lane == (back ? 0 : 63)) vab[u].y = *reinterpret_cast<const double *>(pa + off + 8);
                        wrap[u] = back;  // (here: the direction of the shift)
#else
                        vab[u] = *reinterpret_cast<const double2v *>(pa + off);  // buffer[a], buffer[a+1] as one 16-byte request
                        wrap[u] = false;
#endif
                        ve[u] = *reinterpret_cast<const double *>(pw + voff_f);
                        ok[u] = true;
                    } else {
                        const int gdur = (int)((kd >> 16) & 0x7fffu), gsgn = back ? -1 : 1;
                        const int k = gk0 + lane;
                        ok[u] = inT && k >= 0 && k < gdur;
                        int ia = gbase + lane * gsgn;  // |lane*sgn| < 64 <= len
                        if (ia >= ilen) ia -= ilen;
                        if (ia < 0) ia += ilen;
                        wrap[u] = ia + 1 >= ilen;  // b = 0 (:231-233): patched in from amp0 when consumed
                        // ([len] is readable: mxg_sample_upload's guard)
                        vab[u] = *reinterpret_cast<const double2v *>(A.amp + ia);
                        ve[u] = A.window[ok[u] ? k : 0];
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnitBatch; u++) {
                    if (slot[u] >= 0) {
                        const int stq = slot[u] >> 3;
                        if (stq != cur) {
                            if (cur >= 0) flush();
                            cur = stq;
                            total = 0.0;
                            alive = 0;
                        }
                        const double remainder = 0.0;  // pos is an integer: pos - floor(pos)
                        if constexpr (INTERIOR) {
#if MXG_UNIT_DPP
                            {
                                const int alo = __double2loint(vab[u].x), ahi = __double2hiint(vab[u].x);
                                const int elo = __double2loint(vab[u].y), ehi = __double2hiint(vab[u].y);
                                int blo, bhi;
                                if (wrap[u]) {  // (wave-uniform) backwards: lane L's buffer[a + 1] is lane L - 1's buffer[a]
                                    blo = __builtin_amdgcn_update_dpp(elo, alo, 0x138, 0xf, 0xf, false);  // wave_shr:1
                                    bhi = __builtin_amdgcn_update_dpp(ehi, ahi, 0x138, 0xf, 0xf, false);
                                } else {        // forwards: lane L + 1's
                                    blo = __builtin_amdgcn_update_dpp(elo, alo, 0x130, 0xf, 0xf, false);  // wave_shl:1
                                    bhi = __builtin_amdgcn_update_dpp(ehi, ahi, 0x130, 0xf, 0xf, false);
                                }
                                vab[u].y = __hiloint2double(bhi, blo);
                            }
#endif
                            double o = ((1 - remainder) * vab[u].x + remainder * vab[u].y);  // :236-237, literally
                            o *= ve[u];
                            total += o;  // creation order
                        } else if (ok[u]) {
                            const double b = wrap[u] ? amp0 : vab[u].y;
                            double o = ((1 - remainder) * vab[u].x + remainder * b);
                            o *= ve[u];
                            total += o;
                            alive++;
                        }
                    }
                }
            }
            if (cur >= 0) flush();
        }
    };
    flat_pass(std::true_type{});
    flat_pass(std::false_type{});
    // ---- phase 2b: stream-tiles with no candidate (silence) or more than kSlots of them, one stream after the other
    for (int sg = wave * 16; sg < wave * 16 + 16; sg += kUnitGroup) {
        int cnt[kUnitGroup], cmax = 0;
        bool todo = false;
#pragma unroll
        for (int g = 0; g < kUnitGroup; g++) {
            const int cw = s_cnt[sg + g];
            todo = todo || !(cw & kFlatFlag);
            cnt[g] = (cw & kFlatFlag) ? 0 : (cw & 0xff);
            cmax = cnt[g] > cmax ? cnt[g] : cmax;
        }
        if (!todo) continue;
        double total[kUnitGroup];
        int alive[kUnitGroup];
#pragma unroll
        for (int g = 0; g < kUnitGroup; g++) { total[g] = 0.0; alive[g] = 0; }
        // lane q (< kCand) fetches candidate q's metadata once; readlane turns it into scalars per grain
        int mkd[kUnitGroup], mbase[kUnitGroup];
#pragma unroll
        for (int g = 0; g < kUnitGroup; g++) {
            const int at = (sg + g) * kCand + (lane < kCand ? lane : 0);
            mkd[g] = s_kd[at]; mbase[g] = s_base[at];
        }
        for (int q0 = 0; q0 < cmax; q0 += 8) {
            double va[kUnitGroup][8], vb[kUnitGroup][8], ve[kUnitGroup][8];
            bool ok[kUnitGroup][8];
#pragma unroll
            for (int g = 0; g < kUnitGroup; g++) {
#pragma unroll
                for (int u = 0; u < 8; u++) {  // issue the reads of up to 8 grains of every stream of the group
                    const int q = q0 + u;
                    ok[g][u] = false;
                    va[g][u] = vb[g][u] = ve[g][u] = 0.0;
                    if (q < cnt[g]) {  // wave-uniform
                        const unsigned kd = (unsigned)__builtin_amdgcn_readlane(mkd[g], q);
                        const int gbase = __builtin_amdgcn_readlane(mbase[g], q);
                        const int gk0 = (int)(kd & 0xffffu) - 64, gdur = (int)((kd >> 16) & 0x7fffu);
                        const int gsgn = (kd >> 31) ? -1 : 1;
                        const int k = gk0 + lane;
                        ok[g][u] = inT && k >= 0 && k < gdur;
                        int ia = gbase + lane * gsgn;  // |lane*sgn| < 64 <= len
                        if (ia >= ilen) ia -= ilen;
                        if (ia < 0) ia += ilen;
                        int ib = ia + 1;
                        if (ib >= ilen) ib = 0;  // :231-233
                        va[g][u] = A.amp[ia];
                        vb[g][u] = A.amp[ib];
                        ve[g][u] = A.window[ok[g][u] ? k : 0];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < kUnitGroup; g++) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (ok[g][u]) {
                        const double remainder = 0.0;  // pos is an integer: pos - floor(pos)
                        double o = ((1 - remainder) * va[g][u] + remainder * vb[g][u]);  // :236-237, literally
                        o *= ve[g][u];
                        total[g] += o;  // creation order
                        alive[g]++;
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < kUnitGroup; g++) {
            if (s_cnt[sg + g] & kFlatFlag) continue;  // rendered in phase 2a
            if (alive[g] > kSlots) atomicMax(A.err, 1);  // same capacity rule as the register-slot kernels
            s_tile[lane * 65 + sg + g] = total[g];
        }
    }
    __syncthreads();
    tile_epilogue(A, s_tile, s0, n0, lane, wave, bx);
}
template <int SMODE>
__global__ __launch_bounds__(256, 4) void granular_unit_kernel(UnitArgs A, SchedArgs Q, int *prog, unsigned nsched) {
    granular_unit_body<SMODE>(A, Q, prog, nsched, blockIdx.x, blockIdx.y);
}

// ---- K8d: tile render for arbitrary increments (maxiStretch, maxiPitchShift, maxiTimeStretch off the integer grid) ------
// K8b gives a lane one (stream, chunk) and walks it in time: every wavefront load touches 64 different grains, i.e. 64
// different cache lines, five times per sample -- 320 live lines per wavefront against a 32 KB L1: every gather misses, and
// the kernel moves ~90 GB from L2 for 1.2 GB of output (4.5-4.9 ms for the config-5 shape whatever the number of lanes).
// Here, as in K8c, the lanes of a wavefront are 64 CONSECUTIVE SAMPLES of one stream, so a grain's reads of one tile are
// neighbours (64*inc elements) and its window reads contiguous.  That needs sample k of a grain without walking to it.  The
// position is a recurrence, pos <- fl(pos + inc) with a wrap at len (maxiGrain::play, L/maxiGrains.h:222-228), not a closed
// form -- but inside one binade it IS one on the mantissa grid: X_k = X + k*c with c the (constant) number of ulps
// round-to-nearest adds per step (add_line, mxg_advance.h; fuzzed on the host against the recurrence).  Per (grain, tile):
//   phase 1  the position at the tile's first sample ("anchor") by the exact multi-step advance_until from the grain's
//            start, and the line through it if the next 65 steps stay inside its binade without reaching len;
//   phase 2  lane L evaluates step L + 1 (+ offset for grains born inside the tile) on the line, reads buffer[a], buffer[a+1]
//            as one 16-byte request and the window, and adds the product in creation order.
// A (grain, tile) pair without a line (the tile in which the grain wraps or crosses a binade, a tie, a backward grain)
// walks its at most 64 steps per lane.  Same additions, same products, same order: bit-identical to K8 / K8b.
// A workgroup renders kLineTiles consecutive tiles and carries the anchors from one to the next on their lines, so the
// multi-step advance (~1 us per grain with its divisions) runs once per grain and span; the render is sliced against the
// scheduler like K8c.  Measured, config-5 shape: maxiStretch 4.91 -> 2.60 ms, maxiPitchShift 4.45 -> 2.81 ms per call.  What is
// left is not latency (4, 8 or 16 pairs in flight: the same time), not instruction count (halving it changed nothing) and not
// bytes: SQ counters show ~1200 wavefront cycles per (grain, tile) pair, half of them on vmcnt -- every lane of the 16-byte
// gather is its own address for the texture path, ~1 lane per clock per CU, and a grain-sample needs one whatever the kernel.
// Tried on top and dropped: fetching a pair's span (64*inc + 2 elements) with ONE coalesced wavefront load into LDS and picking
// buffer[a], buffer[a+1] from there -- 0.54 -> 0.93 ms per slice (16 KB more LDS: two workgroups per CU instead of three).  The
// kernel is bound by its phase structure at 8-12 resident wavefronts per CU (four barriers per tile, the per-tile anchor update
// with a dependent global load, 64-lane-wide phases 1a); more tiles in flight per CU is the lever, i.e. less LDS per tile.
struct LineCand {  // 24 bytes: 64 streams x 12 candidates + the 33 KB tile keep a workgroup under 53 KB (three per CU)
    double anchor;  // position after the steps that precede the current tile (the grain's start position until it is born)
    long long c;    // ulps per step on the anchor's binade, when the line holds
    int src;        // where the grain's increment lives: j >= 0 spawn j of the stream, -(k+1) carried-in slot k
    short kb;       // age of the grain at the current tile's first sample: window index of lane L is kb + L
    unsigned char flags;  // bit 0: the line holds for steps 1 .. 65
    unsigned char negoff; // lane L evaluates step L + 1 - negoff from the anchor (> 0: born inside the tile; 64: not born yet)
};

__device__ __forceinline__ double grain_advance_fast(double pos, const double inc, const double dlen, int steps) {
    if (!(inc > 0.0) || !(pos >= 0.0)) return grain_advance(pos, inc, dlen, steps);  // backwards / NaN: the plain walk
    while (steps > 0) {
        bool crossed;
        steps -= advance_until(pos, inc, dlen, true, steps, crossed);
        if (crossed) pos -= dlen;  // :223-224
    }
    return pos;
}

// grains alive after sample T-1, creation order (one lane per stream): positions by the exact multi-step advance
template <bool COH>
__device__ __forceinline__ void line_state_lane(const UnitArgs &A, const size_t s) {
    const size_t S = A.S;
    const double dlen = (double)A.len;
    const long long T = (long long)A.T;
    double gp[kSlots], gi[kSlots], gx[kSlots], gd[kSlots];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kSlots; k++) { gp[k] = gi[k] = gx[k] = gd[k] = 0.0; }
    auto push = [&](double p, double i, double x, double d) {
#pragma unroll
        for (int k = 0; k < kSlots; k++)
            if (k == cnt) { gp[k] = p; gi[k] = i; gx[k] = x; gd[k] = d; }
        cnt++;
    };
    for (int k = 0; k < kSlots; k++) {
        const double ddur = A.gst_in[(3 * kSlots + k) * S + s];
        if (ddur == 0.0) continue;
        const double pos = A.gst_in[(0 * kSlots + k) * S + s], inc = A.gst_in[(1 * kSlots + k) * S + s];
        const double didx = A.gst_in[(2 * kSlots + k) * S + s];
        if (!carried_grain_ok(pos, inc, didx, ddur, dlen, A.sampleDur)) continue;
        const long long idx0 = (long long)didx;
        if (idx0 + T >= (long long)A.sampleDur) continue;
        if (cnt < kSlots) push(grain_advance_fast(pos, inc, dlen, (int)T), inc, (double)(idx0 + T), ddur);
    }
    const int count = ld_list<COH>(A.chunk_first + A.C * S + s);
    int j0 = count;
    while (j0 > 0 && (long long)ld_list<COH>(A.spawn_n + (size_t)(j0 - 1) * S + s) + A.sampleDur > T) j0--;
    for (int j = j0; j < count; j++) {
        const long long born = ld_list<COH>(A.spawn_n + (size_t)j * S + s);
        const double inc = ld_list<COH>(A.spawn_inc + (size_t)j * S + s);
        const long long steps = T - born;
        if (cnt < kSlots)
            push(grain_advance_fast(ld_list<COH>(A.spawn_pos + (size_t)j * S + s), inc, dlen, (int)steps), inc, (double)steps, (double)A.sampleDur);
    }
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        A.gst_out[(0 * kSlots + k) * S + s] = gp[k];
        A.gst_out[(1 * kSlots + k) * S + s] = gi[k];
        A.gst_out[(2 * kSlots + k) * S + s] = gx[k];
        A.gst_out[(3 * kSlots + k) * S + s] = gd[k];
    }
}
__global__ __launch_bounds__(64) void granular_line_state_kernel(UnitArgs A) {
    if (unit_args_skip(A)) return;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.S) return;
    line_state_lane<false>(A, s);
}

constexpr int kLineTiles = 4;  // consecutive tiles per workgroup: the anchors are carried from tile to tile

constexpr int kLineBatch = 4;  // pairs whose gathers are in flight together (4 / 8 / 16 measured: 0.54 / 0.60 / 0.87 ms per slice)

// SMODE as for granular_unit_kernel: < 0 the renderer alone, 0 .. 3 the streamed form of that scheduler mode (1-D grid: `nsched`
// scheduler workgroups, then one workgroup per (stream tile, span of ltiles tiles), spans in dispatch order).
template <int SMODE>
__device__ __forceinline__ void granular_line_body(const UnitArgs &A, const SchedArgs &Q, int *prog, const unsigned nsched, const unsigned bxi,
                                                   const unsigned byi) {
    constexpr bool COH = SMODE >= 0;
    unsigned bx = bxi, by = byi;
    if constexpr (COH) {
        if (bxi < nsched) {
            const size_t s = (size_t)bxi * 256 + threadIdx.x;
            if (s >= A.S) return;
            if (unit_args_skip(A)) return;  // (both renderers are enqueued and the other one was picked: it schedules, too)
            __builtin_amdgcn_s_setprio(3);
            sched_walk<(SMODE >= 0 ? SMODE : 0), true>(Q, s, prog);
            line_state_lane<true>(A, s);
            return;
        }
        const unsigned idx = bxi - nsched, stiles = (unsigned)((A.S + 63) / 64);
        bx = idx % stiles;
        by = idx / stiles;
    }
    if (unit_args_skip(A)) return;
    __shared__ double s_tile[64 * 65];
    __shared__ LineCand s_cand[64 * kCand];
    __shared__ int s_cnt[64];
    // (the wave index in a scalar register: the walk over (stream, candidate) pairs below is scalar control flow only then)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t S = A.S;
    const size_t s0 = (size_t)bx * 64;
    const size_t cA = (size_t)by * A.ltiles + A.c0;
    const size_t cB = cA + A.ltiles < A.cend ? cA + A.ltiles : A.cend;
    const size_t nA = cA * 64;  // first sample of the span
    const double dlen = (double)A.len;
    auto inc_of = [&](const int src, const size_t s) {
        return src >= 0 ? ld_list<COH>(A.spawn_inc + (size_t)src * S + s) : A.gst_in[(1 * kSlots + (size_t)(-src - 1)) * S + s];
    };
    bool listed = true;  // the rows and spawns this span reads exist
    if constexpr (COH) {
        if (threadIdx.x < 64) {
            const size_t s = s0 + threadIdx.x;
            const int need = (int)cB + 1;  // rows cA .. cB
            int spins = 0, seen = -1;
            for (;;) {
                const int have = s < S ? __hip_atomic_load(prog + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__all(have >= need)) break;
                if (__any(have != seen)) spins = 0;  // some stream moved: the count is of polls WITHOUT progress
                seen = have;
                if (++spins > A.spin_limit) {
                    listed = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            if (!listed && threadIdx.x == 0) __hip_atomic_store(A.retry, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    auto set_age = [&](LineCand &q, const long long kb) {  // age at the current tile's first sample (< 0: not born yet)
        q.kb = (short)(kb < -32768 ? -32768 : kb);
        q.negoff = (unsigned char)(kb >= 0 ? 0 : (kb < -64 ? 64 : -kb));
    };
    // ---- phase 1a: one lane per stream lists the grains that play anywhere in the span, in creation order
    if (threadIdx.x < 64) {
        const size_t s = s0 + threadIdx.x;
        int cnt = 0;
        if (s < S && listed) {
            // a grain born at sample `born`; its reference position `pos` is the one it has at age `age`
            auto add = [&](long long born, long long age, double pos, int src) {
                LineCand &q = s_cand[threadIdx.x * kCand + cnt];
                const long long kb = (long long)nA - born;
                q.anchor = pos;
                q.c = kb - age > 0 ? kb - age : 0;  // phase 1b: steps from the reference position to the span's first sample
                q.src = src;
                q.flags = 0;
                set_age(q, kb);
                cnt++;
            };
            if (nA < 32768) {  // carried-in grains can only be alive during the first <= sr/2 samples
                for (int k = 0; k < kSlots; k++) {
                    const double ddur = A.gst_in[(3 * kSlots + k) * S + s];
                    if (ddur == 0.0) continue;
                    const double pos = A.gst_in[(0 * kSlots + k) * S + s], inc = A.gst_in[(1 * kSlots + k) * S + s];
                    const double didx = A.gst_in[(2 * kSlots + k) * S + s];
                    if (!carried_grain_ok(pos, inc, didx, ddur, dlen, A.sampleDur)) {
                        atomicMax(A.err, 4);
                        continue;
                    }
                    const long long idx0 = (long long)didx;
                    if (idx0 + (long long)nA >= (long long)A.sampleDur) continue;  // finished before the span
                    add(-idx0, idx0, pos, -(k + 1));
                }
            }
            const int first = ld_list<COH>(A.chunk_first + cA * S + s), next = ld_list<COH>(A.chunk_first + cB * S + s);
            int bornB[kSlots];
            double posB[kSlots];
#pragma unroll
            for (int u = 0; u < kSlots; u++) {  // independent loads (a walk-back loop would chain them), filtered afterwards
                const int j = first - kSlots + u;
                const int jc = j < 0 ? 0 : j;
                bornB[u] = ld_list<COH>(A.spawn_n + (size_t)jc * S + s);
                posB[u] = ld_list<COH>(A.spawn_pos + (size_t)jc * S + s);
            }
#pragma unroll
            for (int u = 0; u < kSlots; u++) {
                const int j = first - kSlots + u;
                if (j >= 0 && (long long)bornB[u] + A.sampleDur > (long long)nA) {
                    if (cnt >= kCand) { atomicMax(A.err, 1); break; }
                    add(bornB[u], 0, posB[u], j);
                }
            }
            if (first > kSlots && (long long)ld_list<COH>(A.spawn_n + (size_t)(first - kSlots - 1) * S + s) + A.sampleDur > (long long)nA)
                atomicMax(A.err, 1);  // more than kSlots earlier spawns alive: the capacity rule of every kernel
            for (int j = first; j < next; j++) {
                if (cnt >= kCand) {
                    atomicMax(A.err, 1);
                    break;
                }
                add(ld_list<COH>(A.spawn_n + (size_t)j * S + s), 0, ld_list<COH>(A.spawn_pos + (size_t)j * S + s), j);
            }
        }
        s_cnt[threadIdx.x] = cnt;
    }
    __syncthreads();
    for (size_t c = cA; c < cB; c++) {
        const size_t n0 = c * 64;
        // ---- phase 1b / 1c, all 256 lanes: the anchor of every (stream, candidate) pair at this tile's first sample -- from the
        //      grain's reference position by the exact multi-step advance (first tile), from the previous tile's anchor by the
        //      steps that tile took (on its line, or walked) -- and the line through it
        for (int pidx = threadIdx.x; pidx < 64 * kCand; pidx += 256) {
            const int st = pidx / kCand, q = pidx - st * kCand;
            if (q >= s_cnt[st]) continue;
            LineCand &cd = s_cand[pidx];
            const double inc = inc_of(cd.src, s0 + st);
            double anchor = cd.anchor;
            if (c == cA) {
                anchor = grain_advance_fast(anchor, inc, dlen, (int)cd.c);
            } else {
                const int kb = cd.kb;  // age at the previous tile's first sample
                const int steps = kb >= 0 ? 64 : (kb > -64 ? 64 + kb : 0);
                if (steps > 0) {
                    if (cd.flags & 1) {
                        const long long ab = __double_as_longlong(anchor);
                        const long long Xk = ((ab & 0xFFFFFFFFFFFFFLL) | (1LL << 52)) + (long long)steps * cd.c;
                        anchor = __longlong_as_double((ab & 0x7FF0000000000000LL) | (Xk & 0xFFFFFFFFFFFFFLL));
                    } else {
                        anchor = grain_advance(anchor, inc, dlen, steps);
                    }
                }
                set_age(cd, (long long)kb + 64);
            }
            const AddLine l = add_line(anchor, inc, dlen, 65);
            cd.anchor = anchor;
            cd.c = l.c;
            cd.flags = l.ok ? 1 : 0;
        }
        __syncthreads();
        // ---- phase 2: lanes = 64 consecutive samples of one stream.  The (stream, candidate) pairs of the wavefront's 16
        //      streams are walked in (stream, creation) order, kLineBatch at a time: all their gathers are requested before the
        //      first is consumed (the kernel is latency-bound: one exposed memory latency per batch), and a stream's sum is
        //      flushed to the tile when the walk moves on to the next stream.
        const bool inT = (long long)n0 + lane < (long long)A.T;
        const double amp0 = A.amp[0];
#pragma unroll
        for (int i = 0; i < 16; i++) s_tile[lane * 65 + wave * 16 + i] = 0.0;  // streams without a candidate stay silent
        // The pairs of the wavefront's 16 streams, flattened: lane p of a round stands for pair base + p and fetches its metadata
        // (one LDS round trip per 64 pairs); the walk below takes each pair's values from that lane with v_readlane.
        int pre = lane < 16 ? s_cnt[wave * 16 + lane] : 0;  // -> inclusive prefix sums of the 16 candidate counts
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int t = __shfl_up(pre, d);
            if (lane >= d) pre += t;
        }
        const int P = __builtin_amdgcn_readlane(pre, 15);
        int cur = -1, alive = 0;
        double total = 0.0;
        auto flush = [&]() {
            if (alive > kSlots) atomicMax(A.err, 1);  // same capacity rule as the register-slot kernels
            if (cur >= 0) s_tile[lane * 65 + wave * 16 + cur] = total;
        };
        for (int base = 0; base < P; base += 64) {
            const int p = base + lane;
            int stl = 0;  // stream (0..15) of pair p: the number of inclusive prefixes <= p
#pragma unroll
            for (int i = 0; i < 15; i++) stl += (__builtin_amdgcn_readlane(pre, i) <= p) ? 1 : 0;
            const int before = __shfl(pre, stl > 0 ? stl - 1 : 0);
            const int ql = p - (stl > 0 ? before : 0);
            const bool have = p < P;
            const LineCand *cp = s_cand + (wave * 16 + stl) * kCand + (have ? ql : 0);
            const long long m_anchor = have ? __double_as_longlong(cp->anchor) : 0;
            const long long m_c = have ? cp->c : 0;
            const int m_src = have ? cp->src : 0;
            const int m_meta = have ? (((int)cp->kb << 16) | ((int)cp->negoff << 8) | (int)cp->flags) : 0;
            const int m_alo = (int)m_anchor, m_ahi = (int)(m_anchor >> 32), m_clo = (int)m_c, m_chi = (int)(m_c >> 32);
            const int nround = P - base < 64 ? P - base : 64;
            for (int j0 = 0; j0 < nround; j0 += kLineBatch) {
                int pst[kLineBatch];
                double2v vab[kLineBatch];
                double ve[kLineBatch], rem[kLineBatch];
                bool ok[kLineBatch], wrapb[kLineBatch];
#pragma unroll
                for (int u = 0; u < kLineBatch; u++) {
                    pst[u] = -1;
                    ok[u] = false;
                    wrapb[u] = false;
                    rem[u] = 0.0;
                    ve[u] = 0.0;
                    vab[u] = double2v{0.0, 0.0};
                    const int j = j0 + u;
                    if (j < nround) {  // wave-uniform
                        pst[u] = __builtin_amdgcn_readlane(stl, j);
                        const unsigned long long ab = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(m_ahi, j) << 32) |
                                                      (unsigned)__builtin_amdgcn_readlane(m_alo, j);
                        const unsigned clo = (unsigned)__builtin_amdgcn_readlane(m_clo, j);
                        const unsigned chi = (unsigned)__builtin_amdgcn_readlane(m_chi, j);
                        const int meta = __builtin_amdgcn_readlane(m_meta, j);
                        const int kb = meta >> 16;
                        // window index and step of this lane; the lanes outside the grain's life read clamped (valid) addresses
                        const unsigned ku = (unsigned)(kb + lane);
                        const unsigned m = (unsigned)(lane + 1 - ((meta >> 8) & 0xff));
                        ok[u] = inT && ku < (unsigned)A.sampleDur;
                        double pos;
                        if (meta & 1) {
                            // step m on the line: inside the binade, adding m*c ulps is adding m*c to the bit pattern (m <= 64, c < 2^52)
                            const unsigned long long pb = ab + (unsigned long long)m * clo + ((unsigned long long)(m * chi) << 32);
                            pos = __longlong_as_double((long long)pb);
                        } else {
                            const double inc = inc_of(__builtin_amdgcn_readlane(m_src, j), s0 + wave * 16 + pst[u]);
                            pos = __longlong_as_double((long long)ab);
                            for (unsigned i = 0; i < 64; i++) {
                                if (i < m && ok[u]) {
                                    double p2 = pos + inc;  // :222-226
                                    if (p2 >= dlen)
                                        p2 -= dlen;
                                    else if (p2 < 0)
                                        p2 += dlen;
                                    pos = p2;
                                }
                            }
                        }
                        // 0 <= pos < len < 2^31 on every live lane: floor and the index through the 32-bit conversions
                        const int ia32 = (int)pos;
                        rem[u] = pos - (double)ia32;  // pos - floor(pos)
                        const unsigned iu = (unsigned)ia32 < (unsigned)A.len ? (unsigned)ia32 : 0u;
                        wrapb[u] = (size_t)iu + 1 >= A.len;                       // :231-233
                        vab[u] = *reinterpret_cast<const double2v *>(A.amp + iu);  // buffer[a], buffer[a+1]: [len] is readable (guard)
                        const unsigned kc = ku < (unsigned)A.sampleDur ? ku : 0u;
                        ve[u] = A.window[kc];
                    }
                }
#pragma unroll
                for (int u = 0; u < kLineBatch; u++) {
                    if (pst[u] >= 0) {  // wave-uniform
                        if (pst[u] != cur) {
                            flush();
                            cur = pst[u];
                            total = 0.0;
                            alive = 0;
                        }
                        if (ok[u]) {
                            const double vb = wrapb[u] ? amp0 : vab[u].y;
                            double o = ((1 - rem[u]) * vab[u].x + rem[u] * vb);  // :236-237
                            o *= ve[u];
                            total += o;  // creation order
                            alive++;
                        }
                    }
                }
            }
        }
        flush();
        __syncthreads();
        tile_epilogue(A, s_tile, s0, n0, lane, wave, bx);
        __syncthreads();  // the tile and the candidate list are reused by the next tile of the span
    }
}
template <int SMODE>
__global__ __launch_bounds__(256, 3) void granular_line_kernel(UnitArgs A, SchedArgs Q, int *prog, unsigned nsched) {
    granular_line_body<SMODE>(A, Q, prog, nsched, blockIdx.x, blockIdx.y);
}

// The retry of a streamed launch whose renderers gave up (see granular_unit_body): grid = (stream tiles, a few rows), every workgroup
// returns at once unless the stream's retry word is set; otherwise it renders rows blockIdx.y, blockIdx.y + gridDim.y, ... of the call
// with the renderer-alone form (rows = tiles for K8c, spans of ltiles tiles for K8d) -- ALL of them, so that nothing depends on which
// tiles had given up.  The lists are complete: this kernel starts after the streamed launch has ended.
template <bool LINE>
__global__ __launch_bounds__(256) void granular_retry_kernel(UnitArgs A, SchedArgs Q, unsigned rows) {
    if (__hip_atomic_load(A.retry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    for (unsigned by = blockIdx.y; by < rows; by += gridDim.y) {
        if constexpr (LINE) granular_line_body<-1>(A, Q, nullptr, 0u, blockIdx.x, by);
        else granular_unit_body<-1>(A, Q, nullptr, 0u, blockIdx.x, by);
        __syncthreads();
    }
}

// The first kernel of a chunked call: keeps a copy of the carried-in grains (the renders read the copy, the state kernels
// overwrite d_gst while they run), clears the streamed form's progress words and, with `check`, decides K8c's eligibility: every live carried-in grain must have
// inc = +-1 and an integer position (*flag = 1 otherwise; the word is zero between calls, see grain_err_publish_kernel).
__global__ void granular_prologue_kernel(size_t S, const double *__restrict__ gst, double *__restrict__ gst_copy, double dlen,
                                         int sampleDur, int check, int *flag, int32_t *prog) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    prog[s] = 0;  // the streamed form's progress words: no chunk-table row of this call exists yet
    bool bad = false;
    double v[4][kSlots];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < kSlots; k++) v[j][k] = gst[(j * kSlots + k) * S + s];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < kSlots; k++) gst_copy[(j * kSlots + k) * S + s] = v[j][k];
    if (!check) return;
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        if (v[3][k] == 0.0) continue;
        const double pos = v[0][k], inc = v[1][k];
        // (a grain the plan could not have made goes to the general path, which refuses it)
        if (!carried_grain_ok(pos, inc, v[2][k], v[3][k], dlen, sampleDur)) bad = true;
        if (!(inc == 1.0 || inc == -1.0) || pos != floor(pos) || pos < 0.0 || pos > 9.0e15) bad = true;
        if (v[3][k] >= 32000.0) bad = true;  // durations are packed into 15 bits by K8c
    }
    if (bad) atomicMax(flag, 1);
}


}  // namespace
}  // namespace mxg

using namespace mxg;

namespace {
// The auxiliary stream the sliced unit path renders on, and its fork/join events (created once, never destroyed:
// they live as long as the library).  Calls from several host threads share them; every use is ordered by events.
constexpr int kMaxSlices = 32;
hipStream_t g_aux = nullptr, g_aux2 = nullptr;  // g_aux2: the renderer that is NOT taken when both are enqueued (UnitArgs::sel)
hipEvent_t g_aux_ev[kMaxSlices], g_aux_done, g_aux2_done;
std::mutex g_aux_mu;
std::mutex g_aux_init_mu;
int aux_stream_init() {
    std::lock_guard<std::mutex> lock(g_aux_init_mu);
    if (g_aux) return MXG_OK;
    hipStream_t a = nullptr;
    MXG_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    MXG_HIP(hipStreamCreateWithFlags(&g_aux2, hipStreamNonBlocking));
    for (int i = 0; i < kMaxSlices; i++) MXG_HIP(hipEventCreateWithFlags(&g_aux_ev[i], hipEventDisableTiming));
    MXG_HIP(hipEventCreateWithFlags(&g_aux_done, hipEventDisableTiming));
    MXG_HIP(hipEventCreateWithFlags(&g_aux2_done, hipEventDisableTiming));
    g_aux = a;
    return MXG_OK;
}
}  // namespace

extern "C" {

mxg_grain_plan *mxg_grain_plan_create(int window_kind, double grainLength, int mySampleRate) {
    if (window_kind < 0 || window_kind > 8 || !(grainLength > 0) || mySampleRate <= 0) {
        fail(MXG_ERR_INVALID, "mxg_grain_plan_create: bad argument");
        return nullptr;
    }
    const unsigned long sampleDur = grainLength * (double)mySampleRate;  // L/maxiGrains.h:164
    const unsigned long cacheSize = settings().sampleRate / 2.0;         // :98
    if (sampleDur == 0 || sampleDur >= cacheSize) {
        fail(MXG_ERR_INVALID, "mxg_grain_plan_create: grain of %lu samples outside (0, %lu) -- the "
                              "reference's window cache holds up to 500 ms (maxiGrains.h:98)", sampleDur, cacheSize);
        return nullptr;
    }
    mxg_grain_plan *p = new mxg_grain_plan();
    p->window_kind = window_kind;
    p->mySampleRate = mySampleRate;
    p->grainLength = grainLength;
    p->sampleDur = sampleDur;
    p->d_window = nullptr;
    p->h_window.resize(sampleDur);
    for (unsigned long i = 0; i < sampleDur; i++) p->h_window[i] = window_value(window_kind, sampleDur, i);
    if (!ensure_init_only()) {
        if (check_hip(hipMalloc(&p->d_window, sizeof(double) * sampleDur), "hipMalloc") ||
            check_hip(hipMemcpy(p->d_window, p->h_window.data(), sizeof(double) * sampleDur, hipMemcpyHostToDevice),
                      "hipMemcpy")) {
            if (p->d_window) (void)hipFree(p->d_window);
            p->d_window = nullptr;
        }
    }
    return p;
}

int mxg_grain_plan_destroy(mxg_grain_plan *p) {
    if (!p) return MXG_OK;
    if (p->d_window) (void)hipFree(p->d_window);
    delete p;
    return MXG_OK;
}

int mxg_grain_plan_window(const mxg_grain_plan *p, double *h_window) {
    MXG_REQUIRE(p, "null plan");
    if (h_window) memcpy(h_window, p->h_window.data(), sizeof(double) * p->h_window.size());
    return (int)p->sampleDur;
}

// forwards a render's error word (0 = fine) to the library's async error word (mxg_common.h) and leaves both words of the
// stream (error | K8c eligibility) at zero for the next call: the last kernel of a call, so a call starts without a memset
__global__ void grain_err_publish_kernel(int *err, int *async_word) {
    const int e = err[0];
    if (e && async_word) __hip_atomic_store(async_word, (int)mxg::ASYNC_GRAIN_BASE + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // a streamed launch that was rendered again (err[2]): counted in the word behind the async error word (mxg_granular_retries)
    if (err[2] && async_word) __hip_atomic_fetch_add(async_word + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    err[0] = 0;
    err[1] = 0;
    err[2] = 0;
}

static int granular_render_impl(const mxg_grain_plan *p, int mode, size_t S, size_t T, const double *d_samples,
                                size_t len, int overlaps, const double *d_a, const double *d_b,
                                const double *d_posmod, const int32_t *d_rnd, size_t R, double *d_st, double *d_gst,
                                double *d_out, const double *d_pan, double *d_mix, void *stream);

int mxg_granular_render(const mxg_grain_plan *p, int mode, size_t S, size_t T, const double *d_samples,
                        size_t len, int overlaps, const double *d_a, const double *d_b,
                        const double *d_posmod, const int32_t *d_rnd, size_t R, double *d_st, double *d_gst,
                        double *d_out, void *stream) {
    return granular_render_impl(p, mode, S, T, d_samples, len, overlaps, d_a, d_b, d_posmod, d_rnd, R, d_st, d_gst, d_out,
                                nullptr, nullptr, stream);
}

int mxg_granular_render_mix(const mxg_grain_plan *p, int mode, size_t S, size_t T, const double *d_samples,
                            size_t len, int overlaps, const double *d_a, const double *d_b,
                            const double *d_posmod, const int32_t *d_rnd, size_t R, double *d_st, double *d_gst,
                            double *d_out, const double *d_pan, double *d_mix, void *stream) {
    MXG_REQUIRE(d_pan && d_mix, "null pan / mix");
    return granular_render_impl(p, mode, S, T, d_samples, len, overlaps, d_a, d_b, d_posmod, d_rnd, R, d_st, d_gst, d_out,
                                d_pan, d_mix, stream);
}

static int granular_render_impl(const mxg_grain_plan *p, int mode, size_t S, size_t T, const double *d_samples,
                                size_t len, int overlaps, const double *d_a, const double *d_b,
                                const double *d_posmod, const int32_t *d_rnd, size_t R, double *d_st, double *d_gst,
                                double *d_out, const double *d_pan, double *d_mix, void *stream) {
    if (int s = ensure_init()) return s;
    bool mixed = false;  // the unit path mixes inside its render kernel; every other path runs K3 afterwards
    MXG_REQUIRE(p && p->d_window, "null plan (or plan created without a HIP device)");
    MXG_REQUIRE(mode >= 0 && mode <= 3,
                "mode must be 0 (maxiTimeStretch::play), 1 (maxiStretch::play), 2 (playAtPosition) or 3 "
                "(maxiPitchShift::play)");
    MXG_REQUIRE(d_samples && d_a && d_st && d_gst && d_out, "null device pointer");
    MXG_REQUIRE(mode != 1 || d_b, "maxiStretch needs d_b (timestretch)");
    MXG_REQUIRE(len > 0 && overlaps > 0, "empty sample or overlaps <= 0");
    if (S == 0 || T == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    int *g_err = nullptr;  // per-stream words: the render's error | K8c eligibility.  Zero between calls: the publish kernel that ends a
                           // call clears them (the synchronous A/B path: a memset after its read-back), so only a new allocation pays a memset.
    bool err_fresh = false;
    if (int s = scratch_get(SCR_GRAIN_ERR, st, 4 * sizeof(int), (void **)&g_err, &err_fresh)) return s;  // (+ the streamed form's retry word)
    if (err_fresh) MXG_HIP(hipMemsetAsync(g_err, 0, 4 * sizeof(int), st));
    GrainArgs A;
    A.S = S; A.T = T; A.len = len; A.R = R;
    A.amp = d_samples; A.window = p->d_window; A.a = d_a; A.b = d_b; A.posMod = d_posmod;
    A.rnd = d_rnd; A.st = d_st; A.gst = d_gst; A.out = d_out; A.err = g_err;
    A.sr = (double)settings().sampleRate;
    A.cycleLength = p->grainLength * settings().sampleRate / overlaps;  // L/maxiGrains.h:346
    A.grainLength = p->grainLength;
    A.sampleDur = (int)p->sampleDur;
    A.mySampleRate = p->mySampleRate;
    size_t lds = sizeof(double) * p->sampleDur;
    A.winInLds = lds <= 60 * 1024;
    if (!A.winInLds) lds = 0;
    dim3 grid((unsigned)((S + 63) / 64));
    if (!tune_get("grain_chunked")) {  // K8: one lane per stream, serial in time
        KernelTimer kt("granular_kernel", st);
        switch (mode) {
            case 0: hipLaunchKernelGGL((granular_kernel<0>), grid, dim3(64), lds, st, A); break;
            case 1: hipLaunchKernelGGL((granular_kernel<1>), grid, dim3(64), lds, st, A); break;
            case 2: hipLaunchKernelGGL((granular_kernel<2>), grid, dim3(64), lds, st, A); break;
            default: hipLaunchKernelGGL((granular_kernel<3>), grid, dim3(64), lds, st, A); break;
        }
    } else {  // K8a scheduler pre-pass + K8b (stream, chunk) render
        // K8c eligibility: maxiTimeStretch, inc exactly 1.0 (the device evaluates the same IEEE division),
        // carried-in grains on the integer grid too, a window index that exists for every read
        bool unit = false;
        // (maxiTimeStretch::play and playAtPosition spawn every grain with speed +-1; maxiStretch / maxiPitchShift do not)
        // `both`: the static conditions for K8c hold and whether the carried-in grains allow it is only known on the device:
        // K8c and K8d are both enqueued, predicated on the check kernel's word (UnitArgs::sel).  Only with the K8d knob off
        // (A/B runs) the word is read back and the host picks, as before.
        bool both = false;
        const bool line_ok = tune_get("grain_line") && T < (size_t)1 << 30 && len >= 128 && len < ((size_t)1 << 31) - 128 &&
                             p->sampleDur < 32768;
        // the prologue kernel also decides K8c's eligibility on the device (g_err[1]) when the static conditions hold
        bool check = false;
        if ((mode == 0 || mode == 2) && tune_get("grain_unit") && T < (size_t)1 << 30 && len >= 128 && len < ((size_t)1 << 31) - 128 &&
            p->sampleDur < 32000) {
            const double frequency = (1.0 / p->grainLength) * 1.0;
            const double inc = (double)A.sampleDur / (A.sr / frequency);
            check = inc == 1.0;
        }
        // scratch (per stream): spawn lists + chunk table + copy of the carried-in grains, sized for 64-sample tiles (the finest
        // chunking) so that it can be taken before the path is known
        const double minCycle = A.cycleLength;  // randomOffset >= 0 only lengthens a cycle
        const size_t G = (size_t)((double)T / (minCycle > 1.0 ? floor(minCycle) : 1.0)) + 2;
        const size_t nd = 2 * G * S + 4 * kSlots * S;                 // doubles: spawn_pos | spawn_inc | gst copy
        const size_t ni = G * S + ((T + 63) / 64 + 1) * S + 3 * S;    // int32: spawn_n | chunk_first | slice carry | progress (streamed form)
        const size_t bytes = nd * sizeof(double) + ni * sizeof(int32_t);
        void *g_sched_scratch = nullptr;
        if (int s = scratch_get(SCR_GRAIN_SCHED, st, bytes, &g_sched_scratch)) return s;
        double *spawn_pos = (double *)g_sched_scratch;
        double *spawn_inc = spawn_pos + G * S;
        double *gst_copy = spawn_inc + G * S;
        int32_t *spawn_n = (int32_t *)(gst_copy + 4 * kSlots * S);
        int32_t *chunk_first = spawn_n + G * S;
        int32_t *prog = chunk_first + ((T + 63) / 64 + 1) * S + 2 * S;
        hipLaunchKernelGGL(granular_prologue_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, S, (const double *)d_gst,
                           gst_copy, (double)len, A.sampleDur, check ? 1 : 0, g_err + 1, prog);
        if (check) {
            if (line_ok && !tune_get("grain_sync")) {
                both = true;
                unit = true;
            } else {
                int bad = 1;
                MXG_HIP(hipMemcpyAsync(&bad, g_err + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                MXG_HIP(hipStreamSynchronize(st));
                unit = bad == 0;
            }
        }
        size_t C = (T + 255) / 256;
        const size_t cmax = ((size_t)tune_get("grain_lanes_k") * 1024 + S - 1) / S;
        if (C > cmax) C = cmax;
        if (C < 1) C = 1;
        size_t Tc = (T + C - 1) / C;
        // K8d (tile render for arbitrary increments) takes what K8c cannot: any mode, any increment
        // (grain ages travel as 16-bit values in its candidate metadata and carried-in grains are only looked for over the first
        // 32 768 samples: a grain of 32 768 samples or more -- 0.4 s at 96 kHz -- takes the (stream, chunk) walk K8b)
        const bool line = !unit && line_ok;
        if (unit || line) Tc = 64;  // K8c / K8d tiles are 64 samples; the chunk table is indexed per tile
        C = (T + Tc - 1) / Tc;
        SchedArgs Q;
        Q.S = S; Q.T = T; Q.len = len; Q.R = R; Q.G = G; Q.Tc = Tc; Q.C = C;
        Q.a = d_a; Q.b = d_b; Q.posMod = d_posmod; Q.rnd = d_rnd; Q.st = d_st;
        Q.spawn_n = spawn_n; Q.spawn_pos = spawn_pos; Q.spawn_inc = spawn_inc; Q.chunk_first = chunk_first;
        Q.err = g_err;
        Q.sr = A.sr; Q.cycleLength = A.cycleLength; Q.grainLength = A.grainLength; Q.sampleDur = A.sampleDur;
        Q.fast = tune_get("grain_fast_sched");
        Q.n_base = 0; Q.carry = nullptr; Q.c_end = C;
        UnitArgs U;
        U.S = S; U.T = T; U.len = len; U.G = G; U.C = C;
        U.amp = d_samples; U.window = p->d_window;
        U.spawn_n = spawn_n; U.spawn_pos = spawn_pos; U.spawn_inc = spawn_inc; U.chunk_first = chunk_first;
        U.gst_in = gst_copy; U.gst_out = d_gst; U.out = d_out; U.err = g_err; U.sampleDur = A.sampleDur;
        U.c0 = 0;
        U.pan = nullptr;
        U.mixpart = nullptr;
        U.sel = both ? g_err + 1 : nullptr;
        U.want = 0;
        U.retry = g_err + 2;
        U.spin_limit = tune_get("grain_spin_limit") > 0 ? tune_get("grain_spin_limit") : kStreamSpin;
        const size_t stiles = (S + 63) / 64;
        if ((unit || line) && d_pan) {
            if (int e = scratch_get(SCR_GRAIN_MIX, st, sizeof(double) * stiles * T * 2, (void **)&U.mixpart)) {
                (void)hipMemsetAsync(g_err, 0, 4 * sizeof(int), st);  // (the prologue has written the eligibility word: zero between calls)
                return e;
            }
            U.pan = d_pan;
        }
        // maxiTimeStretch::play / playAtPosition on the unit path: the scheduler is a serial walk per stream (32 wavefronts for 2048 streams)
        // and the render fills the chip, so the call is cut into time slices and slice i's render (on the library's
        // auxiliary stream) overlaps slice i+1's scheduling.  The scheduler state carries over in d_st / carry; the
        // spawn list and the chunk table are the same arrays a single launch fills, so the bits do not change.
        // Slice i is twice as long as slice i-1 (weights 1, 2, 4, ...): scheduling a slice takes about half as long as
        // rendering it, so slice i+1 is scheduled in the time slice i renders and only the short first one is exposed.
        const size_t nsched = (S + 255) / 256;  // scheduler workgroups of the streamed form (256 lanes = 256 streams each)
        const bool streamed = (unit || line) && tune_get("grain_streamed") && nsched + stiles * C < ((size_t)1 << 31);
        int slices = streamed ? 1 : tune_get("grain_slices");
        while (slices > 1 && C / ((size_t(1) << slices) - 1) < 16) slices--;  // first slice >= 16 tiles (1024 samples)
        const size_t wsum = (size_t(1) << slices) - 1;
        auto slice_start = [&](int i) { return i >= slices ? C : C * ((size_t(1) << i) - 1) / wsum; };
        // consecutive K8d tiles per workgroup: as many as keep (alive at the span's start) + (born in the span) within kCand
        int lt = (int)((double)(kCand - kSlots - 1) * minCycle / 64.0);
        lt = lt < 1 ? 1 : (lt > kLineTiles ? kLineTiles : lt);
        U.ltiles = (unsigned)lt;
        U.cend = (unsigned)C;
        auto launch_line_streamed = [&]() {  // K8d, one launch (any scheduler mode): granular_line_kernel<mode>
            U.want = 1;
            U.c0 = 0;
            U.cend = (unsigned)C;
            KernelTimer kt("granular_line_kernel", st);
            const dim3 g((unsigned)(nsched + stiles * ((C + lt - 1) / lt)));
            switch (mode) {
                case 0: hipLaunchKernelGGL((granular_line_kernel<0>), g, dim3(256), 0, st, U, Q, prog, (unsigned)nsched); break;
                case 1: hipLaunchKernelGGL((granular_line_kernel<1>), g, dim3(256), 0, st, U, Q, prog, (unsigned)nsched); break;
                case 2: hipLaunchKernelGGL((granular_line_kernel<2>), g, dim3(256), 0, st, U, Q, prog, (unsigned)nsched); break;
                default: hipLaunchKernelGGL((granular_line_kernel<3>), g, dim3(256), 0, st, U, Q, prog, (unsigned)nsched); break;
            }
            const unsigned rows = (unsigned)((C + lt - 1) / lt);  // the retry (returns at once unless a render gave up)
            hipLaunchKernelGGL((granular_retry_kernel<true>), dim3((unsigned)stiles, rows < 24u ? rows : 24u), dim3(256), 0, st, U, Q, rows);
        };
        if (streamed && unit) {
            // K8c, one launch: scheduler lanes and tile renders side by side (granular_unit_kernel<SMODE>), on the caller's stream.
            // With both renderers enqueued (UnitArgs::sel) each is a complete streamed launch -- scheduler, renders, state -- and
            // every workgroup of the one that was not picked returns at once.
            if (both) launch_line_streamed();
            U.want = 0;
            U.c0 = 0;
            U.cend = (unsigned)C;
            {
                KernelTimer kt("granular_unit_kernel", st);
                const dim3 g((unsigned)(nsched + stiles * C));
                if (mode == 0) {
                    hipLaunchKernelGGL((granular_unit_kernel<0>), g, dim3(256), 0, st, U, Q, prog, (unsigned)nsched);
                } else {
                    hipLaunchKernelGGL((granular_unit_kernel<2>), g, dim3(256), 0, st, U, Q, prog, (unsigned)nsched);
                }
                const unsigned rows = (unsigned)C;  // the retry (returns at once unless a render gave up)
                hipLaunchKernelGGL((granular_retry_kernel<false>), dim3((unsigned)stiles, rows < 32u ? rows : 32u), dim3(256), 0, st, U, Q, rows);
            }
            if (U.pan) {
                mix_partials_launch(st, stiles, T * 2, U.mixpart, d_mix);
                mixed = true;
            }
        } else if (streamed) {
            launch_line_streamed();
            if (U.pan) {
                mix_partials_launch(st, stiles, T * 2, U.mixpart, d_mix);
                mixed = true;
            }
        } else if ((unit || line) && slices > 1) {  // the tile renders: any mode
            if (int e = aux_stream_init()) {
                (void)hipMemsetAsync(g_err, 0, 4 * sizeof(int), st);
                return e;
            }
            // one caller at a time enqueues its fork/join: a wait captures the event's latest record, so another
            // thread re-recording the shared events between a record and its wait would tie the render to the wrong slice
            std::lock_guard<std::mutex> lock(g_aux_mu);
            Q.carry = chunk_first + (C + 1) * S;
            for (int i = 0; i < slices; i++) {
                const size_t ci = slice_start(i), cn = slice_start(i + 1);
                Q.n_base = (int)(ci * Tc);
                Q.T = (cn * Tc < T ? cn * Tc : T) - ci * Tc;
                Q.c_end = (i == slices - 1) ? C : cn;
                {
                    KernelTimer kt("granular_sched_kernel", st);
                    if (mode == 0) {
                        hipLaunchKernelGGL((granular_sched_kernel<0>), grid, dim3(64), 0, st, Q);
                    } else if (mode == 1) {
                        hipLaunchKernelGGL((granular_sched_kernel<1>), grid, dim3(64), 0, st, Q);
                    } else if (mode == 2) {
                        Q.a = d_a + (size_t)Q.n_base * S;  // playAtPosition reads its position signal [T][S] at the births
                        hipLaunchKernelGGL((granular_sched_kernel<2>), grid, dim3(64), 0, st, Q);
                    } else {
                        hipLaunchKernelGGL((granular_sched_kernel<3>), grid, dim3(64), 0, st, Q);
                    }
                }
                MXG_HIP(hipEventRecord(g_aux_ev[i], st));
                // With both renderers enqueued, the one that returns at once (UnitArgs::sel) still costs its dispatch (4-5 us a
                // slice): it gets a stream of its own instead of a place in the chain of the one that renders.
                hipStream_t su = g_aux, sl = both ? g_aux2 : g_aux;
                U.c0 = (unsigned)ci;
                if (unit) {
                    MXG_HIP(hipStreamWaitEvent(su, g_aux_ev[i], 0));
                    U.want = 0;
                    U.cend = (unsigned)C;
                    KernelTimer kt("granular_unit_kernel", su);
                    hipLaunchKernelGGL((granular_unit_kernel<-1>), dim3((unsigned)((S + 63) / 64), (unsigned)(cn - ci)), dim3(256), 0,
                                       su, U, Q, (int *)nullptr, 0u);
                }
                if (!unit || both) {
                    MXG_HIP(hipStreamWaitEvent(sl, g_aux_ev[i], 0));
                    U.want = 1;
                    U.cend = (unsigned)cn;
                    KernelTimer kt("granular_line_kernel", sl);
                    hipLaunchKernelGGL((granular_line_kernel<-1>), dim3((unsigned)((S + 63) / 64), (unsigned)((cn - ci + lt - 1) / lt)),
                                       dim3(256), 0, sl, U, Q, (int *)nullptr, 0u);
                }
            }
            // the grains alive after the call follow from the scheduler's lists alone (closed forms / the exact multi-step
            // advance), and the renders read the COPY of the carried-in grains: the state kernels run beside the last slice's render
            if (unit) {
                U.want = 0;
                hipLaunchKernelGGL(granular_unit_state_kernel, grid, dim3(64), 0, st, U);
            }
            if (!unit || both) {
                U.want = 1;
                hipLaunchKernelGGL(granular_line_state_kernel, grid, dim3(64), 0, st, U);
            }
            MXG_HIP(hipEventRecord(g_aux_done, g_aux));
            MXG_HIP(hipStreamWaitEvent(st, g_aux_done, 0));
            if (both) {
                MXG_HIP(hipEventRecord(g_aux2_done, g_aux2));
                MXG_HIP(hipStreamWaitEvent(st, g_aux2_done, 0));
            }
            if (U.pan) {
                mix_partials_launch(st, stiles, T * 2, U.mixpart, d_mix);
                mixed = true;
            }
        } else {
        {
        KernelTimer kt("granular_sched_kernel", st);
        switch (mode) {
            case 0: hipLaunchKernelGGL((granular_sched_kernel<0>), grid, dim3(64), 0, st, Q); break;
            case 1: hipLaunchKernelGGL((granular_sched_kernel<1>), grid, dim3(64), 0, st, Q); break;
            case 2: hipLaunchKernelGGL((granular_sched_kernel<2>), grid, dim3(64), 0, st, Q); break;
            default: hipLaunchKernelGGL((granular_sched_kernel<3>), grid, dim3(64), 0, st, Q); break;
        }
        }
        if (unit) {
            U.want = 0;
            {
                KernelTimer kt("granular_unit_kernel", st);
                hipLaunchKernelGGL((granular_unit_kernel<-1>), dim3((unsigned)((S + 63) / 64), (unsigned)C), dim3(256), 0, st, U, Q,
                                   (int *)nullptr, 0u);
            }
            hipLaunchKernelGGL(granular_unit_state_kernel, grid, dim3(64), 0, st, U);
            if (both) {
                U.want = 1;
                {
                    KernelTimer kt("granular_line_kernel", st);
                    hipLaunchKernelGGL((granular_line_kernel<-1>), dim3((unsigned)((S + 63) / 64), (unsigned)((C + lt - 1) / lt)), dim3(256), 0, st, U, Q, (int *)nullptr, 0u);
                }
                hipLaunchKernelGGL(granular_line_state_kernel, grid, dim3(64), 0, st, U);
            }
            if (U.pan) {
                mix_partials_launch(st, stiles, T * 2, U.mixpart, d_mix);
                mixed = true;
            }
        } else if (line) {
            U.want = 1;
            {
                KernelTimer kt("granular_line_kernel", st);
                hipLaunchKernelGGL((granular_line_kernel<-1>), dim3((unsigned)((S + 63) / 64), (unsigned)((C + lt - 1) / lt)), dim3(256), 0, st, U, Q, (int *)nullptr, 0u);
            }
            hipLaunchKernelGGL(granular_line_state_kernel, grid, dim3(64), 0, st, U);
            if (U.pan) {
                mix_partials_launch(st, stiles, T * 2, U.mixpart, d_mix);
                mixed = true;
            }
        } else {
        RenderArgs Rr;
        Rr.S = S; Rr.T = T; Rr.len = len; Rr.G = G; Rr.Tc = Tc; Rr.C = C;
        Rr.amp = d_samples; Rr.window = p->d_window;
        Rr.spawn_n = spawn_n; Rr.spawn_pos = spawn_pos; Rr.spawn_inc = spawn_inc; Rr.chunk_first = chunk_first;
        Rr.gst_in = gst_copy; Rr.gst_out = d_gst; Rr.out = d_out; Rr.err = g_err;
        Rr.sampleDur = A.sampleDur; Rr.winInLds = A.winInLds;
        const size_t lanes = S * C;
        KernelTimer kt("granular_render_kernel", st);
        hipLaunchKernelGGL(granular_render_kernel, dim3((unsigned)((lanes + 63) / 64)), dim3(64), lds, st, Rr);
        }
        }
    }
    MXG_HIP(hipGetLastError());
    if (d_pan && !mixed) {
        if (int e = mxg_mix_stereo(S, T, d_out, d_pan, d_mix, stream)) {
            (void)hipMemsetAsync(g_err, 0, 4 * sizeof(int), st);
            return e;
        }
    }
    if (!tune_get("grain_sync")) {
        // deferred: the render's error word (1..5, below) is forwarded to the library's async error word by a one-lane kernel at
        // the end of the stream's work; the next call of any entry point -- or the caller's next synchronising call -- returns
        // it (mxg_last_async_error, include/maxigpu.h).  Nothing here blocks the host, and the whole sequence can be captured.
        hipLaunchKernelGGL(grain_err_publish_kernel, dim3(1), dim3(1), 0, st, g_err, async_error_word());
        return check_hip(hipGetLastError(), "mxg_granular_render launch");
    }
    int herr = 0;
    MXG_HIP(hipMemcpyAsync(&herr, g_err, sizeof(int), hipMemcpyDeviceToHost, st));
    MXG_HIP(hipMemsetAsync(g_err, 0, 4 * sizeof(int), st));  // (the words are zero between calls)
    MXG_HIP(hipStreamSynchronize(st));
    if (herr) return async_error_status(ASYNC_GRAIN_BASE + herr);
    return MXG_OK;
}

}  // extern "C"
