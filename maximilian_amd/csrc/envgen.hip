// envgen.hip -- maxiEnvGen (src/maximilian.h:2268-2547) as a voice bank on gfx950.
//
// One envelope shape per bank -- the stage table maxiEnvGen::setup builds (H:2366-2399,
// setupSegmentTime H:2531-2545), evaluated on the host by mxg_envgen_stages_host -- and one trigger
// signal per voice (or one shared gate).  One lane = one envelope: play() (H:2277-2354) is a three-state
// machine (WAITING / TRIGGERED / HOLDING, the switch falls through) around three maxiTrigger zero-crossing
// detectors (H:569-579); the value is linlin(pow(currentlevel, curve), 0, 1, startlevel, endlevel)
// (H:2302-2304, maxiMap::linlin H:801-805).  Only stages[phase] ever holds a non-zero counter /
// currentlevel (every way out of a stage zeroes them), so that pair is the whole per-stage state.
// Counters, phases, states and detectors are integer/compare work: bit-exact.  pow(currentlevel, curve):
// curve == 1 (setupAR/ASR/ADSR, H:2480-2504) is returned exactly (as the host libm does); other curves use the
// device pow => tolerance.  HBM: 8 B out (+ 8 B trigger in when per voice) per sample.
#include <math.h>

#include "mxg_common.h"
#include "mxg_gate.h"
#include "mxg_envgen.h"

namespace mxg {
namespace {

constexpr int kMaxStages = 32;
constexpr double kHold = -46692.0;  // maxiEnvGen::HOLD H:2271

struct EgStage {
    double startlevel, endlevel, gradient, curve;
    long long length;
    int hold;
};

struct EgArgs {
    size_t V, N;
    const double *trig;
    int tpv, nstages, loop, retrigger;
    const double *stages;  // [nstages][6]
    double *dst;           // [5][V]
    int64_t *ist;          // [7][V]
    double *out;
    int px_store;          // pair-row store flavour of the whole chunks (emit_chunk, mxg_common.h); 0 = 8-byte stores
};

template <bool TPV, bool PX>
// trig / out are separate __restrict__ parameters (not members of A): only then may hipcc read the shared gate
// with scalar loads; as vector loads they sit in the same in-order queue as the output stores and drain it.
__global__ void __launch_bounds__(256) envgen_kernel(EgArgs A, const double *__restrict__ trig_in,
                                                      double *__restrict__ out_ptr) {
    __shared__ double s_tab[kMaxStages * 6];
    for (int i = threadIdx.x; i < A.nstages * 6; i += blockDim.x) s_tab[i] = A.stages[i];
    __syncthreads();
    const size_t V = A.V, N = A.N;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((gid & ~(size_t)63) >= V) return;  // the whole wavefront is past the bank
    // surplus lanes shadow voice V-1 (mxg_gate.h); with pair rows the last PAIR of voices, parity kept (voice_kernel, voice.hip)
    const size_t v = PX ? (gid < V ? gid : V - 2 + (gid & 1)) : live_voice(gid, V);
    constexpr int WAITING = EG_WAITING, HOLDING = EG_HOLDING;
    double envval = A.dst[v], currentlevel = A.dst[V + v];
    double tprev = A.dst[2 * V + v], hprev = A.dst[3 * V + v], rprev = A.dst[4 * V + v];
    long long phase = A.ist[v], counter = A.ist[3 * V + v];
    long long i_state = A.ist[V + v], i_nxc = A.ist[2 * V + v];
    long long i_tf = A.ist[4 * V + v], i_hf = A.ist[5 * V + v], i_rf = A.ist[6 * V + v];
    // consume every prologue load here: a use inside the loop would be a counted wait the compiler has to place
    // conservatively (vmcnt(4..8) on every chunk = draining the output stores, which share the counter)
    asm volatile("" : "+v"(envval), "+v"(currentlevel), "+v"(tprev), "+v"(hprev), "+v"(rprev), "+v"(phase), "+v"(counter));
    asm volatile("" : "+v"(i_state), "+v"(i_nxc), "+v"(i_tf), "+v"(i_hf), "+v"(i_rf));
    int state = (int)i_state;
    bool nxc = i_nxc != 0;
    bool tfirst = i_tf != 0, hfirst = i_hf != 0, rfirst = i_rf != 0;
    const long long S = A.nstages;
    const bool loop = A.loop != 0, retrigger = A.retrigger != 0;
    const double *__restrict__ tp = TPV ? trig_in + v : trig_in;
    double *op = out_ptr + v;
    constexpr int U = 8;
    double tn[U];
    GateGroup<U, double> gcur;  // shared gate: lane j keeps chunk j's triggers (mxg_gate.h)
    int gflags = 0;             // ... and whether the trigger (bit 0) / its negation (bit 1) crosses zero upwards inside chunk j
    const auto positive = [](double t) { return t > 0; };
    if constexpr (TPV) {
#pragma unroll
        for (int i = 0; i < U; i++) {
            const size_t m = (size_t)i < N ? (size_t)i : N - 1;
            tn[i] = tp[m * V];
        }
    } else {
        gate_group_load(gcur, trig_in, N, 0, positive);
        const EgCross xg = envgen_cross<U>(gcur.g);
        gflags = (xg.pos ? 1 : 0) | (xg.neg ? 2 : 0);
        asm volatile("" : "+v"(gcur.cls), "+v"(gflags));
    }
    EgRow row = envgen_row(s_tab, A.nstages, phase);  // the stage row of `phase`, re-read when the stage machine has run
    for (size_t n0 = 0; n0 < N; n0 += U) {
        double tc[U];
        if constexpr (TPV) {
#pragma unroll
            for (int i = 0; i < U; i++) {
                tc[i] = tn[i];
                const size_t m = (n0 + U + i < N) ? n0 + U + i : N - 1;  // clamped prefetch, a chunk ahead of the stores
                tn[i] = tp[m * V];
            }
        }
        // Two steady states in which nothing but a detector's previousValue moves and envval is simply repeated:
        // HOLDING while the trigger stays positive (no negative zero crossing, H:2334-2341) and WAITING while it
        // stays <= 0, or stays positive after a positive sample (no trigger either way, H:2281 / onZX H:569-579).  With a shared gate the test is one readlane plus one ballot per chunk.
        const int cc = (int)((n0 / U) & 63);
        if constexpr (!TPV) {
            if (cc == 0 && n0) {  // one drain per 64 chunks
                gate_group_load(gcur, trig_in, N, n0 / (64 * U), positive);
                const EgCross xg = envgen_cross<U>(gcur.g);
                gflags = (xg.pos ? 1 : 0) | (xg.neg ? 2 : 0);
            }
            const int g = lane_value(gcur.cls, cc);
            int fast = 0;
            // (phase != S: the end-of-envelope test after the switch, H:2349-2355, runs on every sample whatever the
            // state; only a host-uploaded state can sit on it while WAITING/HOLDING, and then the stage machine handles it)
            const bool parked = phase != S;
            if (g > 0 && !retrigger && __all(state == HOLDING && !nxc && parked)) fast = 1;
            else if (g < 0 && __all(state == WAITING && parked)) fast = 2;
            else if (g > 0 && __all(state == WAITING && parked && tprev > 0 && !tfirst)) fast = 2;  // gate still up after the end: no crossing either
            if (fast && n0 + U <= N) {
                const double oo[U] = {envval, envval, envval, envval, envval, envval, envval, envval};
                emit_chunk<PX>(op, V, oo, A.px_store);
                const double last = lane_value(gcur.g[U - 1], cc);
                if (fast == 1) { hprev = -last; hfirst = false; }  // holdDetector.onZX(-trigger), no crossing
                else { tprev = last; tfirst = false; }            // trigDetector.onZX(trigger), no crossing
                continue;
            }
        }
        // The general steady chunk (mxg_envgen.h): every lane inside a ramp, a hold or the wait of its own for these U samples.
        // Computed on a copy; committed if the whole wavefront accepts, else the chunk goes through the stage machine.
        if (n0 + U <= N) {
            double oo[U];
            EgCross x;
            if constexpr (TPV) {
                x = envgen_cross<U>(tc);
            } else {  // shared gate: the chunk's crossings were worked out once per group, by lane cc
                const int f = lane_value(gflags, cc);
                x.first = lane_value(gcur.g[0], cc);
                x.last = lane_value(gcur.g[U - 1], cc);
                x.pos = (f & 1) != 0;
                x.neg = (f & 2) != 0;
            }
            EgState s = {envval, currentlevel, tprev, hprev, rprev, tfirst, hfirst, rfirst, phase, counter, state, nxc};
            const bool ok = envgen_steady_chunk<U>(s, row, retrigger, x, oo);
            if (__all(ok)) {
                envval = s.envval; currentlevel = s.currentlevel; tprev = s.tprev; hprev = s.hprev; rprev = s.rprev;
                tfirst = s.tfirst; hfirst = s.hfirst; rfirst = s.rfirst; counter = s.counter; nxc = s.nxc;
                emit_chunk<PX>(op, V, oo, A.px_store);
                continue;
            }
        }
        double og[U];
        const bool whole = n0 + U <= N;  // (wave-uniform; a ragged last chunk goes out sample by sample)
#pragma unroll
        for (int i = 0; i < U; i++) {
            if (n0 + i >= N) break;
            double trigger;
            if constexpr (TPV) trigger = tc[i];
            else trigger = lane_value(gcur.g[i], cc);
            EgState e = {envval, currentlevel, tprev, hprev, rprev, tfirst, hfirst, rfirst, phase, counter, state, nxc};
            envgen_tick(e, s_tab, S, loop, retrigger, trigger);
            envval = e.envval; currentlevel = e.currentlevel; tprev = e.tprev; hprev = e.hprev; rprev = e.rprev;
            tfirst = e.tfirst; hfirst = e.hfirst; rfirst = e.rfirst; phase = e.phase; counter = e.counter;
            state = e.state; nxc = e.nxc;
            if (PX && whole) {
                og[i] = envval;
            } else {
                *op = envval;
                op += V;
            }
        }
        if constexpr (PX)
            if (whole) emit_chunk<PX>(op, V, og, A.px_store);
        row = envgen_row(s_tab, S, phase);
    }
    const bool in = phase < S;
    A.dst[v] = envval;
    A.dst[V + v] = in ? currentlevel : 0.0;
    A.dst[2 * V + v] = tprev; A.dst[3 * V + v] = hprev; A.dst[4 * V + v] = rprev;
    A.ist[v] = phase;
    A.ist[V + v] = state;
    A.ist[2 * V + v] = nxc ? 1 : 0;
    A.ist[3 * V + v] = in ? counter : 0;
    A.ist[4 * V + v] = tfirst ? 1 : 0; A.ist[5 * V + v] = hfirst ? 1 : 0; A.ist[6 * V + v] = rfirst ? 1 : 0;
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

// maxiEnvGen::setup (H:2366-2399) on the host: levels[nlevels], times[nlevels-1] (ms, or maxiEnvGen::HOLD =
// -46692), curves[nlevels-1] -> h_stages [nlevels-1][6] = startlevel, endlevel, gradient, curve, length, hold.
// Returns the number of stages, or MXG_ERR_INVALID where setup() returns false (a second HOLD stage).
int mxg_envgen_stages_host(size_t nlevels, const double *h_levels, const double *h_times, const double *h_curves,
                           double *h_stages) {
    MXG_REQUIRE(h_levels && h_times && h_curves && h_stages, "null pointer");
    MXG_REQUIRE(nlevels >= 2 && nlevels - 1 <= (size_t)kMaxStages, "levels must hold 2..33 values (one more than times/curves)");
    const size_t sr = settings().sampleRate;
    double accumulatedTime = 0;
    bool containsHold = false;
    for (size_t i = 0; i + 1 < nlevels; i++) {
        const double stageTime = h_times[i];
        double length, gradient, hold;
        if (stageTime == kHold) {  // setupSegmentTime H:2531-2545
            if (containsHold) return fail(MXG_ERR_INVALID, "maxiEnvGen::setup - only one hold section allowed");
            length = 0; hold = 1; gradient = 0; containsHold = true;
        } else {
            const double len = ((stageTime / 1000.0) * sr) + accumulatedTime;
            const size_t l = static_cast<size_t>(floor(len));
            accumulatedTime = len - l;
            gradient = 1.0 / l;
            length = (double)l; hold = 0;
        }
        double *st = h_stages + 6 * i;
        st[0] = h_levels[i]; st[1] = h_levels[i + 1]; st[2] = gradient; st[3] = h_curves[i]; st[4] = length; st[5] = hold;
    }
    return (int)(nlevels - 1);
}

int mxg_envgen_render(size_t V, size_t N, const double *d_trig, int tpv, const double *d_stages, int nstages, int loop,
                      int retrigger, double *d_dst, int64_t *d_ist, double *d_out, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_trig && d_stages && d_dst && d_ist && d_out, "null device pointer");
    MXG_REQUIRE(nstages >= 1 && nstages <= kMaxStages, "nstages out of [1, 32]");
    if (V == 0 || N == 0) return MXG_OK;
    int block = tune_get("voice_block");
    if (block > 256) block = 256;
    // the output stream (knob rw_store, rw_store_choice in mxg_common.h: automatic = pair rows of non-temporal stores from 98 304 voices)
    const EgArgs A = {V, N, d_trig, tpv, nstages, loop, retrigger, d_stages, d_dst, d_ist, d_out, rw_store_choice(V, N, d_out, RW_ENVGEN)};
    const dim3 grid((unsigned)((V + block - 1) / block));
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("envgen_kernel", st);
    if (tpv) {
        if (A.px_store) hipLaunchKernelGGL((envgen_kernel<true, true>), grid, dim3(block), 0, st, A, d_trig, d_out);
        else hipLaunchKernelGGL((envgen_kernel<true, false>), grid, dim3(block), 0, st, A, d_trig, d_out);
    } else {
        if (A.px_store) hipLaunchKernelGGL((envgen_kernel<false, true>), grid, dim3(block), 0, st, A, d_trig, d_out);
        else hipLaunchKernelGGL((envgen_kernel<false, false>), grid, dim3(block), 0, st, A, d_trig, d_out);
    }
    return check_hip(hipGetLastError(), "envgen_kernel launch");
}

}  // extern "C"
