// osc.hip -- maxiOsc voice bank on gfx950.
//
// Path: maxiOsc::{sinewave,coswave,phasor,saw,triangle,square,pulse,impulse,sinebuf,
// sinebuf4,sawn,phasorBetween} (reference src/maximilian.cpp:228-373, cited per function
// as C:line).  The reference advances ONE oscillator by ONE sample per call; here one
// wavefront lane owns one voice (or two adjacent voices) and walks the whole block of N
// samples with its phase in VGPRs; the 514-point sine table / 1001-point transition table
// are staged once per workgroup into LDS.  Every sample is stored straight to
// out[n*V + v]: a wavefront store covers 512 B (VPL=1) or 1 KiB (VPL=2) of one row, so
// the only mandatory HBM traffic is the 8 B/sample output stream (kernel K1).
//
// Numerics: the expression trees are the reference's, op for op, in fp64, compiled with
// -ffp-contract=off.  +,-,*,/ and floor are IEEE-exact on gfx950, so every waveform except
// sinewave/coswave is bit-identical to the reference; those two go through mxg_sincos.h.
#include "mxg_common.h"
#include "maxi_tables.h"
#include "mxg_sincos.h"
#include "mxg_osc.h"

namespace mxg {

namespace {

// Device-resident copies of the two static tables (with their guard elements).
__device__ const double MAXI_SINE_TAB_D[MAXI_SINE_TAB_LEN] = MAXI_SINE_TAB_INIT;
__device__ const double MAXI_TRANS_TAB_D[MAXI_TRANS_TAB_LEN] = MAXI_TRANS_TAB_INIT;

template <int WF>
constexpr bool uses_sine() {
    return WF == MXG_OSC_SINEBUF || WF == MXG_OSC_SINEBUF4;
}

// K1: one lane = VPL adjacent voices; the N-sample recurrence runs in registers.
template <int WF, bool FPS, int VPL, bool NT>
__global__ void osc_kernel(size_t V, size_t N, const double *__restrict__ freq,
                           const double *__restrict__ p1, const double *__restrict__ p2,
                           double *__restrict__ phase_io, double *__restrict__ hold_io,
                           double *__restrict__ out, double sr) {
    // All LDS in ONE array (a second __shared__ object perturbs hipcc's waitcnt placement).
    __shared__ double s_tab[uses_sine<WF>() ? MAXI_SINE_TAB_LEN
                                            : (WF == MXG_OSC_SAWN ? MAXI_TRANS_TAB_LEN : 1)];
    if constexpr (uses_sine<WF>()) {
        for (int i = threadIdx.x; i < MAXI_SINE_TAB_LEN; i += blockDim.x) s_tab[i] = MAXI_SINE_TAB_D[i];
        __syncthreads();
    } else if constexpr (WF == MXG_OSC_SAWN) {
        for (int i = threadIdx.x; i < MAXI_TRANS_TAB_LEN; i += blockDim.x) s_tab[i] = MAXI_TRANS_TAB_D[i];
        __syncthreads();
    }
    const size_t v0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VPL;
    if (v0 >= V) return;

    double ph[VPL], hd[VPL];
    OscPre q[VPL];
#pragma unroll
    for (int j = 0; j < VPL; j++) {
        ph[j] = phase_io[v0 + j];
        hd[j] = hold_io[v0 + j];
        double a = p1 ? p1[v0 + j] : 0.0, b = p2 ? p2[v0 + j] : 0.0;
        if constexpr (!FPS) {
            q[j] = osc_pre<WF>(freq[v0 + j], sr, a, b);
        } else {
            q[j].p1 = a;
            q[j].p2 = b;
        }
    }
    double *o = out + v0;
    const double *fp = freq + v0;
#ifndef MXG_OSC_UNROLL
#define MXG_OSC_UNROLL 4
#endif
#pragma unroll MXG_OSC_UNROLL
    for (size_t n = 0; n < N; n++) {
        double r[VPL];
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            if constexpr (FPS) q[j] = osc_pre<WF>(fp[j], sr, q[j].p1, q[j].p2);
            r[j] = osc_tick<WF>(ph[j], hd[j], q[j], s_tab, s_tab);
        }
        if constexpr (VPL == 2)
            store2<NT>(o, r[0], r[1]);
        else
            store1<NT>(o, r[0]);
        o += V;
        if constexpr (FPS) fp += V;
    }
#pragma unroll
    for (int j = 0; j < VPL; j++) {
        phase_io[v0 + j] = ph[j];
        hold_io[v0 + j] = hd[j];
    }
}

// ---- K1m: K1 + fused maxiMix::stereo partial sums ---------------------------------------------------
// Same per-voice recurrence and (optional) per-voice store as K1; in addition every wavefront
// reduces its 64 voices' panned samples (in*sqrt(1-x), in*sqrt(x), C:503-509) to one (L,R) pair
// per sample, so the mixdown never re-reads the 268 MB block from HBM.  Every 16 samples the 32
// (sample, channel) rows of products are transposed through a padded LDS tile [32][65]: lane
// (c = lane&31, h = lane>>5) sums row c over lanes 32h..32h+31 in lane order, the two halves are
// combined (low half first) and lane c < 32 writes partial[n][wave][channel].  Fixed order =>
// deterministic.  A second tiny kernel (mix_partials_kernel) sums the per-wave partials.
constexpr int kMixChunk = 16;
constexpr int kMixRow = 65;  // 64 lanes + 1 pad: row stride 130 dwords => conflict-free b64 column reads

template <int WF, bool STORE>
__global__ __launch_bounds__(256) void osc_mix_kernel(size_t V, size_t N, const double *__restrict__ freq,
                                                      const double *__restrict__ p1, const double *__restrict__ p2,
                                                      double *__restrict__ phase_io, double *__restrict__ hold_io,
                                                      double *__restrict__ out, const double *__restrict__ gains,
                                                      double *__restrict__ partial, size_t nwaves, double sr) {
    constexpr int kTab = uses_sine<WF>() ? MAXI_SINE_TAB_LEN : (WF == MXG_OSC_SAWN ? MAXI_TRANS_TAB_LEN : 1);
    __shared__ double s_all[kTab + 4 * 2 * (2 * kMixChunk * kMixRow)];
    double *s_tab = s_all;
    if constexpr (uses_sine<WF>()) {
        for (int i = threadIdx.x; i < MAXI_SINE_TAB_LEN; i += blockDim.x) s_tab[i] = MAXI_SINE_TAB_D[i];
    } else if constexpr (WF == MXG_OSC_SAWN) {
        for (int i = threadIdx.x; i < MAXI_TRANS_TAB_LEN; i += blockDim.x) s_tab[i] = MAXI_TRANS_TAB_D[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *tile = s_all + kTab + wave * 2 * (2 * kMixChunk * kMixRow);
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t gwave = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool live = v < V;
    double ph = 0, hd = 0, gl = 0, gr = 0;
    OscPre q = {0, 0, 0, 0};
    if (live) {
        ph = phase_io[v];
        hd = hold_io[v];
        gl = gains[v];
        gr = gains[V + v];
        q = osc_pre<WF>(freq[v], sr, p1 ? p1[v] : 0.0, p2 ? p2[v] : 0.0);
    }
    // Consume the prologue loads HERE.  Otherwise hipcc's waitcnt pass keeps them "pending" at the
    // loop back-edge and puts s_waitcnt vmcnt(0) in every chunk's preheader, which also drains the
    // asynchronous output stores each 16 samples (measured: 74 us -> see profiles/).
    asm volatile("" : "+v"(ph), "+v"(hd), "+v"(gl), "+v"(gr));
    asm volatile("" : "+v"(q.inc), "+v"(q.k), "+v"(q.p1), "+v"(q.p2));
    double *o = out + v;
    const int c = lane & 31, h = lane >> 5;
    // Software-pipelined over 16-sample chunks with two LDS tiles: iteration k renders chunk k into
    // tile k&1 while the column sums of chunk k-1 (other tile) are formed -- independent work the
    // compiler can interleave, so the store stream does not stall behind the LDS round trip.
    constexpr int kTile = 2 * kMixChunk * kMixRow;
    const size_t nch = (N + kMixChunk - 1) / kMixChunk;
    for (size_t k = 0; k <= nch; k++) {
        double *tw = tile + (k & 1) * kTile;
        const double *tr = tile + ((k + 1) & 1) * kTile;
        if (k < nch) {
            const size_t n0 = k * kMixChunk;
            const int cnt = (int)((N - n0) < (size_t)kMixChunk ? (N - n0) : (size_t)kMixChunk);
#pragma unroll 4
            for (int i = 0; i < cnt; i++) {
                double r = 0.0;
                if (live) {
                    r = osc_tick<WF>(ph, hd, q, s_tab, s_tab);
                    if constexpr (STORE) {
                        *o = r;
                        o += V;
                    }
                }
                tw[(2 * i) * kMixRow + lane] = r * gl;      // two[0] = input*sqrt(1.0-x)   C:506
                tw[(2 * i + 1) * kMixRow + lane] = r * gr;  // two[1] = input*sqrt(x)       C:507
            }
        }
        if (k > 0) {
            const size_t n0 = (k - 1) * kMixChunk;
            const int cnt = (int)((N - n0) < (size_t)kMixChunk ? (N - n0) : (size_t)kMixChunk);
            double s = 0.0;
            if (c < 2 * cnt) {
                // four interleaved partial sums keep the fp64 add chains short; fixed order
                const double *row = tr + c * kMixRow + h * 32;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int j = 0; j < 32; j += 4) {
                    s0 += row[j];
                    s1 += row[j + 1];
                    s2 += row[j + 2];
                    s3 += row[j + 3];
                }
                s = (s0 + s1) + (s2 + s3);
            }
            const double other = __shfl_xor(s, 32);
            if (h == 0 && c < 2 * cnt)
                partial[((n0 + (size_t)(c >> 1)) * nwaves + gwave) * 2 + (c & 1)] = s + other;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (live) {
        phase_io[v] = ph;
        hold_io[v] = hd;
    }
}

// mix[n][ch] = sum over waves of partial[n][w][ch]: 256 strided partial sums then a binary tree.
__global__ __launch_bounds__(256) void mix_partials_kernel(size_t nwaves, const double *__restrict__ partial,
                                                           double *__restrict__ mix) {
    __shared__ double s_red[2 * 256];
    const size_t n = blockIdx.x;
    const double *row = partial + n * nwaves * 2;
    double l = 0.0, r = 0.0;
    for (size_t w = threadIdx.x; w < nwaves; w += 256) {
        l += row[2 * w];
        r += row[2 * w + 1];
    }
    s_red[threadIdx.x] = l;
    s_red[256 + threadIdx.x] = r;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_red[threadIdx.x] += s_red[threadIdx.x + s];
            s_red[256 + threadIdx.x] += s_red[256 + threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mix[2 * n] = s_red[0];
        mix[2 * n + 1] = s_red[256];
    }
}

__global__ void osc_pan_gains_kernel(size_t V, const double *__restrict__ pan, double *__restrict__ gains) {
    size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double x = pan[v];
    if (x > 1) x = 1;  // C:504
    if (x < 0) x = 0;  // C:505
    gains[v] = sqrt(1.0 - x);
    gains[V + v] = sqrt(x);
}

typedef void (*osc_mix_fn)(size_t, size_t, const double *, const double *, const double *, double *, double *,
                           double *, const double *, double *, size_t, double);
template <int WF>
osc_mix_fn pick_mix(bool store) {
    return store ? osc_mix_kernel<WF, true> : osc_mix_kernel<WF, false>;
}
osc_mix_fn pick_mix_wf(int wf, bool store) {
    switch (wf) {
        case 0: return pick_mix<0>(store);
        case 1: return pick_mix<1>(store);
        case 2: return pick_mix<2>(store);
        case 3: return pick_mix<3>(store);
        case 4: return pick_mix<4>(store);
        case 5: return pick_mix<5>(store);
        case 6: return pick_mix<6>(store);
        case 7: return pick_mix<7>(store);
        case 8: return pick_mix<8>(store);
        case 9: return pick_mix<9>(store);
        case 10: return pick_mix<10>(store);
        case 11: return pick_mix<11>(store);
    }
    return nullptr;
}

typedef void (*osc_fn)(size_t, size_t, const double *, const double *, const double *, double *,
                       double *, double *, double);

template <int WF>
osc_fn pick(bool fps, int vpl, bool nt) {
    if (fps) return osc_kernel<WF, true, 1, false>;
    if (vpl == 2) return nt ? osc_kernel<WF, false, 2, true> : osc_kernel<WF, false, 2, false>;
    return nt ? osc_kernel<WF, false, 1, true> : osc_kernel<WF, false, 1, false>;
}

osc_fn pick_wf(int wf, bool fps, int vpl, bool nt) {
    switch (wf) {
        case 0: return pick<0>(fps, vpl, nt);
        case 1: return pick<1>(fps, vpl, nt);
        case 2: return pick<2>(fps, vpl, nt);
        case 3: return pick<3>(fps, vpl, nt);
        case 4: return pick<4>(fps, vpl, nt);
        case 5: return pick<5>(fps, vpl, nt);
        case 6: return pick<6>(fps, vpl, nt);
        case 7: return pick<7>(fps, vpl, nt);
        case 8: return pick<8>(fps, vpl, nt);
        case 9: return pick<9>(fps, vpl, nt);
        case 10: return pick<10>(fps, vpl, nt);
        case 11: return pick<11>(fps, vpl, nt);
    }
    return nullptr;
}

}  // namespace

}  // namespace mxg

extern "C" int mxg_osc_render(int waveform, size_t V, size_t N, const double *d_freq, int fps,
                              const double *d_p1, const double *d_p2, double *d_phase,
                              double *d_outhold, double *d_out, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(waveform >= 0 && waveform <= 11, "unknown waveform");
    MXG_REQUIRE(d_freq && d_phase && d_outhold && d_out, "null device pointer");
    MXG_REQUIRE(waveform != MXG_OSC_PULSE || d_p1, "pulse needs d_p1 (duty)");
    MXG_REQUIRE(waveform != MXG_OSC_PHASORBETWEEN || (d_p1 && d_p2),
                "phasorBetween needs d_p1/d_p2 (start/end phase)");
    if (V == 0 || N == 0) return MXG_OK;
    int vpl = tune_get("osc_vpl");
    if (fps || (V & 1) || (((uintptr_t)d_out) & 15)) vpl = 1;
    int block = tune_get("osc_block");
    bool nt = tune_get("osc_nt") != 0;
    osc_fn fn = pick_wf(waveform, fps != 0, vpl, nt);
    size_t lanes = (V + vpl - 1) / vpl;
    dim3 grid((unsigned)((lanes + block - 1) / block)), blk((unsigned)block);
    hipLaunchKernelGGL(fn, grid, blk, 0, resolve_stream(stream), V, N, d_freq, d_p1, d_p2, d_phase,
                       d_outhold, d_out, (double)settings().sampleRate);
    return check_hip(hipGetLastError(), "osc_kernel launch");
}

extern "C" int mxg_osc_render_mix(int waveform, size_t V, size_t N, const double *d_freq, const double *d_p1,
                                  const double *d_p2, double *d_phase, double *d_outhold, double *d_out,
                                  const double *d_pan, double *d_mix, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(waveform >= 0 && waveform <= 11, "unknown waveform");
    MXG_REQUIRE(d_freq && d_phase && d_outhold && d_pan && d_mix, "null device pointer");
    MXG_REQUIRE(waveform != MXG_OSC_PULSE || d_p1, "pulse needs d_p1 (duty)");
    MXG_REQUIRE(waveform != MXG_OSC_PHASORBETWEEN || (d_p1 && d_p2), "phasorBetween needs d_p1/d_p2");
    if (N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const int block = 256;
    const size_t nblocks = (V + block - 1) / block;
    const size_t nwaves = nblocks * (block / 64);
    const size_t need = 2 * V + N * nwaves * 2 + 2;
    double *g_mix_scratch = nullptr;  // per-stream: [2][V] gains | [N][nwaves][2] partials
    if (int s = scratch_get(SCR_OSC_MIX, st, sizeof(double) * need, (void **)&g_mix_scratch)) return s;
    double *gains = g_mix_scratch, *partial = g_mix_scratch + 2 * V;
    if (V) {
        hipLaunchKernelGGL(osc_pan_gains_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, d_pan, gains);
        osc_mix_fn fn = pick_mix_wf(waveform, d_out != nullptr);
        hipLaunchKernelGGL(fn, dim3((unsigned)nblocks), dim3(block), 0, st, V, N, d_freq, d_p1, d_p2, d_phase,
                           d_outhold, d_out, gains, partial, nwaves, (double)settings().sampleRate);
    }
    hipLaunchKernelGGL(mix_partials_kernel, dim3((unsigned)N), dim3(256), 0, st, nwaves, partial, d_mix);
    return check_hip(hipGetLastError(), "osc_mix_kernel launch");
}

// ---- maxiOsc::noise (C:214-220) -----------------------------------------------------------------
//     float r = rand()/(float)RAND_MAX;  output = r*2-1;
// rand() is one process-wide serial stream: the caller supplies the draws (see maxigpu.h), the
// kernel does the reference's float arithmetic.  (float)RAND_MAX = 2^31 exactly; the int -> float
// conversion rounds to nearest even as cvtsi2ss does; r*2-1 is evaluated in float (int operands
// convert to float), then widened to the double member.  Pure streaming: 4 B in, 8 B out.
namespace mxg {
namespace {
__global__ void osc_noise_kernel(size_t count, size_t V, size_t N, const int32_t *__restrict__ rnd,
                                 double *__restrict__ outhold, double *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float r = (float)rnd[i] / 2147483648.0f;
        const double o = (double)(r * 2.0f - 1.0f);
        out[i] = o;
        if (outhold && i >= count - V) outhold[i - (count - V)] = o;
    }
}
}  // namespace
}  // namespace mxg

extern "C" int mxg_osc_noise(size_t V, size_t N, const int32_t *d_rand, double *d_outhold, double *d_out,
                             void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_rand && d_out, "null device pointer");
    if (N == 0 || V == 0) return MXG_OK;
    const size_t count = V * N;
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(osc_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, resolve_stream(stream), count, V,
                       N, d_rand, d_outhold, d_out);
    return check_hip(hipGetLastError(), "osc_noise_kernel launch");
}
