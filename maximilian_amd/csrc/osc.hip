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

namespace mxg {

namespace {

// Device-resident copies of the two static tables (with their guard elements).
__device__ const double MAXI_SINE_TAB_D[MAXI_SINE_TAB_LEN] = MAXI_SINE_TAB_INIT;
__device__ const double MAXI_TRANS_TAB_D[MAXI_TRANS_TAB_LEN] = MAXI_TRANS_TAB_INIT;

// Per-voice values that depend only on (frequency, p1, p2): hoisted out of the sample loop
// when the frequency is block-constant.  Each is the exact sub-expression of the reference.
struct OscPre {
    double inc;  // phase increment
    double k;    // sawn: 8820.22/frequency            (C:346)
    double p1;   // pulse: clamped duty (C:304-305); phasorBetween: startphase
    double p2;   // phasorBetween: endphase
};

template <int WF>
__device__ __forceinline__ OscPre osc_pre(double f, double sr, double p1, double p2) {
    OscPre q;
    q.k = 0.0;
    q.p1 = p1;
    q.p2 = p2;
    if constexpr (WF == MXG_OSC_SINEBUF) {
        q.inc = 512. / (sr / (f * kChandiv));  // C:269
    } else if constexpr (WF == MXG_OSC_SINEBUF4) {
        q.inc = 512. / (sr / (f));  // C:241
    } else if constexpr (WF == MXG_OSC_SAW) {
        q.inc = (1. / (sr / (f))) * 2.0;  // C:337
    } else if constexpr (WF == MXG_OSC_SAWN) {
        q.inc = (1. / (sr / (f)));  // C:345
        q.k = (8820.22 / f);        // C:346
    } else if constexpr (WF == MXG_OSC_PHASORBETWEEN) {
        q.inc = ((p2 - p1) / (sr / (f)));  // C:328
    } else if constexpr (WF == MXG_OSC_PULSE) {
        double duty = p1;
        if (duty < 0.) duty = 0;  // C:304
        if (duty > 1.) duty = 1;  // C:305
        q.p1 = duty;
        q.inc = (1. / (sr / (f)));  // C:307
    } else {
        q.inc = (1. / (sr / (f)));  // C:232, 280, 289, 297, 314, 365
    }
    return q;
}

// One sample of one voice.  `phase`/`hold` are the members `phase`/`output` (H:173,176).
// s_sine[i+1] == sineBuffer[i] (i=-1..513), s_trans[i] == transition[i] (i=0..1001).
template <int WF>
__device__ __forceinline__ double osc_tick(double &phase, double &hold, const OscPre &q,
                                           const double *s_sine, const double *s_trans) {
    if constexpr (WF == MXG_OSC_SINEWAVE) {  // C:228-235
        double r = sin_2pi_phase(phase);
        hold = r;
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_COSWAVE) {  // C:276-283
        double r = cos_2pi_phase(phase);
        hold = r;
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_PHASOR) {  // C:285-291
        double r = phase;
        hold = r;
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_SAW) {  // C:333-340
        double r = phase;
        hold = r;
        if (phase >= 1.0) phase -= 2.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_TRIANGLE) {  // C:362-373
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        double r;
        if (phase <= 0.5) {
            r = (phase - 0.25) * 4;
        } else {
            r = ((1.0 - phase) - 0.25) * 4;
        }
        hold = r;
        return r;
    } else if constexpr (WF == MXG_OSC_SQUARE) {  // C:293-300 (output held at phase==0.5)
        if (phase < 0.5) hold = -1;
        if (phase > 0.5) hold = 1;
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        return hold;
    } else if constexpr (WF == MXG_OSC_PULSE) {  // C:302-311 (output held at phase==duty)
        if (phase >= 1.0) phase -= 1.0;
        phase += q.inc;
        if (phase < q.p1) hold = -1.;
        if (phase > q.p1) hold = 1.;
        return hold;
    } else if constexpr (WF == MXG_OSC_IMPULSE) {  // C:312-319 (member `output` untouched)
        if (phase >= 1.0) phase -= 1.0;
        double r = phase < q.inc ? 1.0 : 0.0;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_PHASORBETWEEN) {  // C:321-330
        double r = phase;
        hold = r;
        if (phase < q.p1) phase = q.p1;
        if (phase >= q.p2) phase = q.p1;
        phase += q.inc;
        return r;
    } else if constexpr (WF == MXG_OSC_SINEBUF) {  // C:266-274
        phase += q.inc;
        if (phase >= 511) phase -= 512;
        double remainder = phase - floor(phase);
        int i = (int)phase;  // (long)phase: truncation toward zero; |phase| < 2^31 here
        double r = (1 - remainder) * s_sine[1 + i + 1] + remainder * s_sine[2 + i + 1];
        hold = r;
        return r;
    } else if constexpr (WF == MXG_OSC_SINEBUF4) {  // C:237-264
        phase += q.inc;
        if (phase >= 511) phase -= 512;
        double remainder = phase - floor(phase);
        int i = (int)phase;
        int ia = (phase == 0) ? 512 : i - 1;  // C:245-256; index -1 is the 0.0 guard
        double a = s_sine[ia + 1];
        double b = s_sine[i + 1];
        double c = s_sine[i + 1 + 1];
        double d = s_sine[i + 2 + 1];
        double a1 = 0.5 * (c - a);
        double a2 = a - 2.5 * b + 2. * c - 0.5 * d;
        double a3 = 0.5 * (d - a) + 1.5 * (b - c);
        double r = ((a3 * remainder + a2) * remainder + a1) * remainder + b;
        hold = r;
        return r;
    } else {  // MXG_OSC_SAWN  C:342-359
        if (phase >= 0.5) phase -= 1.0;
        phase += q.inc;
        double temp = q.k * phase;
        if (temp < -0.5) temp = -0.5;
        if (temp > 0.5) temp = 0.5;
        temp *= 1000.0;
        temp += 500.0;
        double remainder = temp - floor(temp);
        int i = (int)temp;
        double r = ((1.0 - remainder) * s_trans[i] + remainder * s_trans[1 + i]) - phase;
        hold = r;
        return r;
    }
}

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
#pragma unroll 4
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
