// sampler.hip -- maxiSampler (src/libs/maxiSynths.h:137-187, maxiSynths.cpp:262-300) as a bank of samplers.
//
// A maxiSampler is `voices` (<= 32) slots, each a maxiEnv + a maxiSample play head over the same sample;
// play() (maxiSynths.cpp:289-312) is, per slot in index order:
//     envOut = envelope.adsr(envOutGain, envelope.trigger);
//     if (envOut > 0) { outputs[i] = sample.play4(freq, 0, len) * envOut;  output += outputs[i]/voices;
//                       if (trigger == 1 && !sustain) trigger = 0; }
// with freq = pitchRatios[(int)pitch + originalPitch] * ((1./len) * sampleRate) -- block-constant per slot,
// evaluated on the host by mxg_sampler_freq_host (table: src/maximilian.h:112 through maxi_tables.h).
// The control side (trigger(), midiNoteOn/Off: round-robin slot allocation, maxiSynths.cpp:351-391, 484-491)
// changes slot state between play() calls and stays on the host (the facade / Python mirror): the kernel
// renders N samples of every slot between two control events.
// One lane = one slot; a sampler's slots are consecutive lanes of one wavefront -- `voices` live lanes at the
// start of a group of `stride` = the next power of two (any count 1 .. 32, as the reference accepts:
// maxiSynths.cpp:284-289; the surplus lanes of a group shadow a live slot and store nothing) -- so the
// sampler's output is the in-order sum of the active slots' outputs[i]/voices, gathered across lanes with
// __shfl -- the reference's left-to-right order, hence bit-exact.  The arrays stay unpadded: slot i of
// sampler s is element s * voices + i.
#include "mxg_common.h"
#include "mxg_env.h"
#include "mxg_smp.h"
#include "maxi_tables.h"

namespace mxg {
namespace {

struct SamplerArgs {
    size_t V, N;
    int voices, stride, sustain;  // stride: lanes per sampler (the next power of two >= voices)
    const double *amp;
    size_t len;
    const double *freq, *gain, *par;
    const int64_t *holdtime;
    double *position;
    int32_t *trigger;
    double *outhold, *dst;
    int64_t *ist;
    double *mix, *outputs;
    double sr;
};

__global__ void __launch_bounds__(256) sampler_kernel(SamplerArgs A) {
    const size_t V = A.V, N = A.N;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t sampler = t / (size_t)A.stride;
    const int slot = (int)(t % (size_t)A.stride);
    const size_t NS = V / (size_t)A.voices;
    const bool valid = sampler < NS && slot < A.voices;
    const size_t v = sampler * (size_t)A.voices + (size_t)slot;
    const size_t vc = valid ? v : V - 1;  // surplus lanes shadow the last slot (never stored, never summed)
    Env e;
    env_load(e, V, vc, A.par, A.holdtime, A.dst, A.ist);
    Smp s = {A.amp, A.len, A.position[vc], 1.0, 0.0, false, 0.0, 0.0};
    int trigger = A.trigger[vc];
    double outs = A.outhold[vc];
    const double gain = A.gain[vc], frequency = A.freq[vc];
    const double dvoices = (double)A.voices;
    const double end = (double)A.len;  // play4(freq, 0, samples[i].getLength())
    const int lane = threadIdx.x & 63;
    const int g0 = lane - (lane % A.stride);  // first lane of this sampler
    for (size_t n = 0; n < N; n++) {
        const double envOut = env_adsr(e, gain, trigger);
        const bool active = envOut > 0.;
        if (active) {
            SmpReq<7> q;
            smp_gen<7>(s, frequency, 0.0, 0.0, end, A.sr, q);
            double val[4];
#pragma unroll
            for (int l = 0; l < 4; l++) val[l] = A.amp[q.idx[l]];
            outs = smp_eval<7>(q, val) * envOut;
            if (trigger == 1 && !A.sustain) trigger = 0;
        }
        const double contrib = outs / dvoices;
        // output += outputs[i]/voices over the ACTIVE slots in slot order (inactive slots add nothing)
        double output = 0;
        for (int j = 0; j < A.voices; j++) {
            const double cj = __shfl(contrib, g0 + j);
            const int aj = __shfl((int)active, g0 + j);
            if (aj) output += cj;
        }
        if (valid) {
            if (A.outputs) A.outputs[n * V + v] = outs;
            if (lane == g0) A.mix[n * NS + sampler] = output;
        }
    }
    if (valid) {
        env_store(e, V, v, A.dst, A.ist);
        A.position[v] = s.pos;
        A.trigger[v] = trigger;
        A.outhold[v] = outs;
    }
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" {

// pitchRatios[(int)pitch + originalPitch] * ((1./len) * sampleRate), maxiSynths.cpp:297 (originalPitch = 67)
int mxg_sampler_freq_host(size_t V, const double *h_pitch, size_t len, double *h_freq) {
    static const double ratios[256] = MAXI_PITCH_RATIOS_INIT;
    MXG_REQUIRE(h_pitch && h_freq && len > 0, "null pointer / empty sample");
    const size_t sr = settings().sampleRate;
    for (size_t v = 0; v < V; v++) {
        const int idx = (int)h_pitch[v] + 67;
        MXG_REQUIRE(idx >= 0 && idx < 256, "pitch + originalPitch outside pitchRatios[256]");
        h_freq[v] = ratios[idx] * ((1. / len) * sr);
    }
    return MXG_OK;
}

int mxg_sampler_render(size_t V, size_t N, int voices, int sustain, const double *d_samples, size_t len,
                       const double *d_freq, const double *d_gain, const double *d_par, const int64_t *d_holdtime,
                       double *d_position, int32_t *d_trigger, double *d_outhold, double *d_dst, int64_t *d_ist,
                       double *d_mix, double *d_outputs, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(voices >= 1 && voices <= 32, "voices must be 1 .. 32 (maxiSampler holds 32 slots, src/libs/maxiSynths.h:170-175)");
    MXG_REQUIRE(V % (size_t)voices == 0, "V must be a whole number of samplers");
    MXG_REQUIRE(d_samples && d_freq && d_gain && d_par && d_holdtime && d_position && d_trigger && d_outhold &&
                    d_dst && d_ist && d_mix, "null device pointer");
    MXG_REQUIRE(len > 0, "empty sample");
    if (V == 0 || N == 0) return MXG_OK;
    int stride = 1;
    while (stride < voices) stride *= 2;
    const SamplerArgs A = {V, N, voices, stride, sustain, d_samples, len, d_freq, d_gain, d_par, d_holdtime, d_position,
                           d_trigger, d_outhold, d_dst, d_ist, d_mix, d_outputs, (double)settings().sampleRate};
    const int block = 256;
    const size_t lanes = V / (size_t)voices * (size_t)stride;
    hipLaunchKernelGGL(sampler_kernel, dim3((unsigned)((lanes + block - 1) / block)), dim3(block), 0, resolve_stream(stream), A);
    return check_hip(hipGetLastError(), "sampler_kernel launch");
}

}  // extern "C"
