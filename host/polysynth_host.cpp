// host/polysynth_host.cpp -- a headless Maximilian host driving the GPU banks through the
// reference's own plugin shape: user code is `void setup()` + `void play(double *output)`
// (src/maximilian.cpp:205-207), and the host calls play() once per frame exactly like the
// RtAudio callback `routing()` of cpp/commandline/player.cpp:25-44 -- restated here without
// RtAudio (no ALSA in this image).  The patch is the per-voice body of
// cpp/commandline/maximilian_examples/15.polysynth/main.cpp:54-70 (oscillator -> lores filter ->
// ADSR, summed over voices, panned with maxiMix::stereo), for V voices instead of 6.
//
//   polysynth_host <voices> <frames> <out.f64> [gpumix]    writes frames*channels interleaved doubles
//
// With `gpumix` the voice sum is not formed in play() at all: the bank renders the block WITH the maxiMix::stereo mixdown fused into the
// same kernel (mxg_voice_render_mix, round 6) and play() reads the frame's two mix values -- the tree-ordered sum of the same
// per-voice products, within conftest.mix_tol of the voice-order sum below.
//
// The banks render 512-frame blocks on the GPU; play() keeps its per-sample shape and sums the
// voices on the host in voice order, so the output is bit-identical to the reference CPU loop
// (checked by tests/test_gpu_host.py against the oracle).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "maximilian_bank.hpp"

static size_t g_voices = 64;
static maxiVoiceBank *voices = nullptr;
static std::vector<double> panL, panR;
static size_t g_frame = 0;
static bool g_gpumix = false;
static const size_t kBlock = 512;

void setup() {
    maxiSettings::setup(44100, 2, 1024);
    voices = new maxiVoiceBank(g_voices, kBlock);
    std::vector<double> freq(g_voices), cutoff(g_voices), res(g_voices);
    panL.resize(g_voices);
    panR.resize(g_voices);
    for (size_t v = 0; v < g_voices; v++) {
        freq[v] = std::fmin(20.0 + (double)(v * 97 % 16384) * 0.30517578125, 5000.0);
        cutoff[v] = 200 + 4 * freq[v];
        res[v] = 1.0 + (double)(v % 16);
        double x = g_voices > 1 ? (double)v / (double)(g_voices - 1) : 0.5;  // maxiMix::stereo, C:503-509
        panL[v] = std::sqrt(1.0 - x);
        panR[v] = std::sqrt(x);
    }
    voices->setVoices(freq, cutoff, res);
    if (g_gpumix) {
        std::vector<double> pan(g_voices);
        for (size_t v = 0; v < g_voices; v++) pan[v] = g_voices > 1 ? (double)v / (double)(g_voices - 1) : 0.5;
        voices->setPan(pan);
    }
    voices->env.setAttack(10);
    voices->env.setDecay(100);
    voices->env.setSustain(0.5);
    voices->env.setRelease(500);
}

void play(double *output) {
    if (g_frame % kBlock == 0) {  // gate for the block about to be rendered
        std::vector<int32_t> gate(kBlock);
        for (size_t i = 0; i < kBlock; i++) gate[i] = ((g_frame + i) % 4096) < 2048 ? 1 : 0;
        voices->setGate(gate);
    }
    if (g_gpumix) {  // the mixdown came with the block
        output[0] = voices->mixFrame(0);
        output[1] = voices->mixFrame(1);
        voices->tick();
        g_frame++;
        return;
    }
    double l = 0, r = 0;
    for (size_t v = 0; v < g_voices; v++) {  // 15.polysynth/main.cpp:54-68
        double s = voices->frame(v);
        l += s * panL[v];
        r += s * panR[v];
    }
    voices->tick();
    output[0] = l;
    output[1] = r;
    g_frame++;
}

// cpp/commandline/player.cpp:25-44, restated
static int routing(double *buffer, unsigned int nBufferFrames, double *lastValues) {
    for (size_t i = 0; i < nBufferFrames; i++) {
        play(lastValues);
        for (size_t j = 0; j < maxiSettings::channels; j++) *buffer++ = lastValues[j];
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <voices> <frames> <out.f64>\n", argv[0]);
        return 2;
    }
    g_voices = (size_t)std::atol(argv[1]);
    const size_t frames = (size_t)std::atol(argv[2]);
    g_gpumix = argc > 4 && std::string(argv[4]) == "gpumix";
    try {
        setup();
        std::vector<double> out(frames * maxiSettings::channels);
        std::vector<double> lastValues(maxiSettings::channels, 0.0);
        const size_t bufferFrames = maxiSettings::bufferSize;
        for (size_t done = 0; done < frames; done += bufferFrames) {
            unsigned int n = (unsigned int)(frames - done < bufferFrames ? frames - done : bufferFrames);
            routing(out.data() + done * maxiSettings::channels, n, lastValues.data());
        }
        FILE *f = std::fopen(argv[3], "wb");
        if (!f) return 3;
        std::fwrite(out.data(), sizeof(double), out.size(), f);
        std::fclose(f);
        std::printf("rendered %zu frames x %zu channels, %zu voices\n", frames, (size_t)maxiSettings::channels, g_voices);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "polysynth_host: %s\n", e.what());
        return 1;
    }
    return 0;
}
