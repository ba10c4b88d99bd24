// tests/patches/convolve_sampler_patch.cpp -- a patch in the reference's plugin form over maxiConvolve (setup from a WAV file,
// play() once per sample) and maxiSampler (eight slots, midiNoteOn / trigger / midiNoteOff / setPitch between play() calls).
// The working directory holds `mono.wav`.  TEST INFRASTRUCTURE: compiled once against the reference (oracle/Makefile
// _ref/example_p4 -> the golden stream) and once against the drop-in headers (host/Makefile dropin_p4).
#include "maximilian.h"
#include "maxiConvolve.h"
#include "maxiSynths.h"

maxiConvolve conv;
maxiSampler sampler;
maxiOsc src;
int n = 0;

void setup() {
    conv.setup("mono.wav", 256, 64);
    sampler.load("mono.wav");
    sampler.setNumVoices(8);
    sampler.setRelease(200);
}

void play(double *output) {
    if (n % 3000 == 100) {
        sampler.midiNoteOn(-3 + 2 * ((n / 3000) % 6), 90 + (n / 3000) % 30);
        sampler.trigger();
    }
    if (n % 3000 == 1700) sampler.midiNoteOff(-3 + 2 * ((n / 3000) % 6), 0);
    if (n == 20000) sampler.setPitch(5, true);
    output[0] = conv.play((float)src.saw(220));
    output[1] = sampler.play();
    n++;
}
