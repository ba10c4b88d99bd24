// tests/patches/filters2_patch.cpp -- a patch in the reference's plugin form (void setup(); void play(double*);) over the newer
// filter / envelope classes no shipped example uses on their own: maxiSVF, maxiBiquad (three types), maxiDCBlocker, maxiEnvGen
// (ADSR with its HOLD stage, gated by an oscillator).  TEST INFRASTRUCTURE: compiled once against the reference's src/maximilian.h
// (oracle/Makefile _ref/example_p1 -> the golden stream) and once against include/maximilian.h (host/Makefile dropin_p1).
#include "maximilian.h"

maxiOsc src, gate;
maxiSVF svf;
maxiBiquad bq[3];
maxiDCBlocker dc;
maxiEnvGen eg;
int n = 0;

void setup() {
    svf.setCutoff(800);
    svf.setResonance(2.5);
    bq[0].set(maxiBiquad::LOWPASS, 1200, 0.7, 0);
    bq[1].set(maxiBiquad::PEAK, 900, 2.0, 6.0);
    bq[2].set(maxiBiquad::HIGHSHELF, 3000, 0.7, -4.0);
    eg.setupADSR(5, 20, 0.4, 60);
}

void play(double *output) {
    const double x = src.sawn(110);
    const double e = eg.play(gate.square(4));              // +-1 gate: attack / decay / hold while high, release when it drops
    if (n % 1500 == 0) svf.setCutoff(400 + (n / 1500) * 300);  // a parameter change every 1500 samples
    double y = svf.play(x, 0.6, 0.3, 0.0, 0.1);
    y = bq[0].play(y);
    y = bq[1].play(y) + bq[2].play(x) * 0.25;
    output[0] = dc.play(y * e + 0.2, 0.995);
    output[1] = e;
    n++;
}
