// tests/patches/sampler_zx_patch.cpp -- a patch in the reference's plugin form over the maxiSample members no shipped example
// calls: the trigger-driven players (playOnZX, playOnZXAtSpeed, playOnZXAtSpeedFromOffset, playOnZXAtSpeedBetweenPoints,
// loopSetPosOnZX), playWithPhasor under a maxiOsc::phasor, playAtSpeedBetweenPointsFromPos, normalise, loopRecord (recording an
// oscillator into a sample that is being played), reset and the copy assignment.  TEST INFRASTRUCTURE: compiled once against the
// reference (oracle/Makefile _ref/example_p3 -> the golden stream) and once against include/maximilian.h (host/Makefile dropin_p3).
#include "maximilian.h"

maxiSample s1, s2, s3, s4, s5, s6, s7, s8;
maxiOsc pha, rec;
int n = 0;

void setup() {
    vector<double> data(9000);
    for (int i = 0; i < 9000; i++) data[i] = ((((i * 53) % 601) - 300) / 400.0) * (1.0 - i / 18000.0);
    s1.setSample(data); s2.setSample(data); s3.setSample(data); s4.setSample(data);
    s5.setSample(data); s6.setSample(data); s7.setSample(data);
    vector<double> ints(9000);
    for (int i = 0; i < 9000; i++) ints[i] = ((i * 53) % 601) - 300;  // a "16-bit" sample for normalise / loopRecord
    s6.setSample(ints);
    s8 = s1;        // copy assignment: position 0, the global sample rate
    s7.trigger();
}

void play(double *output) {
    const double trig = (n % 2500 < 1250) ? 1.0 : -1.0;  // a square gate: a zero crossing every 2500 samples
    double w = s1.playOnZX(trig);
    w += s2.playOnZXAtSpeedBetweenPoints(trig, 0.75, 0.2, 0.5);
    w += s3.loopSetPosOnZX(trig, 0.3);
    w += s4.playOnZXAtSpeedFromOffset(trig, 1.5, 0.4);
    w += s8.playOnZXAtSpeed(trig, 0.5);
    w += 0.5 * s5.playWithPhasor(pha.phasor(3.0));
    w += 0.25 * s7.playAtSpeedBetweenPointsFromPos(1.0 + (n / 4000), 1000.0, 6000.0, 1000.0 + (n % 5000));
    if (n == 6000) s6.normalise(200.0);
    if (n == 12000) s6.reset();
    s6.loopRecord(rec.saw(5.0), (n % 6000) >= 3000, 0.5, 0.1, 0.6);
    output[0] = w;
    output[1] = s6.play() / 300.0;
    n++;
}
