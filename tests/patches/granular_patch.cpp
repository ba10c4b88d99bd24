// tests/patches/granular_patch.cpp -- a patch in the reference's plugin form (void setup(); void play(double*);) in the shape
// of the reference's openFrameworks granular example (ts->play(1., speed, 0.05, 4, 0) + mymix.stereo): maxiTimeStretch, maxiPitchShift
// and maxiStretch with three different window functors over one maxiSample, their arguments stepping every few thousand samples,
// and a maxiOsc::noise() between them -- ANOTHER consumer of the process-wide rand() stream the grain schedulers draw their jitter
// from, so the order of the draws is part of what is compared.  TEST INFRASTRUCTURE: compiled once against the reference's
// src/maximilian.h + src/libs/maxiGrains.h (oracle/Makefile _ref/example_p2 -> the golden stream) and once against
// include/maximilian.h + include/maxiGrains.h (host/Makefile dropin_p2).
#include "maximilian.h"
#include "maxiGrains.h"

maxiSample samp;
maxiTimeStretch<hannWinFunctor> *ts;
maxiPitchShift<hammingWinFunctor> *ps;
maxiStretch<triangleWinFunctor> *st;
maxiOsc hiss;
maxiMix mymix;
vector<double> outs(2);
int n = 0;

void setup() {
    vector<double> data(60000);
    for (int i = 0; i < 60000; i++)  // integer arithmetic only: a rough tone with a slow tremolo and some grit
        data[i] = ((((i * 37) % 401) - 200) / 300.0) * (0.5 + ((i / 2000) % 4) * 0.125) + (((i * 7919) % 2001) - 1000) / 20000.0;
    samp.setSample(data);
    ts = new maxiTimeStretch<hannWinFunctor>(&samp);
    ps = new maxiPitchShift<hammingWinFunctor>(&samp);
    st = new maxiStretch<triangleWinFunctor>(&samp);
    ts->setPosition(0.25);
    st->setPosition(0.6);
}

void play(double *output) {
    const double speed = 0.5 + 0.25 * ((n / 3000) % 5);  // 0.5 ... 1.5, a step every 3000 samples
    double w = ts->play(speed, 0.05, 4, 0.0);
    w += 0.5 * ps->play(1.0 + 0.1 * ((n / 5000) % 3), 0.04, 3, 0.0);
    w += 0.3 * st->play(1.5, 0.7 + 0.1 * ((n / 7000) % 2), 0.05, 2, 0.0);
    if (n % 7 == 0) w += 0.01 * hiss.noise();
    mymix.stereo(w, outs, 0.25 + 0.5 * ((n / 10000) % 2));
    output[0] = outs[0];
    output[1] = outs[1];
    n++;
}
