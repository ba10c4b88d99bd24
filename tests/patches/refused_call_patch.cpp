// tests/patches/refused_call_patch.cpp -- a patch in the reference's plugin form whose granular call is one the C-ABI REFUSES
// (maxiTimeStretch::play with ten overlaps of 0.05 s: ten grains alive at once, the device renderer holds eight; then overlaps = 0),
// next to an oscillator that must not notice.  The reference plays such calls with no effect on any other object
// (src/libs/maxiGrains.h:341-355); through include/maximilian.h the refused call prints ONE line and returns silence, and
// maxiOsc::sinewave in the left channel stays the stream of cpp/commandline/main.cpp (golden ex01).  TEST INFRASTRUCTURE
// (host/Makefile dropin_p6, tests/test_gpu_dropin.py).
#include "maximilian.h"
#include "maxiGrains.h"

maxiSample samp;
maxiTimeStretch<hannWinFunctor> *ts;
maxiOsc mySine;
int n = 0;

void setup() {
    vector<double> data(60000);
    for (int i = 0; i < 60000; i++) data[i] = (((i * 37) % 401) - 200) / 300.0;
    samp.setSample(data);
    ts = new maxiTimeStretch<hannWinFunctor>(&samp);
    ts->setPosition(0.25);
}

void play(double *output) {
    output[0] = mySine.sinewave(440);
    output[1] = n < 6000 ? ts->play(1.0, 0.05, 10, 0.0) : ts->play(1.0, 0.05, 0, 0.0);
    n++;
}
