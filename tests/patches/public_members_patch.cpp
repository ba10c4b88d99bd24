// tests/patches/public_members_patch.cpp -- a patch in the reference's plugin form that touches EVERY public member of the classes on
// the hot path (src/maximilian.h: maxiOsc 169-215, maxiFilter 289-366, maxiSample 602-783 minus the file I/O and the trigger-driven
// players -- tests/test_gpu_wav.py and sampler_zx_patch.cpp have those --, maxiEnv 888-932) and uses the classes the way plain value
// types are used: objects in std::vector, copy construction, copy assignment, state members read and written from user code.
// TEST INFRASTRUCTURE: compiled once against the reference (oracle/Makefile _ref/example_p5 -> the golden stream) and once against
// include/maximilian.h (host/Makefile dropin_p5); the GPU test compares the two streams bit for bit.
#include "maximilian.h"

std::vector<maxiOsc> oscs;       // copies of copies (push_back reallocates)
maxiOsc lfo, car;
maxiFilter f1, f2, f3, f4, f5;
maxiFilter fcopy;
std::vector<maxiEnv> envs(3);    // value-initialised: all state 0, holdtime 1
maxiEnv e1, e2;
maxiSample s1, s2, s3;
int n = 0;
double held = 0;

void setup() {
    for (int i = 0; i < 5; i++) {
        maxiOsc o;
        o.phaseReset(0.1 * i);
        oscs.push_back(o);  // a COPY of o (and, on reallocation, copies of the copies)
    }
    e1.setAttack(50);
    e1.setDecay(200);
    e1.setSustain(0.4);
    e1.setRelease(800);
    e1.holdtime = 300;
    e2.setAttackMS(5);
    e2.decay = 0.9995;
    e2.sustain = 0.3;
    e2.release = 0.9997;
    envs[0].attack = 0.001; envs[0].decay = 0.9999; envs[0].sustain = 0.5; envs[0].release = 0.999;
    f1.setCutoff(0.2);
    f1.setResonance(3);
    vector<double> data(6000);
    for (int i = 0; i < 6000; i++) data[i] = (((i * 37) % 401) - 200) / 250.0;
    s1.setSample(data);
    s2.setSampleAndRate(data, 22050);
    s1.myChannels = 1;
    s1.myBitsPerSample = 16;
    s1.trigger();
    s2.trigger();
}

void play(double *output) {
    // ---- maxiOsc: every waveform; the vector's elements are independent oscillators
    double w = 0.2 * oscs[0].sinewave(220) + 0.2 * oscs[1].coswave(110) + 0.1 * oscs[2].phasor(3) + 0.1 * oscs[3].saw(55) +
               0.1 * oscs[4].triangle(330);
    w += 0.1 * car.square(80) + 0.1 * car.pulse(81, 0.3) + 0.1 * car.impulse(5) + 0.1 * car.sinebuf(440) + 0.1 * car.sinebuf4(441) +
         0.1 * car.sawn(100) + 0.1 * car.phasorBetween(2, 0.25, 0.75);
    if (n == 3000) oscs[1] = oscs[0];  // copy assignment: from here on the two run in step
    if (n == 5000) {
        maxiOsc c(oscs[2]);            // copy construction
        w += c.phasor(3);              // the copy's next sample == what oscs[2] will return
    }
    // ---- maxiFilter: five filters, the public members, copies
    double g = f1.lores(w, 800 + 600 * lfo.sinebuf(0.5), 2.5);
    g += 0.5 * f2.hires(w, 3000, 1.5);
    g += 0.5 * f3.bandpass(w, 1200, 0.3);  // (the reference's recurrence is unstable from resonance ~0.42 up)
    g += 0.5 * f4.lopass(w, f1.getCutoff() * 0.0001);       // the member lores() left (the clamped cutoff)
    g += 0.5 * f5.hipass(w, 0.1 + 0.01 * f1.getResonance());  // setResonance(3): the member, not lores()'s parameter
    if (n == 2000) fcopy = f1;         // copy assignment: state and members
    if (n >= 2000) g += 0.25 * fcopy.lores(w, f1.cutoff, f1.resonance);
    if (n == 7000) {
        f2.setCutoff(1234);
        f2.setResonance(2);
        f2.cutoff += 1;
    }
    // ---- maxiEnv: both adsr forms, ar, every data member
    const int gate = (n % 4000) < 1500;
    e1.trigger = gate;
    double ev = e1.adsr(1.0, e1.trigger);
    ev += e2.adsr(0.5, e2.attack, e2.decay, e2.sustain, e2.release, 100, gate);
    envs[0].setTrigger((n % 3000) < 10);
    ev += envs[0].ar(0.5, 0.01, 0.9995, 200, envs[0].getTrigger());
    if (n == 2500) envs[1] = e1;       // copy assignment in mid-flight
    if (n >= 2500) ev += 0.5 * envs[1].adsr(1.0, gate);
    if (n == 6000) {                   // state members read and written from user code
        held = e1.amplitude + e1.output + e1.holdcount + e1.attackphase + e1.decayphase + e1.sustainphase + e1.holdphase + e1.releasephase +
               e1.input + e1.holdtime;
        e1.amplitude = 0.25;
        e1.holdcount = 0;
        e2.amplitude *= 0.5;
        envs[2] = envs[0];
        envs[2].attackphase = 1;
        envs[2].amplitude = envs[0].amplitude;
    }
    if (n >= 6000) ev += 0.25 * envs[2].ar(1.0, 0.02, 0.999, 50, 0);
    maxiEnv ecopy(e2);                 // copy construction every sample: the copy continues from e2's state
    ev += 0.1 * ecopy.adsr(0.5, e2.attack, e2.decay, e2.sustain, e2.release, 100, gate);
    // ---- maxiSample: the players, the buffer as a public vector, copies
    double sv = s1.play() + s2.playAtSpeed(0.7);
    if (n == 1000) {
        s3 = s1;                       // operator=: position 0, the global rate
        s3.amplitudes[10] = 0.75;      // element write
        s3.amplitudes[11] = s3.amplitudes[10] * 0.5 + s1.amplitudes[11];
    }
    if (n >= 1000 && n < 20000) {  // (index < 2500: inside the buffer also while it holds 3000 samples; none after clear())
        sv += s3.playOnce() + 0.5 * s3.amplitudes[(size_t)(n % 2500)] + 0.001 * (double)s3.amplitudes.size();
    }
    if (n == 8000) {
        maxiSample sc(s2);             // copy construction: the play head travels with it
        sv += sc.playAtSpeed(0.7) + 0.001 * sc.getLength() + sc.mySampleRate * 1e-6 + sc.myChannels + sc.myBitsPerSample;
        vector<double> half(s1.amplitudes.begin(), s1.amplitudes.begin() + 3000);
        s3.amplitudes = half;          // assignment: the buffer only
        s3.setPosition(0.5);
    }
    if (n == 9000) sv += s1.isReady() + (double)s1.getSummary().size() + s1.mySampleRate * 1e-6;
    if (n >= 8000 && n < 16000) sv += s3.playLoop(0.1, 0.9) + s3.playUntil(0.8) + s3.playUntilAtSpeed(0.95, 1.5) + s3.playOnceAtSpeed(0.5);
    if (n == 16000) {
        s3.setSample(const_cast<vector<double> &>(static_cast<const vector<double> &>(s1.amplitudes)));
        s3.trigger();
    }
    if (n >= 16000 && n < 20000) sv += s3.play4(0.5, 100, 5000) + s3.playAtSpeedBetweenPoints(1.5, 200, 4000);
    // the public member zxTrig (H:606), called directly from user code and then driven by a player: the detector's state is shared
    if (n >= 12000 && n < 14000) {
        const double tr = ((n / 250) & 1) ? 1.0 : -1.0;
        if ((n % 500) == 125) sv += 2.0 * s2.zxTrig.onZX(tr) + s2.zxTrig.onChanged(tr * 0.5, 0.1);
        sv += 0.5 * s2.playOnZX(tr);
    }
    if (n == 14000) {
        s1.zxTrig = s2.zxTrig;         // maxiTrigger's implicit copy assignment
        sv += s1.zxTrig.onZX(1.0);
    }
    if (n == 20000) s3.clear();
    if (n >= 20000) sv += (double)s3.amplitudes.size() + s3.isReady();
    output[0] = g + held * (n >= 6000);
    output[1] = ev + 0.1 * sv;
    n++;
}
