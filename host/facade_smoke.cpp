// host/facade_smoke.cpp -- drives the newer facade classes of include/maximilian_bank.hpp from plain C++ (no HIP
// headers): maxiSVFBank, maxiBiquadBank, maxiDCBlockerBank, maxiEnvGenBank, maxiSampleBank::load/playOnZX,
// maxiMixBank::quad, maxiFFTBatch + features, maxiIFFTBatch, maxiPitchShiftBank.  Writes every result block as raw
// doubles/floats to <outdir>/<name>.bin; tests/test_gpu_host.py compares them with the Python mirror of the same calls
// (same C-ABI underneath, so the bytes must be identical).
//
//   facade_smoke <wav file> <outdir>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "maximilian_bank.hpp"

using maxigpu::DeviceArray;

template <typename T>
static void dump(const std::string &dir, const char *name, const DeviceArray<T> &a) {
    std::vector<T> h = a.download();
    FILE *f = fopen((dir + "/" + name + ".bin").c_str(), "wb");
    if (!f) { perror(name); exit(2); }
    fwrite(h.data(), sizeof(T), h.size(), f);
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s <wav> <outdir>\n", argv[0]); return 1; }
    const std::string wav = argv[1], dir = argv[2];
    try {
        maxiSettings::setup(44100, 2, 512);
        const size_t V = 96, N = 300;
        std::vector<double> x(N * V), cutoff(V), q(V), gain(V), R(V), px(V), py(V);
        for (size_t i = 0; i < N * V; i++) x[i] = ((double)((i * 37) % 1000) / 1000.0 - 0.5) * 1.6;  // exact IEEE ops only: the test rebuilds the same bits in numpy
        for (size_t v = 0; v < V; v++) {
            cutoff[v] = 100.0 + 37.0 * v; q[v] = 0.5 + 0.05 * v; gain[v] = -12.0 + 0.25 * v; R[v] = 0.99 + 0.0001 * v;
            px[v] = (double)v / (V - 1); py[v] = 1.0 - px[v] * 0.5;
        }
        DeviceArray<double> dx(N * V), out(N * V);
        dx.upload(x);

        maxiSVFBank svf(V);
        svf.setCutoff(cutoff); svf.setResonance(q); svf.setMix(0.5, 0.25, 0.125, 1.0);
        svf.play(N, dx.get(), out.get());
        dump(dir, "svf", out);

        maxiBiquadBank bq(V);
        bq.set(maxiBiquadBank::PEAK, cutoff, q, gain);
        bq.play(N, dx.get(), out.get());
        dump(dir, "biquad", out);

        maxiDCBlockerBank dc(V);
        dc.setR(R);
        dc.play(N, dx.get(), out.get());
        dump(dir, "dcblock", out);

        maxiEnvGenBank eg(V);
        eg.setupADSR(2, 3, 0.5, 4);
        std::vector<double> gate(N);
        for (size_t n = 0; n < N; n++) gate[n] = (n % 200) < 120 ? 1.0 : -1.0;
        DeviceArray<double> dgate(N);
        dgate.upload(gate);
        eg.play(N, dgate.get(), false, out.get());
        dump(dir, "envgen", out);

        maxiMixBank mix(V);
        mix.setPan(px, py);
        DeviceArray<double> m4(N * 4);
        mix.quad(N, dx.get(), m4.get());
        dump(dir, "quad", m4);

        maxiSampleBank sb(V);
        if (!sb.load(wav)) { fprintf(stderr, "load failed: %s\n", mxg_last_error()); return 3; }
        std::vector<double> trig(N * V);
        for (size_t n = 0; n < N; n++)
            for (size_t v = 0; v < V; v++) trig[n * V + v] = (double)((n * 3 + v * 7) % 64) / 32.0 - 1.0;
        DeviceArray<double> dtrig(N * V);
        dtrig.upload(trig);
        sb.playOnZX(N, dtrig.get(), out.get());
        dump(dir, "playonzx", out);

        const int fs = 256, hop = 64;
        const size_t nframes = 40;
        std::vector<float> sig(fs + hop * (nframes - 1));
        for (size_t i = 0; i < sig.size(); i++) sig[i] = (float)((double)((i * 29) % 200) / 200.0 - 0.5);
        DeviceArray<float> dsig(sig.size()), mags(nframes * fs / 2), phases(nframes * fs / 2), flat(nframes), cen(nframes),
            resynth(nframes * hop);
        dsig.upload(sig);
        maxiFFTBatch fft;
        fft.setup(fs, hop, fs);
        fft.process(dsig.get(), hop, nframes, mags.get(), phases.get());
        fft.features(mags.get(), nframes, nullptr, flat.get(), cen.get());
        dump(dir, "mags", mags);
        dump(dir, "centroid", cen);
        maxiIFFTBatch ifft;
        ifft.setup(fs, hop, fs);
        ifft.process(mags.get(), phases.get(), nframes, resynth.get());
        dump(dir, "resynth", resynth);

        maxiSampleBank one(1);
        if (!one.load(wav)) return 3;
        const size_t S = 32, T = 1500;
        maxiPitchShiftBank ps(S, &one, 0);
        std::vector<double> speed(S);
        for (size_t s = 0; s < S; s++) speed[s] = 0.5 + 0.05 * s;
        ps.setSpeeds(speed);
        DeviceArray<double> gout(T * S);
        ps.play(0.01, 3, T, gout.get());
        dump(dir, "pitchshift", gout);

        // round 3: maxiStretch bank, the block convolver and a bank of samplers
        maxiStretchBank sbk(S, &one, 4);  // triangle window
        std::vector<double> rate(S);
        for (size_t s = 0; s < S; s++) rate[s] = 0.25 + 0.05 * s;
        sbk.setPitch(speed);
        sbk.setRate(rate);
        sbk.play(0.02, 2, T, gout.get());
        dump(dir, "stretch", gout);

        std::vector<double> imp = one.download();
        maxiConvolveBlock cv;
        cv.setup(imp, (double)imp.size(), fs, hop);
        const size_t nb = 6;
        std::vector<float> cin(nb * fs);
        for (size_t i = 0; i < cin.size(); i++) cin[i] = (float)((double)((i * 13) % 97) / 97.0 - 0.5);
        DeviceArray<float> dcin(cin.size()), dcout(cin.size());
        dcin.upload(cin);
        cv.play(dcin.get(), nb, dcout.get(), true);
        dump(dir, "convolve", dcout);

        const size_t NS = 3, NP = 900;
        maxiSamplerBank sp(NS, 4, &one);
        DeviceArray<double> smix(NP * NS);
        for (size_t k = 0; k < NS; k++) {
            sp.midiNoteOn(k, -2.0 + 3.0 * k, 100.0 + k);
            sp.trigger(k);
        }
        sp.play(300, smix.get());
        sp.midiNoteOff(1, 1.0);
        sp.midiNoteOn(0, 4.0, 64.0);
        sp.trigger(0);
        sp.play(600, smix.get() + 300 * NS);
        dump(dir, "sampler", smix);
        mxg_sync();
    } catch (const std::exception &e) {
        fprintf(stderr, "facade_smoke: %s\n", e.what());
        return 4;
    }
    return 0;
}
