// oracle/example_host.cpp -- TEST INFRASTRUCTURE: the same headless host as host/dropin_main.cpp, linked with the
// UNMODIFIED reference sources and one of the reference's own example patches (oracle/Makefile builds
// oracle/_ref/example_<n>).  It is the oracle for the drop-in header (tests/golden/dropin.npz is dumped from it) and the
// config-1 CPU baseline: cpp/commandline/main.cpp through the loop of cpp/commandline/player.cpp:25-44.
//   example_<n> <frames> <out.f64>      prints the CPU time of the render loop on stderr
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "maximilian.h"

void setup();
void play(double *output);

static int routing(double *buffer, unsigned int nBufferFrames, double *lastValues) {  // player.cpp:25-44
    for (unsigned int i = 0; i < nBufferFrames; i++) {
        play(lastValues);
        for (unsigned int j = 0; j < maxiSettings::channels; j++) *buffer++ = lastValues[j];
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const size_t frames = (size_t)std::atol(argv[1]);
    setup();
    const size_t ch = maxiSettings::channels, buf = maxiSettings::bufferSize;
    std::vector<double> out(frames * ch), last(ch, 0.0);
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t done = 0; done < frames; done += buf)
        routing(out.data() + done * ch, (unsigned int)(frames - done < buf ? frames - done : buf), last.data());
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 3;
    std::fwrite(out.data(), sizeof(double), out.size(), f);
    std::fclose(f);
    std::fprintf(stderr, "reference CPU: %zu frames x %zu channels in %.6f s = %.3f Msamples/s\n", frames, ch, secs,
                 frames * ch / secs / 1e6);
    return 0;
}
