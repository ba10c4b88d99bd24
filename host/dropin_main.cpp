// host/dropin_main.cpp -- the headless host for UNMODIFIED reference patches (`void setup(); void play(double*);`,
// src/maximilian.cpp:205-207) compiled against the drop-in header include/maximilian.h: the same loop as
// cpp/commandline/player.cpp:25-44 through the C-ABI's mxg_host_render, bufferSize frames per "callback".
//   dropin_<patch> <frames> <out.f64>      writes frames*channels interleaved doubles; statistics on stderr
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "maximilian.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <frames> <out.f64>\n", argv[0]);
        return 2;
    }
    const size_t frames = (size_t)std::atol(argv[1]);
    try {
        setup();
        const size_t ch = maxiSettings::channels, buf = maxiSettings::bufferSize;
        std::vector<double> out(frames * ch), last(ch, 0.0);
        const auto t0 = std::chrono::steady_clock::now();
        auto t_half = t0;  // the second half of the run is timed by itself: the first launches pay for the HIP runtime's start-up
        size_t half_at = 0;
        for (size_t done = 0; done < frames; done += buf) {
            const size_t n = frames - done < buf ? frames - done : buf;
            if (half_at == 0 && done >= frames / 2) {
                half_at = done;
                t_half = std::chrono::steady_clock::now();
            }
            if (mxg_host_render(play, ch, n, out.data() + done * ch, last.data()) < 0) {
                std::fprintf(stderr, "mxg_host_render: %s\n", mxg_last_error());
                return 1;
            }
        }
        const auto t1 = std::chrono::steady_clock::now();
        const double secs = std::chrono::duration<double>(t1 - t0).count();
        const double steady = std::chrono::duration<double>(t1 - t_half).count();
        if (half_at) std::fprintf(stderr, "steady state: %zu frames in %.6f s\n", frames - half_at, steady);
        FILE *f = std::fopen(argv[2], "wb");
        if (!f) return 3;
        std::fwrite(out.data(), sizeof(double), out.size(), f);
        std::fclose(f);
        using namespace maxigpu::ps;
        std::fprintf(stderr, "rendered %zu frames x %zu channels in %.3f s; launches: osc %zu (async blocks %zu), env %zu (%zu), filter %zu\n",
                     frames, ch, secs, pool<OscPool>().launches, pool<OscPool>().async_hits, pool<EnvPool>().launches,
                     pool<EnvPool>().async_hits, pool<FilterPool>().launches);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "dropin host: %s\n", e.what());
        return 1;
    }
    return 0;
}
