// tests/host_envgen.cpp -- maxiEnvGen::play as the device runs it per lane (maximilian_amd/csrc/mxg_envgen.h), compiled for
// the host: one envelope after the other, every sample through envgen_tick.  tests/test_envgen_host.py compares it with
// the oracle from arbitrary (valid-index) states under random triggers.
#include "mxg_envgen.h"

using namespace mxg;

// fast = 0: every sample through envgen_tick.  fast = 1: chunks of 8 samples through envgen_steady_chunk when it accepts them
// (what a wavefront does when every lane accepts), the rest through envgen_tick; returns the per-mille of chunks it took.
extern "C" int envgen_host(size_t V, size_t N, const double *trig, int tpv, const double *stages, int nstages, int loop,
                           int retrigger, double *dst, int64_t *ist, double *out, int fast) {
    size_t chunks = 0, steady = 0;
    for (size_t v = 0; v < V; v++) {
        EgState e;
        envgen_load(e, V, v, dst, ist);
        size_t n = 0;
        if (fast) {
            constexpr int U = 8;
            for (; n + U <= N; n += U) {
                double t[U], o[U];
                for (int i = 0; i < U; i++) t[i] = tpv ? trig[(n + i) * V + v] : trig[n + i];
                EgState s = e;
                chunks++;
                const EgRow row = envgen_row(stages, nstages, e.phase);
                if (envgen_steady_chunk<U>(s, row, retrigger != 0, envgen_cross<U>(t), o)) {
                    e = s;
                    steady++;
                    for (int i = 0; i < U; i++) out[(n + i) * V + v] = o[i];
                } else {
                    for (int i = 0; i < U; i++)
                        out[(n + i) * V + v] = envgen_tick(e, stages, nstages, loop != 0, retrigger != 0, t[i]);
                }
            }
        }
        for (; n < N; n++)
            out[n * V + v] = envgen_tick(e, stages, nstages, loop != 0, retrigger != 0, tpv ? trig[n * V + v] : trig[n]);
        envgen_store(e, nstages, V, v, dst, ist);
    }
    return chunks ? (int)((double)steady * 1000.0 / (double)chunks) : 0;
}
