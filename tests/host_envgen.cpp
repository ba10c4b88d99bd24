// tests/host_envgen.cpp -- maxiEnvGen::play as the device runs it per lane (maximilian_amd/csrc/mxg_envgen.h), compiled for
// the host: one envelope after the other, every sample through envgen_tick.  tests/test_envgen_host.py compares it with
// the oracle from arbitrary (valid-index) states under random triggers.
#include "mxg_envgen.h"

using namespace mxg;

extern "C" int envgen_host(size_t V, size_t N, const double *trig, int tpv, const double *stages, int nstages, int loop,
                           int retrigger, double *dst, int64_t *ist, double *out) {
    for (size_t v = 0; v < V; v++) {
        EgState e;
        envgen_load(e, V, v, dst, ist);
        for (size_t n = 0; n < N; n++)
            out[n * V + v] = envgen_tick(e, stages, nstages, loop != 0, retrigger != 0, tpv ? trig[n * V + v] : trig[n]);
        envgen_store(e, nstages, V, v, dst, ist);
    }
    return 0;
}
