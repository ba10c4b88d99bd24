// tests/host_osc.cpp -- maxiOsc's waveforms as the device runs them per lane (maximilian_amd/csrc/mxg_osc.h: osc_pre /
// osc_tick over the two static tables), compiled for the host.  tests/test_osc_host.py compares it with the oracle from
// arbitrary phase / held-output states and frequencies of either sign.
#include <stdint.h>

#include "maxi_tables.h"
#include "mxg_osc.h"

using namespace mxg;

static const double kSine[MAXI_SINE_TAB_LEN] = MAXI_SINE_TAB_INIT;
static const double kTrans[MAXI_TRANS_TAB_LEN] = MAXI_TRANS_TAB_INIT;

template <int WF>
static void run(size_t V, size_t N, double sr, const double *freq, const double *p1, const double *p2, double *phase,
                double *hold, double *out) {
    for (size_t v = 0; v < V; v++) {
        const OscPre q = osc_pre<WF>(freq[v], sr, p1 ? p1[v] : 0.0, p2 ? p2[v] : 0.0);
        double ph = phase[v], hd = hold[v];
        for (size_t n = 0; n < N; n++) out[n * V + v] = osc_tick<WF>(ph, hd, q, kSine, kTrans);
        phase[v] = ph;
        hold[v] = hd;
    }
}

extern "C" int osc_host(int wf, size_t V, size_t N, int sampleRate, const double *freq, const double *p1, const double *p2,
                        double *phase, double *hold, double *out) {
    const double sr = (double)sampleRate;
    switch (wf) {
#define CASE(W) case W: run<W>(V, N, sr, freq, p1, p2, phase, hold, out); break;
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11)
#undef CASE
        default: return -1;  // sinewave / coswave: libm-internal arithmetic, see tests/host_sincos_accuracy.cpp
    }
    return 0;
}
