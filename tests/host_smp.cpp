// tests/host_smp.cpp -- the maxiSample players of maximilian_amd/csrc/mxg_smp.h (smp_gen / smp_eval: what sample.hip and
// sampler.hip run per lane) compiled for the host, one voice after the other in the plain order gen -> gather -> eval.
// tests/test_smp_host.py compares it with the oracle over parameter ranges far wider than the GPU tests can afford.
#include <stdint.h>

#include "mxg_smp.h"

using namespace mxg;

template <int MODE>
static void run(size_t V, size_t N, const double *amp, size_t len, double step_div, double sr, const double *a, int aps,
                const double *start, const double *end, double *position, double *out) {
    for (size_t v = 0; v < V; v++) {
        Smp s = {amp, len, position[v], step_div, 0.0, false, 0.0, 0.0};
        const double st = start ? start[v] : 0.0, en = end ? end[v] : 1.0;
        for (size_t n = 0; n < N; n++) {
            const double x = a ? (aps ? a[n * V + v] : a[v]) : 1.0;
            SmpReq<MODE> q;
            smp_gen<MODE>(s, x, 0.0, st, en, sr, q);
            double val[SmpReq<MODE>::L];
            for (int l = 0; l < SmpReq<MODE>::L; l++) val[l] = amp[q.idx[l]];
            out[n * V + v] = smp_eval<MODE>(q, val);
        }
        position[v] = s.pos;
    }
}

extern "C" int smp_host(int mode, size_t V, size_t N, const double *amp, size_t len, int sampleRate, int mySampleRate,
                        const double *a, int aps, const double *start, const double *end, double *position,
                        double *out) {
    const double step_div = (double)((size_t)sampleRate / (size_t)mySampleRate), sr = (double)sampleRate;
    aps = aps && mode >= 4;
    switch (mode) {
        case 0: run<0>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 1: run<1>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 2: run<2>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 3: run<3>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 4: run<4>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 5: run<5>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 6: run<6>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 7: run<7>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        case 8: run<8>(V, N, amp, len, step_div, sr, a, aps, start, end, position, out); break;
        default: return -1;
    }
    return 0;
}
