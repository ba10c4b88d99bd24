// tests/host_env.cpp -- maxiEnv as the device runs it per lane (maximilian_amd/csrc/mxg_env.h: the predicated state machine
// env_adsr / env_ar and the steady-state sustain / release ticks with their entry tests), compiled for the host.
// tests/test_env_host.py compares it with the oracle from arbitrary states (any flag values, any amplitude).
#include "mxg_env.h"

using namespace mxg;

// fast = 1: per sample, take the steady-state tick whenever its entry test holds (what a wavefront does when all its lanes
// agree); fast = 2: the general steady chunk; fast = 0: always the state machine.  All must give the oracle's bits.
extern "C" int env_host(int mode, int fast, size_t V, size_t N, const double *x, const int32_t *trig, int tpv,
                        const double *par, const int64_t *holdtime, double *dst, int64_t *ist, double *out) {
    int steady = 0;  // fast = 2: chunks taken by env_steady_chunk, per mille of all full chunks
    size_t chunks = 0;
    for (size_t v = 0; v < V; v++) {
        Env e;
        env_load(e, V, v, par, holdtime, dst, ist);
        if (fast == 2 && mode == 0) {
            // fast = 2: chunks of 8 samples with a constant gate go through env_steady_chunk when it accepts them (what a
            // wavefront does when every lane accepts), everything else through the state machine
            constexpr int U = 8;
            size_t n = 0;
            for (; n + U <= N; n += U) {
                double xc[U], o[U];
                bool constant = true;
                const int t0 = tpv ? trig[n * V + v] : trig[n];
                for (int i = 0; i < U; i++) {
                    xc[i] = x ? x[(n + i) * V + v] : 1.0;
                    const int t = tpv ? trig[(n + i) * V + v] : trig[n + i];
                    constant = constant && ((t == 1) == (t0 == 1));
                }
                Env s = e;
                chunks++;
                if (constant && env_steady_chunk<U>(s, xc, t0 == 1, o)) {
                    e = s;
                    steady++;
                    for (int i = 0; i < U; i++) out[(n + i) * V + v] = o[i];
                } else {
                    for (int i = 0; i < U; i++) out[(n + i) * V + v] = env_adsr(e, xc[i], tpv ? trig[(n + i) * V + v] : trig[n + i]);
                }
            }
            for (; n < N; n++) out[n * V + v] = env_adsr(e, x ? x[n * V + v] : 1.0, tpv ? trig[n * V + v] : trig[n]);
            env_store(e, V, v, dst, ist);
            continue;
        }
        for (size_t n = 0; n < N; n++) {
            const double in = x ? x[n * V + v] : 1.0;
            const int t = tpv ? trig[n * V + v] : trig[n];
            double o;
            if (mode == 1) o = env_ar(e, in, t);
            else if (fast && t == 1 && env_in_sustain(e)) o = env_sustain_tick(e, in);
            else if (fast && t != 1 && env_in_release(e)) o = env_release_tick(e, in);
            else o = env_adsr(e, in, t);
            out[n * V + v] = o;
        }
        env_store(e, V, v, dst, ist);
    }
    return chunks ? (int)((double)steady * 1000.0 / (double)chunks) : 0;
}
