// tests/host_sched_events.cpp -- the event-driven walk of the granular scheduler (maximilian_amd/csrc/mxg_sched.h:
// sched_run_events, what K8a runs per stream) against the one-sample step (sched_step, the transcription of
// L/maxiGrains.h:341-355, 359-367, 412-430, 512-530 that the GPU tests pin to the oracle): same births at the same samples
// with the same (pos0, inc) bits, same final state.  Built and run by tests/test_sched_host.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>

#include "mxg_sched.h"

using namespace mxg;

struct Birth {
    int n;
    double pos0, inc;
};
static bool same(double a, double b) { return memcmp(&a, &b, 8) == 0; }

template <int MODE>
static long run_mode(long cases, std::mt19937_64 &g, long &births_total) {
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    long bad = 0;
    for (long i = 0; i < cases; i++) {
        const int Tn = 1 + (int)(g() % 6000);
        SchedConst c;
        const size_t len = (g() % 3 == 0) ? 3000 + g() % 5000 : 44100 + g() % 400000;
        c.dlen = (double)len;
        c.sr = 44100.0;
        c.grainLength = (g() % 2) ? 0.05 : 0.001 + 0.1 * u01(g);
        const int overlaps = 1 + (int)(g() % 7);
        c.cycleLength = c.grainLength * 44100.0 / overlaps;
        c.sampleDur = (int)(c.grainLength * 44100.0);
        c.pm = (g() % 2) ? 0.0 : 0.1 * (u01(g) - 0.5);
        double a;
        switch (g() % 5) {
            case 0: a = 0.25 + 1.5 * u01(g); break;
            case 1: a = (double)(1 + g() % 64) / 32.0; break;
            case 2: a = ldexp(1.0 + u01(g), -(int)(g() % 30)); break;
            case 3: a = -(0.1 + u01(g)); break;        // negative speed: the stepwise path must take over
            default: a = 1.0; break;
        }
        c.speed = MODE == 2 ? 1.0 : a;
        c.rate = MODE == 1 ? (g() % 4 ? 0.05 + 4.0 * u01(g) : -0.3) : c.speed;
        std::vector<double> ps;
        if (MODE == 2) {
            ps.resize((size_t)Tn);
            for (auto &p : ps) p = 1.4 * u01(g) - 0.2;
        }
        c.a_ps = MODE == 2 ? ps.data() : nullptr;
        c.S = 1;
        std::vector<int32_t> rnd(64);
        for (auto &r : rnd) r = (int32_t)(g() % 10);
        const bool use_rnd = MODE <= 1 && (g() % 2);
        c.rnd = use_rnd ? rnd.data() : nullptr;
        c.R = rnd.size();
        SchedState q0;
        q0.position = (g() % 8 == 0) ? c.dlen : c.dlen * u01(g);
        q0.looper = (MODE >= 2) ? (double)(g() % 100000) + ((g() % 16 == 0) ? 0.5 : 0.0) : c.cycleLength * 1.2 * u01(g);
        q0.randomOffset = use_rnd ? (double)(g() % 10) : 0.0;
        q0.cursor = 0;
        q0.thr = c.cycleLength + q0.randomOffset;
        // A: event-driven, then the one-sample step for whatever it left
        std::vector<Birth> ba, bb;
        SchedState qa = q0, qb = q0;
        int fa = 0, fb = 0;
        const int nstart = sched_run_events<MODE>(qa, c, Tn, true, fa, [&](int n, double p, double inc) { ba.push_back({n, p, inc}); });
        for (int n = nstart; n < Tn; n++) {
            double p, inc;
            if (sched_step<MODE>(qa, c, (size_t)n, p, inc, fa)) ba.push_back({n, p, inc});
        }
        // B: the one-sample step alone
        for (int n = 0; n < Tn; n++) {
            double p, inc;
            if (sched_step<MODE>(qb, c, (size_t)n, p, inc, fb)) bb.push_back({n, p, inc});
        }
        bool ok = ba.size() == bb.size() && fa == fb && same(qa.position, qb.position) && same(qa.looper, qb.looper) &&
                  same(qa.randomOffset, qb.randomOffset) && qa.cursor == qb.cursor;
        for (size_t k = 0; ok && k < ba.size(); k++)
            ok = ba[k].n == bb[k].n && same(ba[k].pos0, bb[k].pos0) && same(ba[k].inc, bb[k].inc);
        if (!ok) {
            if (bad < 3)
                printf("mode %d mismatch: Tn=%d len=%zu cyc=%a speed=%a rate=%a pos0=%a looper0=%a: %zu vs %zu births, "
                       "pos %a vs %a, looper %a vs %a\n", MODE, Tn, len, c.cycleLength, c.speed, c.rate, q0.position,
                       q0.looper, ba.size(), bb.size(), qa.position, qb.position, qa.looper, qb.looper);
            bad++;
        }
        births_total += (long)bb.size();
    }
    return bad;
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 20000;
    std::mt19937_64 g(0x4D415849);
    long births = 0;
    const long b0 = run_mode<0>(cases, g, births), b1 = run_mode<1>(cases, g, births), b2 = run_mode<2>(cases, g, births),
               b3 = run_mode<3>(cases, g, births);
    printf("event-driven walk vs one-sample step: %ld streams per mode, %ld births, mismatches %ld %ld %ld %ld\n", cases, births,
           b0, b1, b2, b3);
    return (b0 || b1 || b2 || b3) ? 1 : 0;
}
