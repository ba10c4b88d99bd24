// tests/host_ps_engine.cpp -- the per-sample engine of include/maximilian.h (block prediction, derived arguments, rewinds, lock-step
// groups, zero-copy renders) driven on the HOST: the C-ABI entry points the engine itself uses (memory, copies, streams, events) are
// stubbed with plain host memory, and a test pool renders a simple recurrence on the CPU.  This is a harness for the engine's LOGIC
// (tests/test_ps_engine_host.py, -m "not gpu"); the product pools render through libmaxigpu.so and have no such path.
//
// A "patch" of eight objects is called once per sample, in order, with arguments computed from the values the earlier calls RETURNED
// -- constants, affine forms of one output, the sum of two outputs, a non-linear map, a constant that changes now and then, and a
// one-call perturbation every few hundred samples (a prediction that fails inside a block).  Every returned value must equal, bit for
// bit, what the same recurrence gives when it is simply evaluated call by call; the derivable objects must need far fewer renders than
// calls, the non-linear one about one per call.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../include/maxigpu.h"

// ---- the C-ABI the engine uses, on host memory, with a DEFERRED stream ------------------------------------------------------------
// Copies, kernels (the test pool's renders) and event records are queued and run only when the host waits (stream / event sync) or
// when an event is polled -- in order, like a stream -- so the engine's asynchronous next-block renders, event polls and buffer
// re-use are exercised against work that has really not happened yet.  (MXG_STUB_EAGER=1: everything runs at once.)
#include <deque>
#include <functional>
#include <vector>
static std::deque<std::function<void()>> g_q;
static const bool g_eager = [] { const char *e = std::getenv("MXG_STUB_EAGER"); return e && e[0] == '1'; }();
static void submit(std::function<void()> f) { if (g_eager) f(); else g_q.push_back(std::move(f)); }
static void drain() { while (!g_q.empty()) { auto f = std::move(g_q.front()); g_q.pop_front(); f(); } }
struct StubEvent { bool done = true; };
extern "C" {
int mxg_init(int) { return 0; }
const char *mxg_last_error(void) { return "stub"; }
void *mxg_malloc(size_t b) { return std::calloc(b ? b : 8, 1); }
int mxg_free(void *p) { drain(); std::free(p); return 0; }
void *mxg_host_alloc(size_t b) { return std::calloc(b ? b : 8, 1); }
int mxg_host_free(void *p) { drain(); std::free(p); return 0; }
// (an asynchronous H2D copy reads its pinned source when the stream reaches it: the engine must keep it intact until then)
int mxg_memcpy_h2d_async(void *d, const void *s, size_t b, void *) { submit([=] { std::memcpy(d, s, b); }); return 0; }
int mxg_memcpy_d2h_async(void *d, const void *s, size_t b, void *) { submit([=] { std::memcpy(d, s, b); }); return 0; }
int mxg_memcpy_d2d_async(void *d, const void *s, size_t b, void *) { submit([=] { std::memmove(d, s, b); }); return 0; }
void *mxg_stream_create(void) { static int dummy; return &dummy; }
int mxg_stream_destroy(void *) { drain(); return 0; }
int mxg_stream_sync(void *) { drain(); return 0; }
void *mxg_event_create(void) { return new StubEvent; }
int mxg_event_destroy(void *e) { drain(); delete static_cast<StubEvent *>(e); return 0; }
int mxg_event_record(void *e, void *) { StubEvent *ev = static_cast<StubEvent *>(e); ev->done = false; submit([ev] { ev->done = true; }); return 0; }
int mxg_event_sync(void *e) { StubEvent *ev = static_cast<StubEvent *>(e); while (!ev->done && !g_q.empty()) { auto f = std::move(g_q.front()); g_q.pop_front(); f(); } return 0; }
int mxg_event_query(void *e) {  // a poll lets the stream make a little progress, like a device running beside the host
    StubEvent *ev = static_cast<StubEvent *>(e);
    for (int i = 0; i < 3 && !ev->done && !g_q.empty(); i++) { auto f = std::move(g_q.front()); g_q.pop_front(); f(); }
    return ev->done ? 1 : 0;
}
}

#include "../include/maximilian.h"

using namespace maxigpu::ps;

// a ramp whose rate follows a0 (it never settles, so every downstream argument keeps moving):
//   y <- y + 0.001 + a0 * 0.001, wrapped into [0, 1);   out = (2 y - 1) * a2 + a1     (state: y; a0, a1 derivable, a2 a plain parameter)
static double step(double &y, double a0, double a1, double a2) {
    y = y + 0.001 + a0 * 0.001;
    if (y >= 1.0) y -= 1.0;
    if (y < 0.0) y += 1.0;
    return (2.0 * y - 1.0) * a2 + a1;
}
struct TestPool : Pool {
    TestPool() : Pool(1, 0) {}
    unsigned derivable(int) const override { return 3u; }
    void enqueue(Group &G) override {  // "launch": the arguments are captured now (as a launch copies its parameters), the work runs later
        const size_t n = G.m.size(), L = G.L;
        std::vector<double> a0(L * n), a1(L * n), a2(n);
        for (size_t j = 0; j < n; j++) {
            a2[j] = G.sig[j].a[2];
            for (size_t t = 0; t < L; t++) { a0[t * n + j] = G.arg(j, 0, t); a1[t * n + j] = G.arg(j, 1, t); }
        }
        double *st = G.d_state.p, *out = G.d_out.p;
        submit([=] {
            for (size_t j = 0; j < n; j++) {
                double y = st[j];
                for (size_t t = 0; t < L; t++) out[t * n + j] = step(y, a0[t * n + j], a1[t * n + j], a2[j]);
                st[j] = y;
            }
        });
    }
};
struct Obj {
    Slot slot;
    double y = 0;  // the truth: the same recurrence, call by call
    Obj() { pool<TestPool>().attach(slot); }
    ~Obj() { pool<TestPool>().detach(slot); }
    double call(double a0, double a1, double a2, double &truth) {
        Call c;
        c.method = 0;
        c.a[0] = a0; c.a[1] = a1; c.a[2] = a2;
        truth = step(y, a0, a1, a2);
        return pool<TestPool>().call(slot, c);
    }
};

// ---- randomised patches: `fuzz <frames> <seed>` ---------------------------------------------------------------------------------
// A random graph of objects (each argument a constant, a form of one or two EARLIER objects' outputs, or a non-linear map), random
// events (a one-call perturbation, a constant that changes, an object's state settled through the pool as a setter does, an object
// destroyed and replaced while others' forms still name it), compared call by call with the plain evaluation.
struct Rng {
    unsigned long long s;
    unsigned next() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(s >> 33); }
    double uni() { return next() / 2147483648.0; }
    int below(int n) { return (int)(next() % (unsigned)n); }
};
struct Recipe { int kind = 0, p1 = 0, p2 = 0; double a = 1, b = 0; };  // 0 const b | 1 x | 2 x*a+b | 3 (x+b)*a | 4 x1+x2 | 5 x1*x2 | 6 sin(x) | 7 (x1+x2)*a
static double eval(const Recipe &r, const double *out) {
    double t;
    switch (r.kind) {
        case 1: return out[r.p1];
        case 2: t = out[r.p1] * r.a; t = t + r.b; return t;
        case 3: t = out[r.p1] + r.b; t = t * r.a; return t;
        case 4: t = out[r.p1] + out[r.p2]; return t;
        case 5: t = out[r.p1] * out[r.p2]; return t;
        case 6: return std::sin(out[r.p1] * 3.0);
        case 7: t = out[r.p1] + out[r.p2]; t = t * r.a; return t;
    }
    return r.b;
}
static int fuzz(long frames, unsigned long long seed) {
    Rng R{seed * 2654435761ULL + 12345};
    const int N = 6 + R.below(7);
    const double nice[8] = {0.5, 2.0, 1.5, 0.125, 3.0, 10.0, 0.75, 440.0};
    std::vector<Obj *> obj((size_t)N);
    std::vector<Recipe> r0((size_t)N), r1((size_t)N);
    std::vector<double> par((size_t)N, 1.0), out((size_t)N, 0.0);
    auto recipe = [&](int i) {
        Recipe r;
        r.kind = i == 0 ? 0 : R.below(8);
        r.p1 = i ? R.below(i) : 0;
        r.p2 = i ? R.below(i) : 0;
        r.a = nice[R.below(8)] * (R.below(4) ? 1.0 : 0.001);
        r.b = R.below(3) ? nice[R.below(8)] * 0.01 : R.uni();  // a short decimal, or any double
        return r;
    };
    for (int i = 0; i < N; i++) { obj[(size_t)i] = new Obj; r0[(size_t)i] = recipe(i); r1[(size_t)i] = recipe(i); if (R.below(2)) r1[(size_t)i].kind = 0; }
    long bad = 0;
    for (long n = 0; n < frames; n++) {
        const int ev = R.below(400);
        const int who = R.below(N);
        if (ev == 0) r0[(size_t)who].b += 0.25;                         // a constant changes
        if (ev == 1) par[(size_t)who] = par[(size_t)who] == 1.0 ? 0.5 : 1.0;
        if (ev == 2) {                                                   // a setter: settle the object, edit its state, mirror the edit
            pool<TestPool>().settle(obj[(size_t)who]->slot);
            obj[(size_t)who]->slot.sd[0] = 0.125;
            obj[(size_t)who]->y = 0.125;
        }
        if (ev == 3 && who > 0) {                                        // the object goes away and a fresh one takes its place
            delete obj[(size_t)who];
            obj[(size_t)who] = new Obj;
        }
        if (ev == 4) r0[(size_t)who] = recipe(who);                      // the patch changes what it feeds the object
        for (int i = 0; i < N; i++) {
            double a0 = eval(r0[(size_t)i], out.data()), a1 = eval(r1[(size_t)i], out.data());
            if (ev == 5 && i == who) a0 += 1e-7;                         // one call off every form
            double t;
            const double got = obj[(size_t)i]->call(a0, a1, par[(size_t)i], t);
            if (std::memcmp(&got, &t, 8)) {
                if (bad < 5) std::printf("MISMATCH seed %llu object %d of %d frame %ld: %.17g != %.17g\n", seed, i, N, n, got, t);
                bad++;
            }
            out[(size_t)i] = got;
        }
    }
    TestPool &P = pool<TestPool>();
    std::printf("fuzz seed %llu: %d objects, %ld frames, %ld mismatches; %zu renders, %zu blocks with derived arguments\n", seed, N, frames, bad, P.launches,
                P.derived_blocks);
    for (Obj *o : obj) delete o;
    return bad == 0 ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc > 1 && !std::strcmp(argv[1], "fuzz")) return fuzz(argc > 2 ? std::atol(argv[2]) : 20000, argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1);
    const long frames = argc > 1 ? std::atol(argv[1]) : 30000;
    Obj A, B, C, D, E, F0, F1, G;
    long bad = 0, calls = 0;
    auto chk = [&](double got, double truth, const char *who, long n) {
        calls++;
        if (std::memcmp(&got, &truth, 8)) {
            if (bad < 5) std::printf("MISMATCH %s at frame %ld: %.17g != %.17g\n", who, n, got, truth);
            bad++;
        }
    };
    double gain = 0.25, t;
    for (long n = 0; n < frames; n++) {
        if (n % 4100 == 4099) gain = gain == 0.25 ? 0.75 : 0.25;  // a parameter that changes now and then
        const double a = A.call(0.37, 0.0, 1.0, t); chk(a, t, "A (constant arguments)", n);
        double barg = a * 3.0 + 1.0;                                   // x * a + b
        if (n % 777 == 776) barg += 1e-9;                              // one call off the form: a prediction fails inside a block
        const double b = B.call(barg, 0.0, gain, t); chk(b, t, "B (x*3+1, perturbed every 777 frames)", n);
        const double c = C.call(a + b, 0.5, 1.0, t); chk(c, t, "C (x1 + x2)", n);
        const double d = D.call(std::fabs(std::sin(c)), 0.0, 1.0, t); chk(d, t, "D (a non-linear map: never derived)", n);
        const double e = E.call((c + 2.0) * 0.125, b, 0.5, t); chk(e, t, "E ((x+b)*a and a second derived argument)", n);
        const double f0 = F0.call(e, 0.0, 1.0, t); chk(f0, t, "F0 (identity; lock-step with F1)", n);
        const double f1 = F1.call(e, 0.0, 1.0, t); chk(f1, t, "F1", n);
        const double g = G.call(f0 * f1, 523.2511306011972 + (a * 1.5), 1.0, t); chk(g, t, "G (x1*x2; x*a + a constant that is no short decimal)", n);
    }
    const Obj *objs[8] = {&A, &B, &C, &D, &E, &F0, &F1, &G};
    const char *names = "ABCDEFfG";
    for (int i = 0; i < 8; i++) std::printf("  %c: %llu of %ld calls not served from a cached block\n", names[i], (unsigned long long)objs[i]->slot.misses, frames);
    TestPool &P = pool<TestPool>();
    std::printf("%ld frames, %ld calls, %ld mismatches; %zu renders (%zu blocks with derived arguments, %zu asynchronous blocks)\n", frames, calls, bad,
                P.launches, P.derived_blocks, P.async_hits);
    // (eight objects: one render per call would be `calls` renders; D alone -- a non-linear map -- needs ~frames of them.  The caller
    // judges the counts: they depend on MXG_PS_DERIVE.)
    return bad == 0 ? 0 : 1;
}
