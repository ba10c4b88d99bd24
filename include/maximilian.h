// include/maximilian.h -- the DROP-IN header: the reference's own class names and per-sample method signatures
// (src/maximilian.h, src/libs/maxiFFT.h, src/libs/maxiMFCC.h of micknoise/Maximilian), executed on the GPU through
// the C-ABI of libmaxigpu.so (include/maxigpu.h).  An unmodified `void setup(); void play(double *output);` patch --
// e.g. cpp/commandline/maximilian_examples/14.monosynth/main.cpp or 15.polysynth/main.cpp -- compiles against this
// file instead of the reference's src/maximilian.h and produces the reference's samples bit for bit (sinewave /
// coswave: <= 1 ULP); tests/test_gpu_dropin.py builds those two example files verbatim and checks exactly that.
//
// How a one-sample call is served by a block renderer.  Every maxiOsc / maxiEnv / maxiFilter / maxiSample object is a
// SLOT of a process-wide pool (one pool per class).  A call `osc.pulse(f, d)` is looked up in the block the pool has
// already rendered for that slot under the prediction "the object keeps being called with the arguments of its last
// call".  While the prediction holds, a call is an array read; blocks grow 1 -> 8 -> 64 -> 512 samples as long as it
// keeps holding, objects called in lock-step (the `for (i < 6)` of a polysynth) are rendered together in ONE launch
// (voices of a bank), and at the full block length the NEXT block is rendered asynchronously on the pool's stream into
// pinned host memory while the current one is being served (no render inside the audio thread in steady state).
// When a call arrives with other arguments -- a new pitch, a trigger, an input that is itself a signal -- the slot is
// rewound to the state it had at that sample (the block is re-advanced from its start state: same kernel, same bits)
// and continues from there with blocks of one sample.  Nothing of the signal path is ever computed on the CPU.  An
// argument that is itself a signal (a filter fed by `(VCO1out+VCO2out)*0.5`, a cutoff of `ADSRout*10000`) is predicted
// PER SAMPLE: the pool recognises small expressions of other objects' recent outputs and fills the consumer's next block
// from the producers' cached blocks ("derived arguments", below; every call is still verified bit for bit).  Only an
// argument it cannot recognise (a pow / fabs of another output, noise()'s rand() draw) costs one small launch per call;
// the throughput path for big graphs is the fused bank API (include/maximilian_bank.hpp, maxiVoiceBank).
// State is authoritative on the host between launches (a few doubles per object), so rewinding and regrouping are
// plain copies.  maxiFilter's cos/pow/sqrt coefficients are evaluated with THIS machine's libm
// (mxg_filter_coeffs_host), which is what keeps the recursive filter bit-identical to the reference.
//
// Not thread-safe, like the reference: all calls come from the one audio thread (cpp/commandline/player.cpp:25-44).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "maxigpu.h"

using namespace std;  // the reference header does (src/maximilian.h:54); example patches rely on it (cout, vector)

#ifndef PI
#define PI 3.1415926535897932384626433832795
#endif
#define TWOPI 6.283185307179586476925286766559

namespace maxigpu {
namespace ps {  // per-sample engine

// Device failures (a HIP error, an exhausted allocation, no device at all) have no counterpart in the reference, whose classes never
// fail, never throw and never exit from a per-sample call (its one runtime complaint is printf("ERROR: Could not load sample."),
// src/maximilian.cpp:686, after which play() returns silence).  Same here by default: the FIRST failure prints one line to stderr,
// marks the engine dead, and from then on every per-sample call returns silence (0 / false) without touching the device; the C-ABI's
// own status stays readable through mxg_last_error() / mxg_last_async_error().  (A refused ARGUMENT -- MXG_ERR_INVALID -- is not a device
// failure: it silences the one call, see check().)  A C-ABI call that reports a failure does not unwind
// (check(): the C-ABI validates every pointer it is handed, so whatever follows a failed call fails too, quietly, and results that
// were to be produced stay zeros); a failed ALLOCATION or plan creation (fatal()) unwinds to the public method that was called with an
// internal exception that never leaves this header (MAXIGPU_TRY / MAXIGPU_CATCH).  Two opt-ins: -DMAXIGPU_THROW (the round-4
// behaviour: std::runtime_error out of the failing call, for a host that wants to handle it) and -DMAXIGPU_NO_EXCEPTIONS (a build
// without exception support cannot unwind: the failure prints and abort()s -- choose it only where a dead device should end the
// process).  API misuse the reference lets pass (process() before setup(), a short vector) never throws in any build: it prints
// once and returns silence / false (complain()).
struct DeviceFailure {};
inline bool &dead() {
    static bool d = false;
    return d;
}
[[noreturn]] inline void fatal(const std::string &msg) {
#if defined(MAXIGPU_NO_EXCEPTIONS)
    std::fprintf(stderr, "maxigpu: %s\n", msg.c_str());
    std::abort();
#elif defined(MAXIGPU_THROW)
    throw std::runtime_error(msg);
#else
    if (!dead()) std::fprintf(stderr, "ERROR: maxigpu: %s -- the device path is off, every unit generator returns silence from here on\n", msg.c_str());
    dead() = true;
    throw DeviceFailure{};
#endif
}
#if defined(MAXIGPU_NO_EXCEPTIONS) || defined(MAXIGPU_THROW)
#define MAXIGPU_TRY if (true)
#define MAXIGPU_CATCH(...) else { __VA_ARGS__; }
#else
#define MAXIGPU_TRY try
#define MAXIGPU_CATCH(...) catch (const maxigpu::ps::DeviceFailure &) { __VA_ARGS__; }
#endif
inline void complain(const char *msg) {  // the reference's style (printf("ERROR: ...")), once per message
    static std::vector<const char *> *seen = new std::vector<const char *>;
    for (const char *m : *seen)
        if (m == msg) return;
    seen->push_back(msg);
    std::fprintf(stderr, "ERROR: %s\n", msg);
}
// The status of a C-ABI call; true = it succeeded.  Only a DEVICE failure (a HIP error, no device, an exhausted allocation) turns the
// engine off for the whole process.  MXG_ERR_INVALID -- an argument or a capacity the C-ABI refuses, e.g. maxiTimeStretch::play with
// more overlaps than the renderer's eight live grains, overlaps <= 0 -- is the CALLER's object only: one printed line per call site
// (in the reference's printf("ERROR: ...") style), that object returns silence for the call, every other unit generator plays on
// (the reference plays such calls with no global effect).
inline bool check(int status, const char *what) {
    if (status >= 0) return true;
    if (status == MXG_ERR_INVALID) {
        static std::vector<const char *> *seen = new std::vector<const char *>;
        bool said = false;
        for (const char *m : *seen) said = said || m == what;
        if (!said) {
            seen->push_back(what);
            std::fprintf(stderr, "ERROR: maxigpu: %s: %s -- this call returns silence\n", what, mxg_last_error());
        }
        return false;
    }
#if defined(MAXIGPU_NO_EXCEPTIONS) || defined(MAXIGPU_THROW)
    fatal(std::string(what) + ": " + mxg_last_error());
#else
    if (!dead()) std::fprintf(stderr, "ERROR: maxigpu: %s: %s -- the device path is off, every unit generator returns silence from here on\n", what, mxg_last_error());
    dead() = true;
    return false;
#endif
}

constexpr size_t kMaxBlock = 512;

template <typename T>
struct DevBuf {  // grow-only device array -- with a twin in pinned, device-mapped HOST memory for the renders of a few samples (`mapped`:
                 // the kernel reads its arguments from and writes its results to host memory directly, no copy commands at all)
    T *p = nullptr;  // the array of the current mode
    T *dev = nullptr, *map = nullptr;
    size_t n = 0, mn = 0;
    bool mapped = false;
    T *need(size_t count) {
        if (mapped) {
            if (count > mn) {
                if (map) mxg_host_free(map);
                map = static_cast<T *>(mxg_host_alloc(count * sizeof(T)));
                if (!map) maxigpu::ps::fatal(std::string("mxg_host_alloc: ") + mxg_last_error());
                mn = count;
            }
            return p = map;
        }
        if (count > n) {
            if (dev) mxg_free(dev);
            dev = static_cast<T *>(mxg_malloc(count * sizeof(T)));
            if (!dev) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
            n = count;
        }
        return p = dev;
    }
    void mode(bool m) { mapped = m; p = m ? map : dev; }
    bool holds(const void *q) const {  // q points into the mapped twin
        const char *c = static_cast<const char *>(q), *b = reinterpret_cast<const char *>(map);
        return map && c >= b && c < b + mn * sizeof(T);
    }
    ~DevBuf() {
        if (dev) mxg_free(dev);
        if (map) mxg_host_free(map);
    }
};
template <typename T>
struct PinBuf {  // grow-only pinned host array
    T *p = nullptr;
    size_t n = 0;
    T *need(size_t count) {
        if (count > n) {
            if (p) mxg_host_free(p);
            p = static_cast<T *>(mxg_host_alloc(count * sizeof(T)));
            if (!p) maxigpu::ps::fatal(std::string("mxg_host_alloc: ") + mxg_last_error());
            n = count;
        }
        return p;
    }
    ~PinBuf() { if (p) mxg_host_free(p); }
};

struct Call {  // one per-sample call: which method, with which arguments (compared bit for bit)
    int method = -1;
    const void *key = nullptr;  // objects that can share a launch must agree on it (e.g. the sample buffer)
    double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool same(const Call &o) const { return method == o.method && key == o.key && !std::memcmp(a, o.a, sizeof(a)); }
};

// ---- derived arguments ------------------------------------------------------------------------------------------------------
// "The same arguments as the last call" cannot predict an argument that is itself a signal: `VCF.lores((VCO1out+VCO2out)*0.5, ...)`,
// `mySine.sinewave(440+(myOtherSine.sinewave(1)*100))`.  But such an argument is a small expression of what OTHER objects returned a
// moment ago, and those objects' next outputs are already sitting in their cached blocks.  When an argument stops matching, the pool
// looks for a form  x | x*a | x+b | x*a+b | (x+b)*a | ((x+b)*a)+c | x1+x2 | (x1+x2)*a | x1*x2  over the most recent outputs that
// reproduces it BIT FOR BIT (constants fitted from two observations and rounded to short decimals); a form that has reproduced three
// consecutive calls is used to fill a per-sample argument array for the consumer's next block from the producers' cached blocks.
// Nothing is trusted: every real call is still compared bit for bit with the predicted argument of its sample, a mismatch rewinds
// the object exactly as a changed constant does.  No signal arithmetic happens here -- the expression is only evaluated to GUESS the
// number the patch's own C++ is about to compute.  (Environment MXG_PS_DERIVE=0 turns the guessing off.)
struct Hyp {
    int k = -1, form = 0;     // argument index; 1 x  2 x*a  3 x+b  4 x*a+b  5 (x+b)*a  6 x1+x2  7 (x1+x2)*a  8 x1*x2  9 ((x+b)*a)+c
    uint64_t s1 = 0, s2 = 0;  // producer slot ids
    int64_t d1 = 0, d2 = 0;   // producer call ordinal = consumer call ordinal + d
    double a = 0, b = 0, c = 0;
    int streak = 0;           // consecutive calls reproduced
};

struct Slot {
    Call sig;                      // the prediction the cached block was rendered under (its first sample, where arguments are derived)
    std::vector<double> blk;       // cached outputs
    size_t pos = 0, len = 0;       // blk[pos .. len) not yet served
    size_t nextLen = 1;
    std::vector<double> sd, ed;    // state (doubles) at the block's first sample / after its last
    std::vector<int64_t> si, ei;
    int group = -1;                // index of the asynchronous next-block render this slot is part of
    // derived arguments
    uint64_t id = 0, count = 0;    // count = calls served so far = the ordinal of the next call
    uint64_t lastTick = 0;
    double lastOut = 0, prevOut = 0;  // outputs of calls count - 1, count - 2
    int nOut = 0;                  // how many of them exist (0, 1, 2)
    std::vector<Hyp> hyps;
    std::vector<std::vector<double>> dv;  // dv[k]: per-sample value of argument k over the cached block (empty: the constant sig.a[k])
    double lastArg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool hasLastArg = false;
    uint64_t misses = 0;           // calls that were not served from the cached block (statistics)
    int unstable = 0;              // > 0: an argument NO form explains changed a few calls ago -- blocks of one sample until it settles
    int fitFails[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // consecutive fruitless searches per argument (the search backs off)
    uint64_t fitAgain[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

inline uint64_t &ps_tick() { static uint64_t t = 0; return t; }
inline std::vector<Slot *> &ps_live() { static std::vector<Slot *> *v = new std::vector<Slot *>; return *v; }  // (never destroyed: see pool())
inline Slot *ps_slot(uint64_t id) {
    for (Slot *s : ps_live())
        if (s->id == id) return s;
    return nullptr;
}
inline bool ps_derive_on() {
    static const bool on = [] { const char *e = std::getenv("MXG_PS_DERIVE"); return !(e && e[0] == '0'); }();
    return on;
}
inline bool ps_same_bits(double x, double y) { return !std::memcmp(&x, &y, sizeof(double)); }
// what slot X returned (or is predicted to return) at call ordinal `ord`
inline bool ps_out_at(const Slot &X, uint64_t ord, double &v) {
    const uint64_t base = X.count - X.pos;  // ordinal of blk[0]
    if (X.len > 0 && ord >= base && ord < base + X.len) { v = X.blk[(size_t)(ord - base)]; return true; }
    if (X.nOut >= 1 && ord + 1 == X.count) { v = X.lastOut; return true; }
    if (X.nOut >= 2 && ord + 2 == X.count) { v = X.prevOut; return true; }
    return false;
}
inline uint64_t ps_last_ord(const Slot &X) {  // the last ordinal ps_out_at can answer (valid when X.count > 0 or a block is cached)
    const uint64_t base = X.count - X.pos;
    return X.len > 0 ? base + X.len - 1 : X.count - 1;
}
// The shortest decimal (<= 12 significant digits) within `rel` of x -- patch constants are literals like 440, 0.5, 10000, and a constant
// recovered from two observations carries their rounding noise.  false: x is not such a number (an argument that is no affine form of
// the producer: nothing is fitted, nothing is paid).
inline bool ps_nice(double x, double rel, double &y) {
    if (!(x == x) || std::isinf(x)) return false;
    if (x == 0.0) { y = 0.0; return true; }
    for (int digits = 3; digits <= 12; digits += 3) {
        char buf[40];
        std::snprintf(buf, sizeof buf, "%.*g", digits, x);
        y = std::strtod(buf, nullptr);
        if (std::fabs(y - x) <= rel * std::fabs(x)) return true;
    }
    return false;
}
inline double ps_eval(const Hyp &h, double x1, double x2) {  // (statement by statement: the roundings of the patch's own expression)
    double t;
    switch (h.form) {
        case 1: return x1;
        case 2: t = x1 * h.a; return t;
        case 3: t = x1 + h.b; return t;
        case 4: t = x1 * h.a; t = t + h.b; return t;
        case 5: t = x1 + h.b; t = t * h.a; return t;
        case 6: t = x1 + x2; return t;
        case 7: t = x1 + x2; t = t * h.a; return t;
        case 8: t = x1 * x2; return t;
        case 9: t = x1 + h.b; t = t * h.a; t = t + h.c; return t;
    }
    return 0.0;
}
inline bool ps_predict(const Hyp &h, uint64_t ord, double &v) {
    const Slot *X1 = ps_slot(h.s1);
    double x1 = 0, x2 = 0;
    if (!X1 || !ps_out_at(*X1, (uint64_t)((int64_t)ord + h.d1), x1)) return false;
    if (h.form == 6 || h.form == 7 || h.form == 8) {
        const Slot *X2 = ps_slot(h.s2);
        if (!X2 || !ps_out_at(*X2, (uint64_t)((int64_t)ord + h.d2), x2)) return false;
    }
    v = ps_eval(h, x1, x2);
    return true;
}

// One render of L samples for a set of slots, kept alive so the next block can continue on the device.
struct Group {
    std::vector<Slot *> m;
    std::vector<Call> sig;
    size_t L = 0;
    bool pending = false;  // an asynchronous render of the NEXT block is in flight
    void *event = nullptr;
    DevBuf<double> d_state, d_par, d_in, d_out;
    DevBuf<int64_t> d_istate, d_ipar;
    DevBuf<int32_t> d_trig;
    PinBuf<double> h_out, h_state;
    PinBuf<int64_t> h_istate;
    PinBuf<unsigned char> h_stage;  // pinned staging of one enqueue's parameter uploads (asynchronous copies: no host wait per array)
    size_t stage_off = 0;
    bool zc = false;  // this render is ZERO-COPY: state, arguments and results live in mapped host memory (renders of a few samples:
                      // a launch and a stream wait instead of five to eight copy commands around them)
    void set_mode(bool m) {
        zc = m;
        d_state.mode(m); d_par.mode(m); d_in.mode(m); d_out.mode(m);
        d_istate.mode(m); d_ipar.mode(m); d_trig.mode(m);
    }
    bool in_mapped(const void *q) const {
        return d_state.holds(q) || d_par.holds(q) || d_in.holds(q) || d_out.holds(q) || d_istate.holds(q) || d_ipar.holds(q) || d_trig.holds(q);
    }
    bool restart = false;  // this render starts at the members' BLOCK START state again (a rewind): pools whose state is partly on the
                           // device (the delay line's memory) undo the block they rendered last before they render
    std::vector<std::vector<std::vector<double>>> dv;  // [member][argument][sample]: derived arguments of this render (empty: constant)
    double arg(size_t j, int k, size_t t) const {       // argument k of member j at sample t of the block
        if (j < dv.size() && (size_t)k < dv[j].size() && !dv[j][(size_t)k].empty()) return dv[j][(size_t)k][t];
        return sig[j].a[k];
    }
    bool varies(int k) const {
        for (const auto &m : dv)
            if ((size_t)k < m.size() && !m[(size_t)k].empty()) return true;
        return false;
    }
    bool any_derived() const {
        for (const auto &m : dv)
            for (const auto &a : m)
                if (!a.empty()) return true;
        return false;
    }
    ~Group() { if (event) mxg_event_destroy(event); }
};

class Pool {
public:
    Pool(int nD_, int nI_) : nD(nD_), nI(nI_) {}
    virtual ~Pool() {
        for (Group *g : groups) delete g;
        if (stream) mxg_stream_destroy(stream);
    }
    void attach(Slot &s) {
        s.sd.assign(nD, 0.0);
        s.ed.assign(nD, 0.0);
        s.si.assign(nI, 0);
        s.ei.assign(nI, 0);
        slots.push_back(&s);
        static uint64_t next_id = 0;
        s.id = ++next_id;
        ps_live().push_back(&s);
    }
    void detach(Slot &s) {
        leave_group(s);
        slots.erase(std::remove(slots.begin(), slots.end(), &s), slots.end());
        auto &live = ps_live();
        live.erase(std::remove(live.begin(), live.end(), &s), live.end());  // (forms that name it stop evaluating: ps_slot fails)
    }
    // the object's state at its current sample, cached block dropped (before the host edits or reads state)
    void settle(Slot &s) {
        leave_group(s);
        if (s.pos < s.len) {
            if (s.pos > 0 && !dead()) {
                MAXIGPU_TRY { advance(s, s.pos); }
                MAXIGPU_CATCH((void)0)
            }
        } else if (s.len > 0) {
            s.sd = s.ed;
            s.si = s.ei;
        }
        s.pos = s.len = 0;
        s.nextLen = 1;
        s.sig.method = -1;
        s.dv.clear();
    }
    // forget the cached block WITHOUT computing the state at the current sample (the object is going away, or its state is about
    // to be overwritten anyway): no launch -- in particular none from a destructor that runs during static destruction
    void discard(Slot &s) {
        leave_group(s);
        s.pos = s.len = 0;
        s.nextLen = 1;
        s.sig.method = -1;
        s.dv.clear();
    }
    double call(Slot &s, const Call &c) {
        double r;
        if (s.pos < s.len && matches(s, c)) {
            r = s.blk[s.pos++];
            // a hit on a block with DERIVED arguments: the arguments move from sample to sample, and learn() (which runs on misses only)
            // pairs `lastArg` with the producers' outputs one call ago -- so it has to be this call's, not the last miss's
            if (!s.dv.empty()) {
                std::memcpy(s.lastArg, c.a, sizeof(s.lastArg));
                s.hasLastArg = true;
            }
        } else if (dead()) {
            r = 0.0;  // (a device failure was reported once: silence, as after the reference's "ERROR: Could not load sample.")
        } else {
            MAXIGPU_TRY { r = miss(s, c); }
            MAXIGPU_CATCH(r = 0.0)
        }
        s.prevOut = s.lastOut;
        s.lastOut = r;
        if (s.nOut < 2) s.nOut++;
        s.count++;
        s.lastTick = ++ps_tick();
        return r;
    }
    size_t launches = 0, async_hits = 0, derived_blocks = 0;  // statistics (tests)

protected:
    // bit mask of the argument indices of `method` that may be DERIVED (predicted per sample from other objects' outputs): the ones
    // this pool's enqueue() reads through Group::arg(j, k, t)
    virtual unsigned derivable(int /*method*/) const { return 0; }
    virtual bool can_prefetch() const { return true; }  // (false: a render mutates device memory that a rewind must be able to undo)

private:
    // blocks grow 1 -> 8 -> 64 -> 512 while the prediction holds: a longer block costs the GPU nothing more than a short one (the
    // launch and the round trip are the cost), a failed prediction costs one re-run of the samples already served whatever the length
    static constexpr size_t kGrow = 8;
    static size_t grow(size_t len) { return std::min(kGrow * std::max<size_t>(len, 1), kMaxBlock); }
    // the cached block was rendered for exactly this call at its current sample
    static bool matches(const Slot &s, const Call &c) {
        if (s.dv.empty()) return s.sig.same(c);
        if (c.method != s.sig.method || c.key != s.sig.key) return false;
        for (size_t k = 0; k < 10; k++) {
            const double want = (k < s.dv.size() && !s.dv[k].empty()) ? s.dv[k][s.pos] : s.sig.a[k];
            if (!ps_same_bits(want, c.a[k])) return false;
        }
        return true;
    }
    // ---- learning the forms (called on a miss, before the call is served: its ordinal is s.count) ----------------------------------
    void learn(Slot &s, const Call &c) {
        const unsigned mask = ps_derive_on() ? derivable(c.method) : 0u;
        const uint64_t o = s.count;
        for (int k = 0; k < 10; k++) {
            if (!(mask >> k & 1u)) continue;
            const double v = c.a[k];
            bool have = false;
            for (size_t i = 0; i < s.hyps.size();) {  // forms on trial and forms in use: reproduce this call or go
                Hyp &h = s.hyps[i];
                if (h.k != k) { i++; continue; }
                double pv;
                if (ps_predict(h, o, pv) && ps_same_bits(pv, v)) {
                    h.streak++;
                    have = true;
                    i++;
                } else {
                    s.hyps.erase(s.hyps.begin() + (long)i);
                }
            }
            if (!have && s.hasLastArg && !ps_same_bits(v, s.lastArg[k]) && o >= s.fitAgain[k]) {
                if (propose(s, k, o, v)) {
                    s.fitFails[k] = 0;
                } else {  // nothing fits (a non-linear map, a random draw): look again after 4, 8, ... 256 calls
                    s.fitFails[k] = std::min(s.fitFails[k] + 1, 7);
                    s.fitAgain[k] = o + ((uint64_t)2 << s.fitFails[k]);
                }
            }
        }
        // forms of arguments that are not derivable for this method (the object switched methods): drop
        for (size_t i = 0; i < s.hyps.size();)
            if (!(mask >> s.hyps[i].k & 1u)) s.hyps.erase(s.hyps.begin() + (long)i); else i++;
        // an argument that changed and that no form in use (three calls reproduced) explains: a block longer than one sample would
        // only be rendered to be rewound at its second call (10.Filters: a derivable input beside a cutoff that follows an envelope)
        if (s.unstable > 0) s.unstable--;
        if (s.hasLastArg)
            for (int k = 0; k < 10; k++) {
                if (ps_same_bits(c.a[k], s.lastArg[k])) continue;
                bool explained = false;
                for (const Hyp &h : s.hyps) explained = explained || (h.k == k && h.streak >= 3);
                if (!explained) s.unstable = 8;
            }
        std::memcpy(s.lastArg, c.a, sizeof(s.lastArg));
        s.hasLastArg = true;
    }
    bool propose(Slot &s, int k, uint64_t o, double v) {
        // the objects that returned a value most recently (this sample's graph evaluation), newest first
        std::vector<Slot *> recent;
        for (Slot *X : ps_live())
            if (X != &s && X->nOut >= 1 && ps_tick() - X->lastTick < 64) recent.push_back(X);
        std::sort(recent.begin(), recent.end(), [](const Slot *p, const Slot *q) { return p->lastTick > q->lastTick; });
        if (recent.size() > 12) recent.resize(12);
        const double vp = s.lastArg[k];  // the argument one call ago (s.hasLastArg)
        size_t added = 0;
        auto push = [&](Hyp h) {
            if (added >= 8) return;
            h.k = k;
            h.streak = 1;
            s.hyps.push_back(h);
            added++;
        };
        for (Slot *X : recent) {
            const double x = X->lastOut;
            Hyp h;
            h.s1 = X->id;
            h.d1 = (int64_t)(X->count - 1) - (int64_t)o;
            if (ps_same_bits(x, v)) { h.form = 1; push(h); continue; }
            double xp;
            if (!(o >= 1 && ps_out_at(*X, (uint64_t)((int64_t)(o - 1) + h.d1), xp)) || xp == x || v == vp) continue;
            auto fits = [&](const Hyp &t) { return ps_same_bits(ps_eval(t, x, 0), v) && ps_same_bits(ps_eval(t, xp, 0), vp); };
            // constants recovered from the two observations carry their rounding noise, amplified by the cancellation in the differences
            const double amp = 8.0 * DBL_EPSILON * (1.0 + (std::fabs(v) + std::fabs(vp)) / std::fabs(v - vp) + (std::fabs(x) + std::fabs(xp)) / std::fabs(x - xp));
            const double tol = std::min(amp, 1e-6);
            // a constant: the short decimal near the estimate if there is one (a literal), else the estimate itself and its
            // neighbours (a variable of the patch, e.g. a pitch from mtof) -- whichever reproduces BOTH observations
            auto settle = [&](double est, double &slot) {
                double cand[6];
                int nc = 0;
                if (ps_nice(est, tol, cand[nc])) nc++;
                const double up = std::nextafter(est, HUGE_VAL), dn = std::nextafter(est, -HUGE_VAL);
                cand[nc++] = est;
                cand[nc++] = up;
                cand[nc++] = dn;
                cand[nc++] = std::nextafter(up, HUGE_VAL);
                cand[nc++] = std::nextafter(dn, -HUGE_VAL);
                for (int i = 0; i < nc; i++) {
                    slot = cand[i];
                    if (fits(h)) return true;
                }
                return false;
            };
            if (x != 0.0) { h.form = 2; if (settle(v / x, h.a)) { push(h); continue; } }
            h.form = 3; if (settle(v - x, h.b)) { push(h); continue; }
            double slope;
            if (!ps_nice((v - vp) / (x - xp), tol, slope) || slope == 0.0) continue;  // (a multiplier is a literal)
            h.a = slope;
            h.form = 4; if (settle(v - x * slope, h.b)) { push(h); continue; }
            h.form = 5; if (settle(v / slope - x, h.b)) { push(h); continue; }
            h.form = 9;  // ((x + b) * a) + c with a small integer b (15.polysynth: 250 + ((pitch + lfo) * 1000))
            bool got = false;
            for (int b = -12; b <= 12 && !got; b++) {
                if (b == 0) continue;
                h.b = (double)b;
                got = settle(v - (x + h.b) * slope, h.c);
            }
            if (got) push(h);
        }
        // two sources: the sum or the product of two recent outputs, optionally scaled
        for (size_t i = 0; i < recent.size() && i < 4; i++)
            for (size_t j = i + 1; j < recent.size() && j < 4; j++) {
                Slot *X1 = recent[i], *X2 = recent[j];
                Hyp h;
                h.s1 = X1->id; h.d1 = (int64_t)(X1->count - 1) - (int64_t)o;
                h.s2 = X2->id; h.d2 = (int64_t)(X2->count - 1) - (int64_t)o;
                const double x1 = X1->lastOut, x2 = X2->lastOut;
                h.form = 6; if (ps_same_bits(ps_eval(h, x1, x2), v)) { push(h); continue; }
                h.form = 8; if (ps_same_bits(ps_eval(h, x1, x2), v)) { push(h); continue; }
                if (x1 + x2 != 0.0 && ps_nice(v / (x1 + x2), 8.0 * DBL_EPSILON, h.a)) {
                    h.form = 7;
                    if (ps_same_bits(ps_eval(h, x1, x2), v)) push(h);
                }
            }
        return added > 0;
    }
    // the forms in use for the block that starts at ordinal o: per derivable argument the longest-standing one that has reproduced
    // three calls in a row; `span` = how many samples from o on every one of them can be evaluated
    bool derive_for(const Slot &s, unsigned mask, uint64_t o, std::vector<const Hyp *> &use, size_t &span) const {
        use.assign(10, nullptr);
        span = kMaxBlock;
        bool any = false;
        for (const Hyp &h : s.hyps) {
            if (h.streak < 3 || !(mask >> h.k & 1u)) continue;
            if (use[(size_t)h.k] && use[(size_t)h.k]->streak >= h.streak) continue;
            use[(size_t)h.k] = &h;
        }
        for (int k = 0; k < 10; k++) {
            const Hyp *h = use[(size_t)k];
            if (!h) continue;
            const Slot *X1 = ps_slot(h->s1), *X2 = (h->form >= 6 && h->form <= 8) ? ps_slot(h->s2) : nullptr;
            if (!X1 || ((h->form >= 6 && h->form <= 8) && !X2)) { use[(size_t)k] = nullptr; continue; }
            auto room = [&](const Slot &X, int64_t d) -> int64_t { return (int64_t)ps_last_ord(X) - ((int64_t)o + d) + 1; };
            int64_t r = room(*X1, h->d1);
            if (X2) r = std::min(r, room(*X2, h->d2));
            if (r < 1) { use[(size_t)k] = nullptr; continue; }
            span = std::min(span, (size_t)r);
            any = true;
        }
        return any;
    }
    static void fill_derived(const std::vector<const Hyp *> &use, uint64_t o, size_t L, std::vector<std::vector<double>> &dv) {
        dv.assign(10, std::vector<double>());
        for (int k = 0; k < 10; k++) {
            if (!use[(size_t)k]) continue;
            dv[(size_t)k].resize(L);
            for (size_t t = 0; t < L; t++) ps_predict(*use[(size_t)k], o + t, dv[(size_t)k][t]);
        }
    }

public:

protected:
    // Enqueue on `stream` the render of G.L samples for G.m under G.sig, starting from the state in G.d_state /
    // G.d_istate ([nD][n] / [nI][n], already on the device), updating that state in place and writing d_out [L][n].
    virtual void enqueue(Group &G) = 0;
    int nD, nI;
    void *stream = nullptr;
    // parameter uploads of one enqueue: copied into the group's pinned staging area and sent asynchronously on the pool's stream
    // (a synchronous copy per array cost one host wait each: four of them per filter call).  The staging area is free again
    // when the render that used it has been waited for -- which every caller of enqueue does before it enqueues for that group again.
    void stage_begin(Group &G, size_t bytes) {
        G.h_stage.need(bytes + 256);
        G.stage_off = 0;
    }
    void put(Group &G, void *d_dst, const void *h_src, size_t bytes, const char *what) {
        if (G.zc && G.in_mapped(d_dst)) {  // the destination IS host memory: written here, read by the kernel over the bus
            std::memcpy(d_dst, h_src, bytes);
            return;
        }
        std::memcpy(G.h_stage.p + G.stage_off, h_src, bytes);
        check(mxg_memcpy_h2d_async(d_dst, G.h_stage.p + G.stage_off, bytes, stream), what);
        G.stage_off += (bytes + 15) & ~(size_t)15;
    }

private:
    std::vector<Slot *> slots;
    std::vector<Group *> groups;

    void ensure_stream() {
        if (!stream) {
            check(mxg_init(-1), "mxg_init");
            stream = mxg_stream_create();
            if (!stream) maxigpu::ps::fatal(std::string("mxg_stream_create: ") + mxg_last_error());
        }
    }
    void leave_group(Slot &s) {
        if (s.group < 0) return;
        Group &G = *groups[(size_t)s.group];
        for (Slot *&u : G.m)
            if (u == &s) u = nullptr;  // its column of the pending block is simply never installed
        s.group = -1;
    }
    Group &free_group() {
        for (size_t i = 0; i < groups.size(); i++) {
            Group &G = *groups[i];
            bool used = G.pending;
            if (used) {  // a pending group nobody waits for any more can be recycled once its render has finished
                bool any = false;
                for (Slot *u : G.m) any = any || u;
                if (!any && mxg_event_query(G.event) == 1) used = G.pending = false;
            }
            if (!used) return G;
        }
        groups.push_back(new Group());
        groups.back()->event = mxg_event_create();
        return *groups.back();
    }
    int index_of(const Group &G) const {
        for (size_t i = 0; i < groups.size(); i++)
            if (groups[i] == &G) return (int)i;
        return -1;
    }
    // upload the members' start states and run one block synchronously; fills blk / ed / ei of every member
    static constexpr size_t kZeroCopy = 64;  // outputs (samples x members) up to which a render runs on mapped host memory
    static bool zero_copy_on() {
        static const bool on = [] { const char *e = std::getenv("MXG_PS_ZEROCOPY"); return !(e && e[0] == '0'); }();
        return on;
    }
    void render_now(Group &G) {
        ensure_stream();
        const size_t n = G.m.size(), L = G.L;
        G.set_mode(zero_copy_on() && L * n <= kZeroCopy);
        double *hs = G.zc ? G.d_state.need((size_t)nD * n + 1) : G.h_state.need((size_t)nD * n + 1);
        int64_t *hi = G.zc ? G.d_istate.need((size_t)nI * n + 1) : G.h_istate.need((size_t)nI * n + 1);
        for (size_t j = 0; j < n; j++) {
            for (int k = 0; k < nD; k++) hs[(size_t)k * n + j] = G.m[j]->sd[(size_t)k];
            for (int k = 0; k < nI; k++) hi[(size_t)k * n + j] = G.m[j]->si[(size_t)k];
        }
        if (!G.zc) {
            if (nD) check(mxg_memcpy_h2d_async(G.d_state.need((size_t)nD * n), hs, sizeof(double) * nD * n, stream), "h2d state");
            if (nI) check(mxg_memcpy_h2d_async(G.d_istate.need((size_t)nI * n), hi, sizeof(int64_t) * nI * n, stream), "h2d istate");
        }
        G.d_out.need(L * n);
        enqueue(G);
        if (!G.zc) fetch(G);
        check(mxg_stream_sync(stream), "mxg_stream_sync");
        install(G);
        launches++;
    }
    void fetch(Group &G) {  // asynchronous download of the block and of the state after it
        const size_t n = G.m.size();
        check(mxg_memcpy_d2h_async(G.h_out.need(G.L * n), G.d_out.p, sizeof(double) * G.L * n, stream), "d2h out");
        if (nD) check(mxg_memcpy_d2h_async(G.h_state.need((size_t)nD * n + 1), G.d_state.p, sizeof(double) * nD * n, stream), "d2h state");
        if (nI) check(mxg_memcpy_d2h_async(G.h_istate.need((size_t)nI * n + 1), G.d_istate.p, sizeof(int64_t) * nI * n, stream), "d2h istate");
    }
    void install(Group &G) {
        const size_t n = G.m.size(), L = G.L;
        const double *ho = G.zc ? G.d_out.p : G.h_out.p, *hsd = G.zc ? G.d_state.p : G.h_state.p;
        const int64_t *hsi = G.zc ? G.d_istate.p : G.h_istate.p;
        for (size_t j = 0; j < n; j++) {
            Slot *u = G.m[j];
            if (!u) continue;
            u->blk.resize(L);
            for (size_t t = 0; t < L; t++) u->blk[t] = ho[t * n + j];
            for (int k = 0; k < nD; k++) u->ed[(size_t)k] = hsd[(size_t)k * n + j];
            for (int k = 0; k < nI; k++) u->ei[(size_t)k] = hsi[(size_t)k * n + j];
            u->pos = 0;
            u->len = L;
            u->sig = G.sig[j];
            u->nextLen = L;
            if (j < G.dv.size()) u->dv = G.dv[j]; else u->dv.clear();
            bool any = false;
            for (const auto &a : u->dv) any = any || !a.empty();
            if (!any) u->dv.clear();
        }
    }
    // the block after the one just installed, rendered while that one is served
    void prefetch(Group &G) {
        enqueue(G);  // continues from the state the previous block left on the device
        fetch(G);
        check(mxg_event_record(G.event, stream), "mxg_event_record");
        G.pending = true;
        const int gi = index_of(G);
        for (Slot *u : G.m)
            if (u) u->group = gi;
        launches++;
    }
    // re-run the first `count` samples of the slot's block from its start state: the state AT sample `count`
    void advance(Slot &s, size_t count) {
        Group &G = free_group();
        G.m.assign(1, &s);
        G.sig.assign(1, s.sig);
        G.dv.assign(1, s.dv);  // the same (verified) derived arguments for the samples re-run
        G.restart = true;
        G.L = count;
        const std::vector<double> keep_ed = s.ed;
        render_now(G);
        s.sd = s.ed;
        s.si = s.ei;
        (void)keep_ed;
        s.group = -1;
    }
    double miss(Slot &s, const Call &c) {
        s.misses++;
        const bool consumed = s.pos >= s.len;
        if (consumed && s.group >= 0 && s.sig.same(c)) {  // the asynchronous next block was rendered for exactly this call
            Group &G = *groups[(size_t)s.group];
            check(mxg_event_sync(G.event), "mxg_event_sync");  // normally long complete
            G.pending = false;
            bool intact = true;
            for (size_t j = 0; j < G.m.size(); j++) {
                Slot *u = G.m[j];
                if (u && u->pos < u->len) {  // not in lock-step after all: that member keeps its own block
                    u->group = -1;
                    G.m[j] = u = nullptr;
                }
                if (!u) {
                    intact = false;
                    continue;
                }
                u->sd = u->ed;
                u->si = u->ei;
                u->group = -1;
            }
            install(G);
            async_hits++;
            if (intact) prefetch(G);
            return s.blk[s.pos++];
        }
        leave_group(s);
        learn(s, c);
        const unsigned mask = ps_derive_on() ? derivable(c.method) : 0u;
        const bool was_derived = !s.dv.empty();
        size_t L;
        bool rewound_at_start = false;
        if (!consumed) {  // the prediction failed inside a block: back to the state at this sample
            if (s.pos > 0) advance(s, s.pos); else rewound_at_start = true;  // (at its first sample: the new render starts there again)
            L = 1;
        } else {
            if (s.len > 0) {
                s.sd = s.ed;
                s.si = s.ei;
            }
            L = (s.len > 0 && (s.sig.same(c) || was_derived)) ? grow(s.nextLen) : 1;
        }
        // derived arguments for the new block, as far as the producers' cached blocks reach
        std::vector<const Hyp *> use;
        size_t span = 0;
        std::vector<std::vector<double>> dv0;
        const bool derived = mask && s.unstable == 0 && derive_for(s, mask, s.count, use, span);
        if (derived) {
            if (consumed && L < kGrow) L = kGrow;  // the forms have just reproduced three calls: start growing
            L = std::min(L, span);
            fill_derived(use, s.count, L, dv0);
            derived_blocks++;
        } else if (was_derived && !s.sig.same(c)) {
            L = 1;
        }
        s.pos = s.len = 0;
        s.dv.clear();
        Group &G = free_group();
        G.m.assign(1, &s);
        G.sig.assign(1, c);
        G.dv.assign(1, dv0);
        G.restart = rewound_at_start;
        G.L = L;
        // objects called in lock-step with this one: same method, also at the end of their block, same growth
        for (Slot *u : slots) {
            if (u == &s || u->group >= 0 || u->len == 0 || u->pos < u->len) continue;
            if (u->sig.method != c.method || u->sig.key != c.key) continue;
            if (!u->dv.empty() || !u->hyps.empty()) {  // a member with derived arguments: its own forms, evaluated at ITS next ordinal
                std::vector<const Hyp *> uu;
                size_t uspan = 0;
                if (!derive_for(*u, mask, u->count, uu, uspan)) continue;
                if (std::min(std::max<size_t>(grow(u->nextLen), kGrow), uspan) != L) continue;
                std::vector<std::vector<double>> udv;
                fill_derived(uu, u->count, L, udv);
                Call uc = u->sig;
                for (size_t k = 0; k < 10; k++)
                    if (!udv[k].empty()) uc.a[k] = udv[k][0];
                u->sd = u->ed;
                u->si = u->ei;
                G.m.push_back(u);
                G.sig.push_back(uc);
                G.dv.push_back(udv);
                continue;
            }
            if (grow(u->nextLen) != L) continue;
            u->sd = u->ed;
            u->si = u->ei;
            G.m.push_back(u);
            G.sig.push_back(u->sig);
            G.dv.push_back(std::vector<std::vector<double>>());
        }
        render_now(G);
        if (L == kMaxBlock && !G.any_derived() && can_prefetch()) prefetch(G);
        return s.blk[s.pos++];
    }
};

template <typename P>
P &pool() {
    static P *p = new P;  // never destroyed: objects with static storage outlive any order the runtime tears things down in
    return *p;
}

// ---- the pools -------------------------------------------------------------------------------------------
// A public DATA member of a reference class whose value lives in the slot's state (the device's, between blocks): a read or a
// write first settles the slot -- the state at the object's current sample, cached block dropped -- like the class's own setters.
// (Reading such a member every sample costs the block cache: a patch that does it runs one launch per sample.)
template <class PoolT, typename T, bool IS_INT, int IDX>
class StateMember {
    Slot *s_;
    T get() const {
        pool<PoolT>().settle(*s_);
        return IS_INT ? (T)s_->si[IDX] : (T)s_->sd[IDX];
    }
    void put(T v) {
        pool<PoolT>().settle(*s_);
        if (IS_INT) s_->si[IDX] = (int64_t)v; else s_->sd[IDX] = (double)v;
    }

public:
    explicit StateMember(Slot &s) : s_(&s) {}
    StateMember(const StateMember &) = delete;  // (bound to ONE object's slot: the owner's copy operations copy the state itself)
    operator T() const { return get(); }
    StateMember &operator=(T v) { put(v); return *this; }
    StateMember &operator=(const StateMember &o) { put(o.get()); return *this; }
    StateMember &operator+=(T v) { put(get() + v); return *this; }
    StateMember &operator-=(T v) { put(get() - v); return *this; }
    StateMember &operator*=(T v) { put(get() * v); return *this; }
    StateMember &operator/=(T v) { put(get() / v); return *this; }
    StateMember &operator++() { put(get() + 1); return *this; }
    StateMember &operator--() { put(get() - 1); return *this; }
};

struct OscPool : Pool {  // state: phase, output (H:173,176)
    OscPool() : Pool(2, 0) {}
    unsigned derivable(int method) const override {  // the frequency; pulse's width too (not noise()'s rand() draw)
        return method == 12 ? 0u : (method == MXG_OSC_PULSE ? 3u : 1u);
    }
    void enqueue(Group &G) override {
        const size_t n = G.m.size();
        const int wf = G.sig[0].method;
        std::vector<double> hp(3 * n);
        for (size_t j = 0; j < n; j++)
            for (int k = 0; k < 3; k++) hp[(size_t)k * n + j] = G.sig[j].a[k];
        double *dp = G.d_par.need(3 * n);
        stage_begin(G, sizeof(double) * 3 * n + (sizeof(int32_t) + 2 * sizeof(double)) * G.L * n + 64);
        put(G, dp, hp.data(), sizeof(double) * 3 * n, "h2d osc");
        if (wf != 12 && (G.varies(0) || G.varies(1))) {  // a derived frequency (/ pulse width): one value per sample, [L][n] (the
            // kernel's per-sample form: the same `phase += 1./(sampleRate/frequency)` with this sample's frequency, as a call with
            // that argument computes)
            const bool w = G.varies(1);
            std::vector<double> hf((w ? 2 : 1) * G.L * n);
            for (size_t t = 0; t < G.L; t++)
                for (size_t j = 0; j < n; j++) {
                    hf[t * n + j] = G.arg(j, 0, t);
                    if (w) hf[G.L * n + t * n + j] = G.arg(j, 1, t);
                }
            double *df = G.d_in.need(2 * G.L * n);
            put(G, df, hf.data(), sizeof(double) * hf.size(), "h2d osc freq");
            check(mxg_osc_render(wf, n, G.L, df, w ? 2 : 1, w ? df + G.L * n : dp + n, dp + 2 * n, G.d_state.p, G.d_state.p + n, G.d_out.p,
                                 stream), "mxg_osc_render");
            return;
        }
        if (wf == 12) {  // noise(): a[0] holds the rand() draw
            std::vector<int32_t> ht(G.L * n);
            for (size_t t = 0; t < G.L; t++)
                for (size_t j = 0; j < n; j++) ht[t * n + j] = (int32_t)G.sig[j].a[0];
            put(G, G.d_trig.need(G.L * n), ht.data(), sizeof(int32_t) * G.L * n, "h2d rand");
            check(mxg_osc_noise(n, G.L, G.d_trig.p, G.d_state.p + n, G.d_out.p, stream), "mxg_osc_noise");
            return;
        }
        check(mxg_osc_render(wf, n, G.L, dp, 0, dp + n, dp + 2 * n, G.d_state.p, G.d_state.p + n, G.d_out.p, stream),
              "mxg_osc_render");
    }
};

struct EnvPool : Pool {  // state: amplitude, output | holdcount, attack/decay/sustain/hold/release phase (H:888-932)
    EnvPool() : Pool(2, 6) {}
    unsigned derivable(int) const override { return 1u; }  // the input signal
    void enqueue(Group &G) override {
        const size_t n = G.m.size(), L = G.L;
        const int mode = G.sig[0].method;  // 0 adsr, 1 ar
        std::vector<double> in(L * n), par(4 * n);
        std::vector<int32_t> trig(L * n);
        std::vector<int64_t> hold(n);
        for (size_t j = 0; j < n; j++) {
            const double *a = G.sig[j].a;  // input, trigger, attack, decay, sustain, release, holdtime
            for (size_t t = 0; t < L; t++) {
                in[t * n + j] = G.arg(j, 0, t);
                trig[t * n + j] = (int32_t)a[1];
            }
            for (int k = 0; k < 4; k++) par[(size_t)k * n + j] = a[2 + k];
            hold[j] = (int64_t)a[6];
        }
        stage_begin(G, sizeof(double) * (L * n + 4 * n) + sizeof(int32_t) * L * n + sizeof(int64_t) * n + 64);
        put(G, G.d_in.need(L * n), in.data(), sizeof(double) * L * n, "h2d env in");
        put(G, G.d_trig.need(L * n), trig.data(), sizeof(int32_t) * L * n, "h2d env trig");
        put(G, G.d_par.need(4 * n), par.data(), sizeof(double) * 4 * n, "h2d env par");
        put(G, G.d_ipar.need(n), hold.data(), sizeof(int64_t) * n, "h2d env hold");
        check(mxg_env_render(mode, n, L, G.d_in.p, G.d_trig.p, 1, G.d_par.p, G.d_ipar.p, G.d_state.p, G.d_istate.p, G.d_out.p,
                             stream), "mxg_env_render");
    }
};

struct FilterPool : Pool {  // state: x, y, outputs[0..2] (H:289-302)
    FilterPool() : Pool(5, 0) {}
    unsigned derivable(int) const override { return 7u; }  // the input signal, the cutoff, the resonance
    void enqueue(Group &G) override {
        const size_t n = G.m.size(), L = G.L;
        const int kind = G.sig[0].method;
        std::vector<double> in(L * n), cut(n), res(n), coef(3 * n, 0.0);
        for (size_t j = 0; j < n; j++) {
            const double *a = G.sig[j].a;  // input, cutoff, resonance
            for (size_t t = 0; t < L; t++) in[t * n + j] = G.arg(j, 0, t);
            cut[j] = a[1];
            res[j] = a[2];
        }
        const bool moving = G.varies(1) || G.varies(2);  // a derived cutoff / resonance: one value per sample
        if (moving && kind <= MXG_FLT_BANDPASS) {
            // the coefficients of EVERY sample with the host libm (cos / pow / sqrt of C:459-461, :492-495, as the reference evaluates
            // them on every call), rows [L][3][n]
            std::vector<double> cps(L * 3 * n, 0.0);
            for (size_t t = 0; t < L; t++) {
                for (size_t j = 0; j < n; j++) {
                    cut[j] = G.arg(j, 1, t);
                    res[j] = G.arg(j, 2, t);
                }
                check(mxg_filter_coeffs_host(kind, n, cut.data(), res.data(), cps.data() + t * 3 * n), "mxg_filter_coeffs_host");
            }
            stage_begin(G, sizeof(double) * (L * n + L * 3 * n) + 64);
            put(G, G.d_in.need(L * n), in.data(), sizeof(double) * L * n, "h2d flt in");
            put(G, G.d_par.need(L * 3 * n), cps.data(), sizeof(double) * L * 3 * n, "h2d flt coefs");
            check(mxg_filter_render_coefs(kind, n, L, G.d_in.p, G.d_par.p, G.d_state.p, G.d_out.p, stream), "mxg_filter_render_coefs");
            return;
        }
        if (moving) {  // lopass / hipass use the cutoff itself (C:442-453): the kernel's per-sample form is the same arithmetic
            std::vector<double> cs(L * n);
            for (size_t t = 0; t < L; t++)
                for (size_t j = 0; j < n; j++) cs[t * n + j] = G.arg(j, 1, t);
            stage_begin(G, sizeof(double) * 2 * L * n + 64);
            put(G, G.d_in.need(L * n), in.data(), sizeof(double) * L * n, "h2d flt in");
            put(G, G.d_par.need(L * n), cs.data(), sizeof(double) * L * n, "h2d flt cutoffs");
            check(mxg_filter_render(kind, n, L, G.d_in.p, G.d_par.p, 1, nullptr, 0, nullptr, G.d_state.p, G.d_out.p, stream),
                  "mxg_filter_render");
            return;
        }
        if (kind <= MXG_FLT_BANDPASS)  // cos / pow / sqrt of C:459-461, :492-495 on the host libm
            check(mxg_filter_coeffs_host(kind, n, cut.data(), res.data(), coef.data()), "mxg_filter_coeffs_host");
        double *dp = G.d_par.need(5 * n);
        stage_begin(G, sizeof(double) * (L * n + 5 * n) + 64);
        put(G, G.d_in.need(L * n), in.data(), sizeof(double) * L * n, "h2d flt in");
        put(G, dp, cut.data(), sizeof(double) * n, "h2d flt cutoff");
        put(G, dp + n, res.data(), sizeof(double) * n, "h2d flt res");
        put(G, dp + 2 * n, coef.data(), sizeof(double) * 3 * n, "h2d flt coef");
        check(mxg_filter_render(kind, n, L, G.d_in.p, dp, 0, dp + n, 0, dp + 2 * n, G.d_state.p, G.d_out.p, stream),
              "mxg_filter_render");
    }
};

// state: position, zxTrig.previousValue, phasorPrev | zxTrig.firstTrigger, phasorFirst (H:606, 593-594, 731-732); key = the
// device sample buffer.  Call arguments: a[0] speed / frequency, a[1] a[2] start, end (or offset, length / pos of the playOnZX
// family), a[3] the per-sample first argument of the trigger-driven players (trigger signal, phasor) or the caller's `pos`.
struct SamplePool : Pool {
    SamplePool() : Pool(3, 2) {}
    struct Buf { double *d = nullptr; size_t len = 0; int rate = 44100; };
    static constexpr int kFromPos = 15;  // playAtSpeedBetweenPointsFromPos: a pure function of its arguments
    void enqueue(Group &G) override {
        const size_t n = G.m.size(), L = G.L;
        const Buf *b = static_cast<const Buf *>(G.sig[0].key);
        const int mode = G.sig[0].method;
        std::vector<double> par(3 * n);
        for (size_t j = 0; j < n; j++)
            for (int k = 0; k < 3; k++) par[(size_t)k * n + j] = G.sig[j].a[k];
        double *dp = G.d_par.need(3 * n);
        stage_begin(G, sizeof(double) * (3 * n + L * n) + 64);
        put(G, dp, par.data(), sizeof(double) * 3 * n, "h2d smp");
        if (mode <= MXG_SMP_PLAYATSPEEDBETWEENPOINTS) {
            check(mxg_sample_render(mode, n, L, b->d, b->len, b->rate, dp, 0, dp + n, dp + 2 * n, G.d_state.p, G.d_out.p, stream),
                  "mxg_sample_render");
            return;
        }
        std::vector<double> sig(L * n);  // the first argument, held over the block (the prediction)
        for (size_t j = 0; j < n; j++)
            for (size_t t = 0; t < L; t++) sig[t * n + j] = G.sig[j].a[3];
        put(G, G.d_in.need(L * n), sig.data(), sizeof(double) * L * n, "h2d smp signal");
        if (mode == kFromPos) {
            check(mxg_sample_render_frompos(n, L, b->d, b->len, dp, 0, dp + n, dp + 2 * n, G.d_in.p, G.d_out.p, stream),
                  "mxg_sample_render_frompos");
            return;
        }
        const bool pha = mode == MXG_SMP_PLAYWITHPHASOR;  // its own previous value / first flag (H:731-732)
        int32_t *tf = G.d_trig.need(n);
        int64_t *flag = G.d_istate.p + (pha ? n : 0);
        check(mxg_i32_from_i64(tf, flag, n, stream), "mxg_i32_from_i64");
        check(mxg_sample_render_trig(mode, n, L, b->d, b->len, b->rate, G.d_in.p, dp, 0, dp + n, dp + 2 * n, G.d_state.p,
                                     G.d_state.p + (pha ? 2 : 1) * n, tf, G.d_out.p, stream), "mxg_sample_render_trig");
        check(mxg_i64_from_i32(flag, tf, n, stream), "mxg_i64_from_i32");
    }
};

struct Filter2Pool : Pool {  // maxiDCBlocker / maxiSVF / maxiBiquad: three doubles of state each (mxg_filter2_render)
    Filter2Pool() : Pool(3, 0) {}
    unsigned derivable(int) const override { return 1u; }  // the input signal
    void enqueue(Group &G) override {
        const size_t n = G.m.size(), L = G.L;
        const int kind = G.sig[0].method;                    // 0 DC blocker, 1 SVF, 2 biquad
        const size_t rows = kind == 0 ? 1 : (kind == 1 ? 9 : 5);  // coefficient rows, call arguments a[1 .. rows]
        std::vector<double> in(L * n), coef(rows * n);
        for (size_t j = 0; j < n; j++) {
            for (size_t t = 0; t < L; t++) in[t * n + j] = G.arg(j, 0, t);
            for (size_t k = 0; k < rows; k++) coef[k * n + j] = G.sig[j].a[1 + k];
        }
        stage_begin(G, sizeof(double) * (L * n + rows * n) + 64);
        put(G, G.d_in.need(L * n), in.data(), sizeof(double) * L * n, "h2d filter2 in");
        put(G, G.d_par.need(rows * n), coef.data(), sizeof(double) * rows * n, "h2d filter2 coef");
        check(mxg_filter2_render(kind, n, L, G.d_in.p, G.d_par.p, G.d_state.p, G.d_out.p, stream), "mxg_filter2_render");
    }
};

struct EnvGenPool : Pool {  // state of mxg_envgen_render: [5] doubles, [7] int64; key = the object's stage table
    EnvGenPool() : Pool(5, 7) {}
    struct Shape { double *d_stages = nullptr; int nstages = 0; bool loop = false, retrigger = false; };
    void enqueue(Group &G) override {
        const size_t n = G.m.size(), L = G.L;
        const Shape *sh = static_cast<const Shape *>(G.sig[0].key);
        std::vector<double> trig(L * n);
        for (size_t j = 0; j < n; j++)
            for (size_t t = 0; t < L; t++) trig[t * n + j] = G.sig[j].a[0];
        stage_begin(G, sizeof(double) * L * n + 64);
        put(G, G.d_in.need(L * n), trig.data(), sizeof(double) * L * n, "h2d envgen trig");
        check(mxg_envgen_render(n, L, G.d_in.p, 1, sh->d_stages, sh->nstages, sh->loop, sh->retrigger, G.d_state.p, G.d_istate.p,
                                G.d_out.p, stream), "mxg_envgen_render");
    }
};

// maxiDelayline (H:268-284; C:415-439): the 88200 * 8 doubles of `memory` live on the device, per object (key).  state: phase.
// A render writes the cells it passes; a rewind (Group::restart) first puts back the cells the previous render of that object
// overwrote -- saved, in the order they were first touched, before every render (at most kMaxBlock of them: a block is <= 512
// samples, and a line shorter than that is saved whole) -- so the re-run starts from the memory the block started from.  No
// asynchronous next block (it would write ahead of a block that may still be rewound); one object per launch.
struct DelayPool : Pool {
    DelayPool() : Pool(0, 1) {}
    static constexpr size_t kCap = 88200 * 8;  // double memory[88200 * 8], H:273
    struct Line {
        double *d_mem = nullptr, *d_save = nullptr;
        int32_t *d_i = nullptr;  // size, position, phase
        size_t saved_first = 0, saved_n = 0, saved_wrap = 0;  // the last render's cells: [first, first + n - wrap) then [0, wrap)
        bool saved = false;
    };
    unsigned derivable(int) const override { return 1u; }  // the input signal
    bool can_prefetch() const override { return false; }
    void enqueue(Group &G) override {
        if (G.m.size() != 1) fatal("maxiDelayline: one object per launch");  // (an engine invariant, not a user error)
        Line *ln = const_cast<Line *>(static_cast<const Line *>(G.sig[0].key));
        const size_t L = G.L;
        const double *a = G.sig[0].a;  // input, size, feedback, position
        const int mode = G.sig[0].method;
        int32_t size = (int32_t)a[1];
        if (size < 1 || (size_t)size > kCap) {  // the reference indexes memory[88200 * 8] with it unchecked (C:420-429): out of bounds there,
            complain("maxiDelayline: size outside [1, 705600] -- clamped");  // clamped into the ring here
            size = size < 1 ? 1 : (int32_t)kCap;
        }
        if (G.restart && ln->saved) {  // undo the block rendered last
            const size_t head = ln->saved_n - ln->saved_wrap;
            if (head) check(mxg_memcpy_d2d_async(ln->d_mem + ln->saved_first, ln->d_save, sizeof(double) * head, stream), "d2d restore");
            if (ln->saved_wrap) check(mxg_memcpy_d2d_async(ln->d_mem, ln->d_save + head, sizeof(double) * ln->saved_wrap, stream), "d2d restore");
        }
        // the cells this render passes: from the start phase (0 if it is >= size, C:421) on, wrapping at `size`
        const Slot *u = G.m[0];
        const int64_t ph0 = u ? u->si[0] : 0;
        const size_t first = (ph0 >= size || ph0 < 0) ? 0 : (size_t)ph0;
        const size_t n = std::min(L, (size_t)size);
        const size_t head = std::min(n, (size_t)size - first), wrap = n - head;
        check(mxg_memcpy_d2d_async(ln->d_save, ln->d_mem + first, sizeof(double) * head, stream), "d2d save");
        if (wrap) check(mxg_memcpy_d2d_async(ln->d_save + head, ln->d_mem, sizeof(double) * wrap, stream), "d2d save");
        ln->saved_first = first; ln->saved_n = n; ln->saved_wrap = wrap; ln->saved = true;
        std::vector<double> in(L + 1);
        for (size_t t = 0; t < L; t++) in[t] = G.arg(0, 0, t);
        in[L] = a[2];  // feedback
        const int32_t hi[2] = {size, (int32_t)a[3]};
        stage_begin(G, sizeof(double) * (L + 1) + sizeof(hi) + 64);
        put(G, G.d_in.need(L + 1), in.data(), sizeof(double) * (L + 1), "h2d delay in");
        put(G, ln->d_i, hi, sizeof(hi), "h2d delay size");
        check(mxg_i32_from_i64(ln->d_i + 2, G.d_istate.p, 1, stream), "mxg_i32_from_i64");
        check(mxg_delay_render(mode, 1, L, G.d_in.p, ln->d_i, G.d_in.p + L, ln->d_i + 1, ln->d_mem, kCap, ln->d_i + 2, G.d_out.p, stream),
              "mxg_delay_render");
        check(mxg_i64_from_i32(G.d_istate.p, ln->d_i + 2, 1, stream), "mxg_i64_from_i32");
    }
};

}  // namespace ps
}  // namespace maxigpu

// ---- maxiSettings (H:117-163) -----------------------------------------------------------------------------
class maxiSettings {
public:
    static size_t sampleRate, channels, bufferSize;
    static void setup(size_t initSampleRate, size_t initChannels, size_t initBufferSize) {
        maxigpu::ps::check(mxg_settings(initSampleRate, initChannels, initBufferSize), "mxg_settings");
        sampleRate = initSampleRate;
        channels = initChannels;
        bufferSize = initBufferSize;
    }
    static void setSampleRate(size_t sampleRate_) { setup(sampleRate_, channels, bufferSize); }
    static void setNumChannels(size_t channels_) { setup(sampleRate, channels_, bufferSize); }
    static void setBufferSize(size_t bufferSize_) { setup(sampleRate, channels, bufferSize_); }
    static size_t getSampleRate() { return sampleRate; }
    static size_t getNumChannels() { return channels; }
    static size_t getBufferSize() { return bufferSize; }
};
inline size_t maxiSettings::sampleRate = 44100;
inline size_t maxiSettings::channels = 2;
inline size_t maxiSettings::bufferSize = 1024;

// ---- maxiOsc (H:169-215; C:209-373) -------------------------------------------------------------------------
class maxiOsc {
    using Pool = maxigpu::ps::OscPool;
    maxigpu::ps::Slot slot_;
    double run(int wf, double f, double p1 = 0.0, double p2 = 0.0) {
        maxigpu::ps::Call c;
        c.method = wf;
        c.a[0] = f; c.a[1] = p1; c.a[2] = p2;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }

public:
    maxiOsc() { maxigpu::ps::pool<Pool>().attach(slot_); }
    ~maxiOsc() { maxigpu::ps::pool<Pool>().detach(slot_); }
    maxiOsc(const maxiOsc &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_state(o); }
    maxiOsc &operator=(const maxiOsc &o) { if (this != &o) copy_state(o); return *this; }
    double sinewave(double frequency) { return run(MXG_OSC_SINEWAVE, frequency); }
    double coswave(double frequency) { return run(MXG_OSC_COSWAVE, frequency); }
    double phasor(double frequency) { return run(MXG_OSC_PHASOR, frequency); }
    double phasorBetween(double frequency, double startphase, double endphase) { return run(MXG_OSC_PHASORBETWEEN, frequency, startphase, endphase); }
    double saw(double frequency) { return run(MXG_OSC_SAW, frequency); }
    double triangle(double frequency) { return run(MXG_OSC_TRIANGLE, frequency); }
    double square(double frequency) { return run(MXG_OSC_SQUARE, frequency); }
    double pulse(double frequency, double duty) { return run(MXG_OSC_PULSE, frequency, duty); }
    double impulse(double frequency) { return run(MXG_OSC_IMPULSE, frequency); }
    double sinebuf(double frequency) { return run(MXG_OSC_SINEBUF, frequency); }
    double sinebuf4(double frequency) { return run(MXG_OSC_SINEBUF4, frequency); }
    double sawn(double frequency) { return run(MXG_OSC_SAWN, frequency); }
    double noise() { return run(12, (double)rand()); }  // C:214-220: the process-wide rand() stream, drawn here in call order
    void phaseReset(double phaseIn) {                    // C:222-226
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd[0] = phaseIn;
    }

private:
    void copy_state(const maxiOsc &o) {
        maxigpu::ps::pool<Pool>().settle(const_cast<maxiOsc &>(o).slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd = o.slot_.sd;
    }
};

// ---- maxiEnv (H:888-932; C:1319-1494) ---------------------------------------------------------------------
class maxiEnv {
    using Pool = maxigpu::ps::EnvPool;
    maxigpu::ps::Slot slot_;
    double run(int mode, double in, int trig, double at, double de, double su, double re, long hold) {
        maxigpu::ps::Call c;
        c.method = mode;
        c.a[0] = in; c.a[1] = (double)trig; c.a[2] = at; c.a[3] = de; c.a[4] = su; c.a[5] = re; c.a[6] = (double)hold;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }

public:
    maxiEnv() { maxigpu::ps::pool<Pool>().attach(slot_); }
    ~maxiEnv() { maxigpu::ps::pool<Pool>().detach(slot_); }
    // a plain value type in the reference (H:888-932): a copy is a second envelope that continues from the same state
    maxiEnv(const maxiEnv &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_from(o); }
    maxiEnv &operator=(const maxiEnv &o) { if (this != &o) copy_from(o); return *this; }
    double ar(double input, double attack = 1, double release = 0.9, long holdtime = 1, int trigger = 0) {
        return run(1, input, trigger, attack, 0.0, 0.0, release, holdtime);
    }
    double adsr(double input, double attack = 1, double decay = 0.99, double sustain = 0.125, double release = 0.9,
                long holdtime = 1, int trigger = 0) {
        return run(0, input, trigger, attack, decay, sustain, release, holdtime);
    }
    double adsr(double input, int trigger) { return run(0, input, trigger, attack, decay, sustain, release, holdtime); }
    // the members user code reads and writes (static-storage objects start at 0, like the reference's globals)
    double attack = 0, decay = 0, sustain = 0, release = 0;
    int trigger = 0;
    long holdtime = 1;
    double input = 0;  // (H:895: declared, never written by the class -- the methods' parameter shadows it)
    // the running state (H:896, 901, 916-918): device state behind the slot, readable and writable like the reference's members
    maxigpu::ps::StateMember<Pool, double, false, 0> amplitude{slot_};
    maxigpu::ps::StateMember<Pool, double, false, 1> output{slot_};
    maxigpu::ps::StateMember<Pool, long, true, 0> holdcount{slot_};
    maxigpu::ps::StateMember<Pool, int, true, 1> attackphase{slot_};
    maxigpu::ps::StateMember<Pool, int, true, 2> decayphase{slot_};
    maxigpu::ps::StateMember<Pool, int, true, 3> sustainphase{slot_};
    maxigpu::ps::StateMember<Pool, int, true, 4> holdphase{slot_};
    maxigpu::ps::StateMember<Pool, int, true, 5> releasephase{slot_};
    void setRelease(double releaseMS) { release = mxg_env_coeff_host(2, releaseMS); }  // C:1469-1472
    void setDecay(double decayMS) { decay = mxg_env_coeff_host(1, decayMS); }          // C:1474-1477
    void setAttack(double attackMS) { attack = mxg_env_coeff_host(0, attackMS); }      // C:1479-1482
    void setAttackMS(double attackMS) { attack = mxg_env_coeff_host(3, attackMS); }    // C:1485-1488
    void setSustain(double sustainL) { sustain = sustainL; }                          // C:1491-1494
    int getTrigger() const { return trigger; }
    void setTrigger(int trigger_) { trigger = trigger_; }

private:
    void copy_from(const maxiEnv &o) {
        maxigpu::ps::pool<Pool>().settle(const_cast<maxiEnv &>(o).slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd = o.slot_.sd;
        slot_.si = o.slot_.si;
        attack = o.attack; decay = o.decay; sustain = o.sustain; release = o.release;
        trigger = o.trigger; holdtime = o.holdtime; input = o.input;
    }
};

// ---- maxiMix (H:372-420; C:503-541) ------------------------------------------------------------------------------
// Stateless: one voice, one sample through the bank's bus kernel, whose per-voice bus signals are what the reference leaves in
// two / four / eight (bit-exact, ambisonic's quirks included).  A launch per call -- a panner's input changes every sample by
// nature; banks mix on the device (mxg_mix_bus, mxg_osc_render_mix).
class maxiMix {
    static void run(int channels, double input, double x, double y, double z, double *out) {
        using maxigpu::ps::check;
        // [in, x, y, z | bus 8 | mix 8] in pinned, device-mapped host memory: the kernel reads and writes it over the bus -- a launch
        // and a wait, no copy commands (as the pools' zero-copy renders)
        static maxigpu::ps::PinBuf<double> *d = new maxigpu::ps::PinBuf<double>;  // (never destroyed: static-destruction order)
        double *p = d->need(20);
        p[0] = input; p[1] = x; p[2] = y; p[3] = z;
        check(mxg_mix_bus(channels, 1, 1, p, p + 1, p + 2, p + 3, p + 4, p + 12, nullptr), "mxg_mix_bus");
        check(mxg_stream_sync(nullptr), "mxg_stream_sync");
        for (int c = 0; c < channels; c++) out[c] = p[4 + c];
    }

public:
    void stereo(double input, std::vector<double> &two, double x) { run(2, input, x, 0.0, 0.0, two.data()); }
    void quad(double input, std::vector<double> &four, double x, double y) { run(4, input, x, y, 0.0, four.data()); }
    void ambisonic(double input, std::vector<double> &eight, double x, double y, double z) { run(8, input, x, y, z, eight.data()); }
};

// ---- maxiMap (H:788-854): range mappings a patch evaluates on the host, with the host libm like the reference -----------------------
class maxiMap {
public:
    static double linlin(double val, double inMin, double inMax, double outMin, double outMax) {  // H:801-805
        val = max(min(val, inMax), inMin);
        return ((val - inMin) / (inMax - inMin) * (outMax - outMin)) + outMin;
    }
    static double linexp(double val, double inMin, double inMax, double outMin, double outMax) {  // H:815-820
        val = max(min(val, inMax), inMin);
        return pow((outMax / outMin), (val - inMin) / (inMax - inMin)) * outMin;
    }
    static double explin(double val, double inMin, double inMax, double outMin, double outMax) {  // H:830-835
        val = max(min(val, inMax), inMin);
        return (log(val / inMin) / log(inMax / inMin) * (outMax - outMin)) + outMin;
    }
    static double clamp(double v, const double low, const double high) { return v > high ? high : (v < low ? low : v); }  // H:843-854
};

// ---- maxiConvert (H:937-962) -- conversions a patch does on the host, as the reference does -----------------------
class maxiConvert {
public:
    static double mtof(int midinote) { return mxg_mtof_host(midinote); }  // C:1498-1500 (table lookup)
    static size_t msToSamps(double timeMs) { return static_cast<size_t>(timeMs / 1000.0 * maxiSettings::sampleRate); }
    static double sampsToMs(size_t samples) { return samples / maxiSettings::sampleRate * 1000.0; }  // (integer division first, H:950)
    static double ampToDbs(double amp) { return std::log10(amp) * 20.0; }
    static double dbsToAmp(double dbs) { return std::pow(10.0, dbs * 0.05); }
};
using convert = maxiConvert;  // H:964

// ---- maxiFilter (H:289-366; C:442-500) ---------------------------------------------------------------------
class maxiFilter {
    using Pool = maxigpu::ps::FilterPool;
    maxigpu::ps::Slot slot_;
    double run(int kind, double in, double cut, double res) {
        maxigpu::ps::Call c;
        c.method = kind;
        c.a[0] = in; c.a[1] = cut; c.a[2] = res;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }

public:
    maxiFilter() { maxigpu::ps::pool<Pool>().attach(slot_); }
    ~maxiFilter() { maxigpu::ps::pool<Pool>().detach(slot_); }
    maxiFilter(const maxiFilter &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_from(o); }  // (a value type in the reference, H:289-366)
    maxiFilter &operator=(const maxiFilter &o) { if (this != &o) copy_from(o); return *this; }
    double cutoff = 0, resonance = 0;
    // (the member `cutoff` receives the clamped cutoff1, C:456-458, 472-474, 488-489; `resonance` is shadowed by the parameter)
    double lores(double input, double cutoff1, double resonance_) {
        cutoff = cutoff1 < 10 ? 10 : (cutoff1 > (double)maxiSettings::sampleRate ? (double)maxiSettings::sampleRate : cutoff1);
        return run(MXG_FLT_LORES, input, cutoff1, resonance_);
    }
    double hires(double input, double cutoff1, double resonance_) {
        cutoff = cutoff1 < 10 ? 10 : (cutoff1 > (double)maxiSettings::sampleRate ? (double)maxiSettings::sampleRate : cutoff1);
        return run(MXG_FLT_HIRES, input, cutoff1, resonance_);
    }
    double bandpass(double input, double cutoff1, double resonance_) {
        cutoff = cutoff1 > (maxiSettings::sampleRate * 0.5) ? (maxiSettings::sampleRate * 0.5) : cutoff1;
        return run(MXG_FLT_BANDPASS, input, cutoff1, resonance_);
    }
    double lopass(double input, double cutoff_) { return run(MXG_FLT_LOPASS, input, cutoff_, 0.0); }
    double hipass(double input, double cutoff_) { return run(MXG_FLT_HIPASS, input, cutoff_, 0.0); }
    void setCutoff(double cut) { cutoff = cut; }
    void setResonance(double res) { resonance = res; }
    double getCutoff() const { return cutoff; }
    double getResonance() const { return resonance; }

private:
    void copy_from(const maxiFilter &o) {
        maxigpu::ps::pool<Pool>().settle(const_cast<maxiFilter &>(o).slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd = o.slot_.sd;
        cutoff = o.cutoff;
        resonance = o.resonance;
    }
};

// ---- maxiTrigger (H:564-596), maxiLagExp (H:499-560): host-side helpers of the reference, plain arithmetic -----------
class maxiTrigger {
public:
    double onZX(double input) {  // H:569-579
        double isZX = 0.0;
        if ((previousValue <= 0.0 || firstTrigger) && input > 0) isZX = 1.0;
        previousValue = input;
        firstTrigger = 0;
        return isZX;
    }
    double onChanged(double input, double tolerance) {  // H:582-591
        double changed = 0;
        if (std::abs(input - previousValue) > tolerance) changed = 1;
        previousValue = input;
        return changed;
    }

private:
    double previousValue = 1;
    bool firstTrigger = 1;
};
template <class T>
class maxiLagExp {  // H:499-560
public:
    T alpha, alphaReciprocal;
    T val;
    maxiLagExp() { init(0.5, 0.0); }
    maxiLagExp(T initAlpha, T initVal) { init(initAlpha, initVal); }
    void init(T initAlpha, T initVal) {
        alpha = initAlpha;
        alphaReciprocal = 1.0 - alpha;
        val = initVal;
    }
    inline void addSample(T newVal) { val = (alpha * newVal) + (alphaReciprocal * val); }
    void setAlpha(T alpha_) { alpha = alpha_; }
    void setAlphaReciprocal(T alphaReciprocal_) { alphaReciprocal = alphaReciprocal_; }
    void setVal(T val_) { val = val_; }
    T getAlpha() const { return alpha; }
    T getAlphaReciprocal() const { return alphaReciprocal; }
    inline T value() const { return val; }
};

// ---- maxiSample (H:602-790; C:605-1075): the play family over a buffer uploaded once ---------------------------------
// The buffer lives on the device.  The members that EDIT it (normalise, autoTrim, loopRecord) work on a host copy fetched on
// first use and write back what they changed -- setup-time utilities of the reference, not the per-sample path -- with the
// reference's own expressions.  `amplitudes` itself is not a member: read it with getAmplitudes().
class maxiSample {
    using Pool = maxigpu::ps::SamplePool;
    maxigpu::ps::Slot slot_;
    Pool::Buf buf_;
    std::vector<double> host_;  // host copy of the buffer (normalise / autoTrim / loopRecord / getAmplitudes), empty = not fetched
    double recordPosition_ = 0;
    maxiLagExp<double> loopRecordLag_;
    double run(int mode, double a = 0.0, double p0 = 0.0, double p1 = 0.0, double sig = 0.0) {
        if (!buf_.d) return 0.0;
        maxigpu::ps::Call c;
        c.method = mode;
        c.key = &buf_;
        c.a[0] = a; c.a[1] = p0; c.a[2] = p1; c.a[3] = sig;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }
    void fresh_state() {  // what a newly constructed maxiSample holds: zxTrig {1, first}, phasorPrev 0, phasorFirst (H:593-594, 731-732)
        slot_.sd[1] = 1.0;
        slot_.sd[2] = 0.0;
        slot_.si[0] = 1;
        slot_.si[1] = 1;
    }
    void drop() {  // (every caller replaces the play head afterwards, or destroys the object)
        maxigpu::ps::pool<Pool>().discard(slot_);
        if (buf_.d) mxg_sample_free(buf_.d);
        buf_.d = nullptr;
        buf_.len = 0;
        host_.clear();
    }
    void upload(const double *data, size_t n) {
        MAXIGPU_TRY {
        buf_.d = mxg_sample_upload(data, n);
        if (!buf_.d) maxigpu::ps::fatal(std::string("mxg_sample_upload: ") + mxg_last_error());
        buf_.len = n;
        }
        MAXIGPU_CATCH(return)
    }
    double amp_get(size_t i) {
        fetch_host();
        return i < host_.size() ? host_[i] : 0.0;
    }
    void amp_set(size_t i, double v) {
        fetch_host();
        if (i >= host_.size()) return;
        maxigpu::ps::pool<Pool>().settle(slot_);
        host_[i] = v;
        maxigpu::ps::check(mxg_memcpy_h2d(buf_.d + i, &host_[i], sizeof(double), nullptr), "h2d sample element");
    }
    void amp_assign(const std::vector<double> &v) {  // `amplitudes = v`: the buffer only
        maxigpu::ps::pool<Pool>().settle(slot_);
        const int rate = buf_.rate;
        if (buf_.d) mxg_sample_free(buf_.d);
        buf_.d = nullptr;
        buf_.len = 0;
        host_.clear();
        if (!v.empty()) upload(v.data(), v.size());
        host_ = v;
        buf_.rate = rate;
    }
    void fetch_host() {
        if (host_.size() == buf_.len) return;
        host_.assign(buf_.len, 0.0);
        if (buf_.len) maxigpu::ps::check(mxg_memcpy_d2h(host_.data(), buf_.d, sizeof(double) * buf_.len, nullptr), "d2h sample");
    }

public:
    // `maxiTrigger zxTrig` (H:606): the zero-crossing detector of the playOnZX family.  Its two private fields are slot state
    // (the trigger-driven players advance them on the device), so the member is a view of them with maxiTrigger's own two methods
    // (H:569-591): a direct call settles the slot first, like every other public member that lives in the slot.
    class ZxTrigMember {
        maxigpu::ps::Slot *s_;

    public:
        explicit ZxTrigMember(maxigpu::ps::Slot &s) : s_(&s) {}
        ZxTrigMember(const ZxTrigMember &) = delete;  // (bound to ONE object's slot; the owner's copy operations copy the state)
        ZxTrigMember &operator=(const ZxTrigMember &o) {  // `a.zxTrig = b.zxTrig`: the two fields
            maxigpu::ps::pool<Pool>().settle(*o.s_);
            const double pv = o.s_->sd[1];
            const int64_t ft = o.s_->si[0];
            maxigpu::ps::pool<Pool>().settle(*s_);
            s_->sd[1] = pv;
            s_->si[0] = ft;
            return *this;
        }
        ZxTrigMember &operator=(const maxiTrigger &) {  // a fresh maxiTrigger is all one can assign from outside: {1, first}
            maxigpu::ps::pool<Pool>().settle(*s_);
            s_->sd[1] = 1.0;
            s_->si[0] = 1;
            return *this;
        }
        double onZX(double input) {  // H:569-579
            maxigpu::ps::pool<Pool>().settle(*s_);
            double isZX = 0.0;
            if ((s_->sd[1] <= 0.0 || s_->si[0]) && input > 0) isZX = 1.0;
            s_->sd[1] = input;
            s_->si[0] = 0;
            return isZX;
        }
        double onChanged(double input, double tolerance) {  // H:582-591
            maxigpu::ps::pool<Pool>().settle(*s_);
            double changed = 0;
            if (std::abs(input - s_->sd[1]) > tolerance) changed = 1;
            s_->sd[1] = input;
            return changed;
        }
    };
    ZxTrigMember zxTrig{slot_};
    short myChannels = 1;      // C:546: the constructor's initialiser list -- myChannels(1), mySampleRate(maxiSettings::sampleRate)
    int mySampleRate = (int)maxiSettings::sampleRate;
    short myBitsPerSample = 0;
    string myPath;
    int myChunkSize = 0, mySubChunk1Size = 0, readChannel = 0;
    short myFormat = 0;
    int myByteRate = 0;
    short myBlockAlign = 0;
    maxiSample() { maxigpu::ps::pool<Pool>().attach(slot_); fresh_state(); }
    ~maxiSample() { drop(); maxigpu::ps::pool<Pool>().detach(slot_); }
    // The implicit copy constructor of the reference copies EVERYTHING (buffer, play head, trigger state, header fields) -- unlike
    // its operator=, which resets the head and takes the global rate (H:626-637).
    maxiSample(const maxiSample &source) {
        maxigpu::ps::pool<Pool>().attach(slot_);
        maxiSample &src = const_cast<maxiSample &>(source);
        maxigpu::ps::pool<Pool>().settle(src.slot_);
        if (src.buf_.d && src.buf_.len) {
            src.fetch_host();
            upload(src.host_.data(), src.host_.size());
            host_ = src.host_;
        }
        buf_.rate = src.buf_.rate;
        slot_.sd = src.slot_.sd;
        slot_.si = src.slot_.si;
        recordPosition_ = src.recordPosition_;
        loopRecordLag_ = src.loopRecordLag_;
        myChannels = src.myChannels; mySampleRate = src.mySampleRate; myBitsPerSample = src.myBitsPerSample; myPath = src.myPath;
        myChunkSize = src.myChunkSize; mySubChunk1Size = src.mySubChunk1Size; readChannel = src.readChannel; myFormat = src.myFormat;
        myByteRate = src.myByteRate; myBlockAlign = src.myBlockAlign;
    }
    maxiSample &operator=(const maxiSample &source) {  // H:626-637: position = 0, the source's channels and samples, the GLOBAL rate
        if (this == &source) return *this;
        maxiSample &src = const_cast<maxiSample &>(source);
        src.fetch_host();
        const std::vector<double> data = src.host_;
        drop();
        upload(data.data(), data.size());
        myChannels = source.myChannels;
        buf_.rate = mySampleRate = (int)maxiSettings::sampleRate;
        slot_.sd[0] = 0;
        recordPosition_ = 0;
        return *this;
    }
    bool load(string fileName, int channel = 0) {  // C:605-609 -> read() C:612-692
        drop();
        myPath = fileName;
        readChannel = channel;
        size_t len = 0;
        int32_t hdr[8];
        buf_.d = mxg_sample_load_wav(fileName.c_str(), channel, &len, hdr);
        if (!buf_.d) {
            printf("ERROR: Could not load sample.");  // C:686
            return false;
        }
        buf_.len = len;
        myChunkSize = hdr[0]; mySubChunk1Size = hdr[1]; myFormat = (short)hdr[2]; myChannels = (short)hdr[3];
        buf_.rate = mySampleRate = hdr[4];
        myByteRate = hdr[5]; myBlockAlign = (short)hdr[6]; myBitsPerSample = (short)hdr[7];
        slot_.sd[0] = (double)len;  // position = size, C:681
        return true;
    }
    bool save() { return save(myPath); }  // C:694-696
    bool save(string filename) {           // C:698-725: shorts = round(a * 32767) on the device behind the 44-byte header of the members
        if (!buf_.d) return false;
        const int32_t hdr[8] = {myChunkSize, mySubChunk1Size, myFormat, myChannels, mySampleRate, myByteRate, myBlockAlign, myBitsPerSample};
        return mxg_sample_save_wav(filename.c_str(), buf_.d, buf_.len, hdr, nullptr) == MXG_OK;
    }
    string getSummary() {  // C:727-733: the header fields read() kept
        return " Format: " + std::to_string(myFormat) + "\n Channels: " + std::to_string(myChannels) + "\n SampleRate: " +
               std::to_string(mySampleRate) + "\n ByteRate: " + std::to_string(myByteRate) + "\n BlockAlign: " + std::to_string(myBlockAlign) +
               "\n BitsPerSample: " + std::to_string(myBitsPerSample);
    }
    void setSample(vector<double> &sampleData) {  // H:670-678
        drop();
        upload(sampleData.data(), sampleData.size());
        buf_.rate = mySampleRate = 44100;
        slot_.sd[0] = (double)sampleData.size() - 1;
    }
    void setSampleAndRate(vector<double> &sampleData, int sampleRate) {  // H:684-688
        setSample(sampleData);
        buf_.rate = mySampleRate = sampleRate;
    }
    const vector<double> &getAmplitudes() { fetch_host(); return host_; }
    // The reference's public member `vector<double> amplitudes` (H:621) over the device buffer: size / element reads go through a
    // host copy fetched on demand; an element write lands in the host copy and on the device at once (the play head's cached block is
    // dropped first: a later play() must see it); assigning a vector replaces the buffer and -- as in the reference -- touches neither
    // the play head nor the rates.  (The grain classes of include/maxiGrains.h take the device buffer, not this view.)
    class Amplitudes {
        maxiSample *s_;

    public:
        explicit Amplitudes(maxiSample *s) : s_(s) {}
        Amplitudes(const Amplitudes &) = delete;  // (bound to ONE sample: maxiSample's own copy operations copy the buffer)
        class Ref {
            maxiSample *s_;
            size_t i_;

        public:
            Ref(maxiSample *s, size_t i) : s_(s), i_(i) {}
            operator double() const { return s_->amp_get(i_); }
            Ref &operator=(double v) { s_->amp_set(i_, v); return *this; }
            Ref &operator=(const Ref &o) { s_->amp_set(i_, (double)o); return *this; }
            Ref &operator+=(double v) { s_->amp_set(i_, s_->amp_get(i_) + v); return *this; }
            Ref &operator-=(double v) { s_->amp_set(i_, s_->amp_get(i_) - v); return *this; }
            Ref &operator*=(double v) { s_->amp_set(i_, s_->amp_get(i_) * v); return *this; }
            Ref &operator/=(double v) { s_->amp_set(i_, s_->amp_get(i_) / v); return *this; }
        };
        size_t size() const { return s_->buf_.len; }
        bool empty() const { return s_->buf_.len == 0; }
        Ref operator[](size_t i) { return Ref(s_, i); }
        double operator[](size_t i) const { return s_->amp_get(i); }
        Ref at(size_t i) { return Ref(s_, i); }
        operator const std::vector<double> &() const { s_->fetch_host(); return s_->host_; }
        std::vector<double>::const_iterator begin() const { s_->fetch_host(); return s_->host_.begin(); }
        std::vector<double>::const_iterator end() const { s_->fetch_host(); return s_->host_.end(); }
        const double *data() const { s_->fetch_host(); return s_->host_.data(); }
        Amplitudes &operator=(const std::vector<double> &v) { s_->amp_assign(v); return *this; }
        Amplitudes &operator=(const Amplitudes &o) { o.s_->fetch_host(); s_->amp_assign(std::vector<double>(o.s_->host_)); return *this; }
        void clear() { s_->clear(); }
    };
    Amplitudes amplitudes{this};
    size_t getLength() { return buf_.len; }
    bool isReady() { return buf_.len > 1; }
    void clear() { drop(); }  // H:691: amplitudes.clear()
    void trigger() {  // C:597-600
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd[0] = 0;
        recordPosition_ = 0;
    }
    void reset() {  // H:725
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd[0] = 0;
    }
    void setPosition(double newPos) {  // C:749-751
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd[0] = (newPos < 0.0 ? 0.0 : (newPos > 1.0 ? 1.0 : newPos)) * (double)buf_.len;
    }
    double play() { return run(MXG_SMP_PLAY); }
    double playOnce() { return run(MXG_SMP_PLAYONCE); }
    double playLoop(double start, double end) { return run(MXG_SMP_PLAYLOOP, 0.0, start, end); }
    double playUntil(double end) { return run(MXG_SMP_PLAYUNTIL, 0.0, 0.0, end); }
    double playAtSpeed(double speed) { return run(MXG_SMP_PLAYATSPEED, speed); }
    double playOnceAtSpeed(double speed) { return run(MXG_SMP_PLAYONCEATSPEED, speed); }
    double playUntilAtSpeed(double end, double speed) { return run(MXG_SMP_PLAYUNTILATSPEED, speed, 0.0, end); }
    double play4(double frequency, double start, double end) { return run(MXG_SMP_PLAY4, frequency, start, end); }
    double playAtSpeedBetweenPoints(double frequency, double start, double end) {
        return run(MXG_SMP_PLAYATSPEEDBETWEENPOINTS, frequency, start, end);
    }
    double playAtSpeedBetweenPointsFromPos(double frequency, double start, double end, double pos) {  // C:826-880 (pos by value)
        return run(Pool::kFromPos, frequency, start, end, pos);
    }
    // the trigger-driven players (C:1006-1042) and playWithPhasor (C:753-816): zxTrig / phasorPrev / phasorFirst are slot state
    double playWithPhasor(double pha) { return run(MXG_SMP_PLAYWITHPHASOR, 0.0, 0.0, 0.0, pha); }
    double playOnZX(double trig) { return run(MXG_SMP_PLAYONZX, 0.0, 0.0, 0.0, trig); }
    double playOnZXAtSpeed(double trig, double speed) { return run(MXG_SMP_PLAYONZXATSPEED, speed, 0.0, 0.0, trig); }
    double playOnZXAtSpeedFromOffset(double trig, double speed, double offset) {
        return run(MXG_SMP_PLAYONZXATSPEEDFROMOFFSET, speed, offset, 0.0, trig);
    }
    double playOnZXAtSpeedBetweenPoints(double trig, double speed, double offset, double length) {
        return run(MXG_SMP_PLAYONZXATSPEEDBETWEENPOINTS, speed, offset, length, trig);
    }
    double loopSetPosOnZX(double trig, double position) { return run(MXG_SMP_LOOPSETPOSONZX, 0.0, position, 0.0, trig); }
    // ---- buffer editing (host copy, the reference's expressions) ----
    void normalise(double maxLevel) {  // C:1126-1137
        if (!buf_.len) return;
        maxigpu::ps::pool<Pool>().settle(slot_);
        fetch_host();
        double maxValue = 0;
        for (size_t i = 0; i < host_.size(); i++)
            if (std::abs(host_[i]) > maxValue) maxValue = std::abs(host_[i]);
        float scale = maxLevel / maxValue;
        for (size_t i = 0; i < host_.size(); i++) host_[i] = round(scale * host_[i]);
        maxigpu::ps::check(mxg_memcpy_h2d(buf_.d, host_.data(), sizeof(double) * host_.size(), nullptr), "h2d sample");
    }
    // C:1139-1190.  The reference assigns the (still empty) trimmed vector to `amplitudes` BEFORE it copies the kept range out of
    // it (C:1167-1172: undefined behaviour, in practice a crash); what it is evidently meant to do is restated here: keep
    // [startMarker, endMarker), position = 0, then the fade of C:1179-1186 literally (its `fadeSize` IS the whole length for
    // anything longer than 100 samples, and every element is scaled from both ends).
    void autoTrim(float alpha, float threshold, bool trimStart, bool trimEnd) {
        if (!buf_.len) return;
        maxigpu::ps::pool<Pool>().settle(slot_);
        fetch_host();
        size_t startMarker = 0;
        if (trimStart) {
            maxiLagExp<double> startLag(alpha, 0);
            while (startMarker < host_.size()) {
                startLag.addSample(std::abs(host_[startMarker]));
                if (startLag.value() > threshold) break;
                startMarker++;
            }
        }
        int endMarker = (int)host_.size() - 1;
        if (trimEnd) {
            maxiLagExp<float> endLag(alpha, 0);
            while (endMarker > 0) {
                endLag.addSample(std::abs(host_[(size_t)endMarker]));
                if (endLag.value() > threshold) break;
                endMarker--;
            }
        }
        cout << "Autotrim: start: " << startMarker << ", end: " << endMarker << endl;
        if ((size_t)endMarker > startMarker) {
            const size_t newLength = (size_t)endMarker - startMarker;
            std::vector<double> newAmps(host_.begin() + (long)startMarker, host_.begin() + (long)(startMarker + newLength));
            size_t fadeSize = 100;
            if (newAmps.size() > fadeSize) fadeSize = newAmps.size();
            for (size_t i = 0; i < fadeSize && i < newAmps.size(); i++) {
                double factor = i / (double)fadeSize;
                newAmps[i] = round(newAmps[i] * factor);
                newAmps[newAmps.size() - 1 - i] = round(newAmps[newAmps.size() - 1 - i] * factor);
            }
            const int rate = buf_.rate;
            drop();
            upload(newAmps.data(), newAmps.size());
            host_ = newAmps;
            buf_.rate = rate;
            slot_.sd[0] = 0;
            recordPosition_ = 0;
        }
    }
    // H:706-722: one sample recorded into the buffer; the element it changes goes to the device at once (and whatever block a
    // player had rendered ahead is dropped: a play() of a later sample must see it)
    void loopRecord(double newSample, const bool recordEnabled, const double recordMix, double start, double end) {
        if (!buf_.len) return;
        fetch_host();
        loopRecordLag_.addSample(recordEnabled);
        if (recordPosition_ < start * host_.size()) recordPosition_ = start * host_.size();
        if (recordEnabled) {
            double currentSample = host_[(size_t)(int)recordPosition_] / 32767.0;
            newSample = (recordMix * currentSample) + ((1.0 - recordMix) * newSample);
            newSample *= loopRecordLag_.value();
            const size_t at = (size_t)(unsigned long)recordPosition_;
            if (at < host_.size()) {
                maxigpu::ps::pool<Pool>().settle(slot_);
                host_[at] = newSample * 32767;
                maxigpu::ps::check(mxg_memcpy_h2d(buf_.d + at, &host_[at], sizeof(double), nullptr), "h2d sample element");
            }
        }
        ++recordPosition_;
        if (recordPosition_ >= end * host_.size()) recordPosition_ = start * host_.size();
    }
    // (for the grain classes below: the device buffer they render from)
    const double *deviceSamples() const { return buf_.d; }
};

// ---- maxiDelayline (H:266-284; C:415-439): the ring lives on the device, one sample per launch --------------------
class maxiDelayline {
    using Pool = maxigpu::ps::DelayPool;
    maxigpu::ps::Slot slot_;
    Pool::Line line_;
    void init() {
        MAXIGPU_TRY {
        if (line_.d_mem) return;
        maxigpu::ps::check(mxg_init(-1), "mxg_init");
        line_.d_mem = static_cast<double *>(mxg_malloc(sizeof(double) * Pool::kCap));
        line_.d_save = static_cast<double *>(mxg_malloc(sizeof(double) * maxigpu::ps::kMaxBlock));
        line_.d_i = static_cast<int32_t *>(mxg_malloc(sizeof(int32_t) * 4));
        if (!line_.d_mem || !line_.d_save || !line_.d_i) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        maxigpu::ps::check(mxg_memset(line_.d_mem, 0, sizeof(double) * Pool::kCap, nullptr), "mxg_memset");  // ctor memset, C:415-417
        maxigpu::ps::check(mxg_sync(), "mxg_sync");
        }
        MAXIGPU_CATCH(return)
    }
    double run(int mode, double input, int size, double feedback, int position) {
        init();
        maxigpu::ps::Call c;
        c.method = mode;
        c.key = &line_;
        c.a[0] = input; c.a[1] = (double)size; c.a[2] = feedback; c.a[3] = (double)position;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }

public:
    maxiDelayline() { maxigpu::ps::pool<Pool>().attach(slot_); }
    ~maxiDelayline() {
        maxigpu::ps::pool<Pool>().discard(slot_);
        maxigpu::ps::pool<Pool>().detach(slot_);
        if (line_.d_mem) mxg_free(line_.d_mem);
        if (line_.d_save) mxg_free(line_.d_save);
        if (line_.d_i) mxg_free(line_.d_i);
    }
    // (H:266-284: a value type whose copy carries the 5.6 MB ring -- here a device-to-device copy of it)
    maxiDelayline(const maxiDelayline &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_from(o); }
    maxiDelayline &operator=(const maxiDelayline &o) { if (this != &o) copy_from(o); return *this; }
    double dl(double input, int size, double feedback) { return run(0, input, size, feedback, 0); }
    double dlFromPosition(double input, int size, double feedback, int position) { return run(1, input, size, feedback, position); }

private:
    void copy_from(const maxiDelayline &o) {
        maxiDelayline &src = const_cast<maxiDelayline &>(o);
        maxigpu::ps::pool<Pool>().settle(src.slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        if (src.line_.d_mem) {
            init();
            maxigpu::ps::check(mxg_memcpy_d2d_async(line_.d_mem, src.line_.d_mem, sizeof(double) * Pool::kCap, nullptr), "d2d ring");
            maxigpu::ps::check(mxg_stream_sync(nullptr), "mxg_stream_sync");
        } else if (line_.d_mem) {
            maxigpu::ps::check(mxg_memset(line_.d_mem, 0, sizeof(double) * Pool::kCap, nullptr), "mxg_memset");
        }
        line_.saved = false;
        slot_.sd = o.slot_.sd;
        slot_.si = o.slot_.si;
    }
};

// ---- maxiFFT (L/maxiFFT.h:47-110; L/maxiFFT.cpp:45-132): the hop buffer on the host, every frame on the device ---------------
class maxiFFT {
public:
    enum fftModes { NO_POLAR_CONVERSION = 0, WITH_POLAR_CONVERSION = 1 };
    maxiFFT() {}
    ~maxiFFT() { release(); }
    // (a value type in the reference: a copy is a second analyser with the same hop buffer and the same last frame)
    maxiFFT(const maxiFFT &o) { copy_from(o); }
    maxiFFT &operator=(const maxiFFT &o) { if (this != &o) copy_from(o); return *this; }
    void setup(int _fftSize = 1024, int _hopSize = 512, int _windowSize = 0) {  // L/maxiFFT.cpp:45-60
        MAXIGPU_TRY {
        release();
        askedWindow_ = _windowSize;
        plan_ = mxg_fft_plan_create(_fftSize, _hopSize, _windowSize);
        if (!plan_) maxigpu::ps::fatal(std::string("mxg_fft_plan_create: ") + mxg_last_error());
        fftSize = _fftSize;
        windowSize = _windowSize > fftSize ? _windowSize : fftSize;
        bins = fftSize / 2;
        hopSize = _hopSize;
        buffer.assign(fftSize, 0);
        magnitudes.assign(bins, 0);
        magnitudesDB.assign(bins, 0);
        phases.assign(bins, 0);
        real_.assign(bins, 0);
        imag_.assign(bins, 0);
        pos = windowSize - hopSize;
        newFFT = 0;
        d_in_ = static_cast<float *>(mxg_malloc(sizeof(float) * fftSize));
        d_out_ = static_cast<float *>(mxg_malloc(sizeof(float) * 4 * bins));
        if (!d_in_ || !d_out_) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        }
        MAXIGPU_CATCH(return)
    }
    bool process(float value, fftModes mode = maxiFFT::WITH_POLAR_CONVERSION) {  // L/maxiFFT.cpp:65-91
        if (!plan_) {  // (the reference writes through an empty vector here; no frame ever completes)
            maxigpu::ps::complain("maxiFFT::process before setup()");
            return false;
        }
        buffer[pos++] = value;
        newFFT = pos == windowSize;
        if (newFFT && maxigpu::ps::dead()) {  // the device path is off: the hop buffer keeps moving, the last spectrum stays
            std::memmove(&buffer[0], &buffer[0] + hopSize, (windowSize - hopSize) * sizeof(float));
            pos = windowSize - hopSize;
            return newFFT = false;
        }
        if (newFFT) {
            maxigpu::ps::check(mxg_memcpy_h2d(d_in_, buffer.data(), sizeof(float) * fftSize, nullptr), "h2d frame");
            float *re = d_out_, *im = d_out_ + bins, *mg = d_out_ + 2 * bins, *ph = d_out_ + 3 * bins;
            const bool polar = mode == WITH_POLAR_CONVERSION;
            maxigpu::ps::check(mxg_fft_batch(plan_, d_in_, (size_t)fftSize, 1, re, im, polar ? mg : nullptr, polar ? ph : nullptr,
                                             nullptr), "mxg_fft_batch");
            maxigpu::ps::check(mxg_memcpy_d2h(real_.data(), re, sizeof(float) * bins, nullptr), "d2h real");
            maxigpu::ps::check(mxg_memcpy_d2h(imag_.data(), im, sizeof(float) * bins, nullptr), "d2h imag");
            if (polar) {
                maxigpu::ps::check(mxg_memcpy_d2h(magnitudes.data(), mg, sizeof(float) * bins, nullptr), "d2h mags");
                maxigpu::ps::check(mxg_memcpy_d2h(phases.data(), ph, sizeof(float) * bins, nullptr), "d2h phases");
            }
            std::memmove(&buffer[0], &buffer[0] + hopSize, (windowSize - hopSize) * sizeof(float));
            pos = windowSize - hopSize;
        }
        return newFFT;
    }
    float *getReal() { return real_.data(); }
    float *getImag() { return imag_.data(); }
    std::vector<float> &getMagnitudes() { return magnitudes; }
    std::vector<float> &getPhases() { return phases; }
    std::vector<float> &getMagnitudesDB() { return magsToDB(); }
    std::vector<float> &magsToDB() {  // fft::convToDB, L/fft.cpp:526-534, on the device
        features(true, false, false);
        return magnitudesDB;
    }
    float spectralFlatness() { features(false, true, false); return feat_[0]; }  // L/maxiFFT.cpp:113-123
    float spectralCentroid() { features(false, false, true); return feat_[1]; }  // :125-132
    int getNumBins() { return bins; }
    int getFFTSize() { return fftSize; }
    int getHopSize() { return hopSize; }
    int getWindowSize() { return windowSize; }

private:
    void features(bool db, bool flat, bool cen) {
        if (!plan_ || maxigpu::ps::dead()) return;
        float *mg = d_out_ + 2 * bins;
        maxigpu::ps::check(mxg_memcpy_h2d(mg, magnitudes.data(), sizeof(float) * bins, nullptr), "h2d mags");
        float *ddb = d_out_, *df = d_out_ + bins;  // reuse the real / imag staging
        maxigpu::ps::check(mxg_fft_features(plan_, mg, 1, db ? ddb : nullptr, flat ? df : nullptr, cen ? df + 1 : nullptr, nullptr),
                           "mxg_fft_features");
        if (db) maxigpu::ps::check(mxg_memcpy_d2h(magnitudesDB.data(), ddb, sizeof(float) * bins, nullptr), "d2h db");
        if (flat) maxigpu::ps::check(mxg_memcpy_d2h(&feat_[0], df, sizeof(float), nullptr), "d2h flatness");
        if (cen) maxigpu::ps::check(mxg_memcpy_d2h(&feat_[1], df + 1, sizeof(float), nullptr), "d2h centroid");
    }
    void release() {
        if (plan_) mxg_fft_plan_destroy(plan_);
        if (d_in_) mxg_free(d_in_);
        if (d_out_) mxg_free(d_out_);
        plan_ = nullptr;
        d_in_ = d_out_ = nullptr;
    }
    void copy_from(const maxiFFT &o) {
        if (!o.plan_) {
            release();
            return;
        }
        setup(o.fftSize, o.hopSize, o.askedWindow_);
        buffer = o.buffer; magnitudes = o.magnitudes; magnitudesDB = o.magnitudesDB; phases = o.phases; real_ = o.real_; imag_ = o.imag_;
        pos = o.pos;
        newFFT = o.newFFT;
        feat_[0] = o.feat_[0];
        feat_[1] = o.feat_[1];
    }
    int askedWindow_ = 0;
    mxg_fft_plan *plan_ = nullptr;
    float *d_in_ = nullptr, *d_out_ = nullptr;
    int fftSize = 0, windowSize = 0, hopSize = 0, bins = 0, pos = 0;
    bool newFFT = false;
    std::vector<float> buffer, magnitudes, magnitudesDB, phases, real_, imag_;
    float feat_[2] = {0, 0};
};

// ---- maxiMFCC (L/maxiMFCC.h:41-211) ------------------------------------------------------------------------------
class maxiMFCC {
public:
    maxiMFCC() {}
    ~maxiMFCC() { release(); }
    maxiMFCC(const maxiMFCC &o) { copy_from(o); }
    maxiMFCC &operator=(const maxiMFCC &o) { if (this != &o) copy_from(o); return *this; }
    void setup(unsigned int numBins, unsigned int numFilters, unsigned int numCoeffs, double minFreq, double maxFreq) {  // :56-75
        MAXIGPU_TRY {
        release();
        numFilters_ = numFilters; minFreq_ = minFreq; maxFreq_ = maxFreq;
        plan_ = mxg_mfcc_plan_create(numBins, numFilters, numCoeffs, minFreq, maxFreq);
        if (!plan_) maxigpu::ps::fatal(std::string("mxg_mfcc_plan_create: ") + mxg_last_error());
        numBins_ = numBins;
        coeffs_.assign(numCoeffs, 0.0);
        d_in_ = static_cast<float *>(mxg_malloc(sizeof(float) * numBins));
        d_out_ = static_cast<double *>(mxg_malloc(sizeof(double) * numCoeffs));
        if (!d_in_ || !d_out_) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        }
        MAXIGPU_CATCH(return)
    }
    vector<double> &mfcc(vector<float> &powerSpectrum) {  // :77-81
        if (!plan_ || powerSpectrum.size() < numBins_) {  // (the reference reads past the vector / through null tables here)
            maxigpu::ps::complain(!plan_ ? "maxiMFCC::mfcc before setup()" : "maxiMFCC::mfcc: fewer values than bins");
            return coeffs_;
        }
        if (maxigpu::ps::dead()) {
            std::fill(coeffs_.begin(), coeffs_.end(), 0.0);
            return coeffs_;
        }
        maxigpu::ps::check(mxg_memcpy_h2d(d_in_, powerSpectrum.data(), sizeof(float) * numBins_, nullptr), "h2d spectrum");
        maxigpu::ps::check(mxg_mfcc_batch(plan_, d_in_, numBins_, 1, nullptr, nullptr, d_out_, 0, nullptr), "mxg_mfcc_batch");
        maxigpu::ps::check(mxg_memcpy_d2h(coeffs_.data(), d_out_, sizeof(double) * coeffs_.size(), nullptr), "d2h mfcc");
        return coeffs_;
    }

private:
    void release() {
        if (plan_) mxg_mfcc_plan_destroy(plan_);
        if (d_in_) mxg_free(d_in_);
        if (d_out_) mxg_free(d_out_);
        plan_ = nullptr;
        d_in_ = nullptr;
        d_out_ = nullptr;
    }
    void copy_from(const maxiMFCC &o) {
        if (!o.plan_) {
            release();
            return;
        }
        setup(o.numBins_, o.numFilters_, (unsigned)o.coeffs_.size(), o.minFreq_, o.maxFreq_);
        coeffs_ = o.coeffs_;
    }
    mxg_mfcc_plan *plan_ = nullptr;
    unsigned numBins_ = 0, numFilters_ = 0;
    double minFreq_ = 0, maxFreq_ = 0;
    float *d_in_ = nullptr;
    double *d_out_ = nullptr;
    vector<double> coeffs_;
};

// ---- maxiDCBlocker (H:1255-1267), maxiSVF (H:1281-1338), maxiBiquad (H:1343-1486) -----------------------------------------------
// The coefficient formulas (tan, pow, sqrt) run on the host libm (mxg_svf_coeffs_host / mxg_biquad_coeffs_host), the recurrences on
// the device: bit-exact.
class maxiDCBlocker {
    using Pool = maxigpu::ps::Filter2Pool;
    maxigpu::ps::Slot slot_;

public:
    maxiDCBlocker() { maxigpu::ps::pool<Pool>().attach(slot_); }
    ~maxiDCBlocker() { maxigpu::ps::pool<Pool>().detach(slot_); }
    maxiDCBlocker(const maxiDCBlocker &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_from(o); }
    maxiDCBlocker &operator=(const maxiDCBlocker &o) { if (this != &o) copy_from(o); return *this; }
    double play(double input, double R) {
        maxigpu::ps::Call c;
        c.method = 0;
        c.a[0] = input;
        c.a[1] = R;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }

private:
    void copy_from(const maxiDCBlocker &o) {
        maxigpu::ps::pool<Pool>().settle(const_cast<maxiDCBlocker &>(o).slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd = o.slot_.sd;
    }
};

class maxiSVF {
    using Pool = maxigpu::ps::Filter2Pool;
    maxigpu::ps::Slot slot_;
    double coef_[5] = {0, 0, 0, 0, 0};  // g1, g2, g3, g4, k
    double freq = 1000, res = 1;
    void setParams(double _freq, double _res) {  // H:1320-1332
        freq = _freq;
        res = _res;
        maxigpu::ps::check(mxg_svf_coeffs_host(1, &freq, &res, coef_), "mxg_svf_coeffs_host");
    }

public:
    maxiSVF() { maxigpu::ps::pool<Pool>().attach(slot_); setParams(1000, 1); }
    ~maxiSVF() { maxigpu::ps::pool<Pool>().detach(slot_); }
    maxiSVF(const maxiSVF &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_from(o); }
    maxiSVF &operator=(const maxiSVF &o) { if (this != &o) copy_from(o); return *this; }
    void setCutoff(double cutoff) { setParams(cutoff, res); }
    void setResonance(double q) { setParams(freq, q); }
    double play(double w, double lpmix, double bpmix, double hpmix, double notchmix) {
        maxigpu::ps::Call c;
        c.method = 1;
        c.a[0] = w;
        for (int i = 0; i < 5; i++) c.a[1 + i] = coef_[i];
        c.a[6] = lpmix; c.a[7] = bpmix; c.a[8] = hpmix; c.a[9] = notchmix;
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }

private:
    void copy_from(const maxiSVF &o) {
        maxigpu::ps::pool<Pool>().settle(const_cast<maxiSVF &>(o).slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd = o.slot_.sd;
        for (int i = 0; i < 5; i++) coef_[i] = o.coef_[i];
        freq = o.freq;
        res = o.res;
    }
};

class maxiBiquad {
    using Pool = maxigpu::ps::Filter2Pool;
    maxigpu::ps::Slot slot_;
    double coef_[5] = {0, 0, 0, 0, 0};  // a0, a1, a2, b1, b2 (H:1481)

public:
    enum filterTypes { LOWPASS, HIGHPASS, BANDPASS, NOTCH, PEAK, LOWSHELF, HIGHSHELF };
    maxiBiquad() { maxigpu::ps::pool<Pool>().attach(slot_); }
    ~maxiBiquad() { maxigpu::ps::pool<Pool>().detach(slot_); }
    maxiBiquad(const maxiBiquad &o) { maxigpu::ps::pool<Pool>().attach(slot_); copy_from(o); }
    maxiBiquad &operator=(const maxiBiquad &o) { if (this != &o) copy_from(o); return *this; }
    double play(double input) {
        maxigpu::ps::Call c;
        c.method = 2;
        c.a[0] = input;
        for (int i = 0; i < 5; i++) c.a[1 + i] = coef_[i];
        return maxigpu::ps::pool<Pool>().call(slot_, c);
    }
    void set(filterTypes filtType, double cutoff, double Q, double peakGain) {  // H:1376-1478
        const int32_t t = (int32_t)filtType;
        maxigpu::ps::check(mxg_biquad_coeffs_host(1, &t, &cutoff, &Q, &peakGain, coef_), "mxg_biquad_coeffs_host");
    }

private:
    void copy_from(const maxiBiquad &o) {
        maxigpu::ps::pool<Pool>().settle(const_cast<maxiBiquad &>(o).slot_);
        maxigpu::ps::pool<Pool>().settle(slot_);
        slot_.sd = o.slot_.sd;
        for (int i = 0; i < 5; i++) coef_[i] = o.coef_[i];
    }
};

// ---- maxiEnvGen (H:2268-2547): one envelope = one slot of the maxiEnvGen bank -------------------------------------------------
class maxiEnvGen {
    using Pool = maxigpu::ps::EnvGenPool;
    maxigpu::ps::Slot slot_;
    Pool::Shape shape_;
    Pool &pool() { return maxigpu::ps::pool<Pool>(); }
    void drop_table() {
        if (shape_.d_stages) mxg_free(shape_.d_stages);
        shape_.d_stages = nullptr;
        shape_.nstages = 0;
    }

public:
    static constexpr double HOLD = -46692;  // H:2271
    maxiEnvGen() {
        pool().attach(slot_);
        slot_.sd[2] = slot_.sd[3] = slot_.sd[4] = 1.0;  // maxiTrigger: previousValue = 1, firstTrigger = 1 (H:593-594)
        slot_.si[4] = slot_.si[5] = slot_.si[6] = 1;
    }
    ~maxiEnvGen() { pool().detach(slot_); drop_table(); }
    maxiEnvGen(const maxiEnvGen &o) { pool().attach(slot_); copy_from(o); }  // (H:2268-2547: a value type -- vectors of stages)
    maxiEnvGen &operator=(const maxiEnvGen &o) { if (this != &o) copy_from(o); return *this; }
    double play(double trigger) {  // H:2277-2356
        if (!shape_.d_stages) {  // no stages: a WAITING envelope only feeds its trigger detector (H:2279-2287)
            pool().settle(slot_);
            if (slot_.si[1] == 0) {
                slot_.sd[2] = trigger;
                slot_.si[4] = 0;
            }
            return slot_.sd[0];
        }
        maxigpu::ps::Call c;
        c.method = 0;
        c.key = &shape_;
        c.a[0] = trigger;
        return pool().call(slot_, c);
    }
    bool setup(vector<double> levels, vector<double> times, vector<double> curves, bool looping, bool allowRetrigger = false) {  // H:2366-2399
        MAXIGPU_TRY {
        if (!(levels.size() == times.size() + 1 && levels.size() == curves.size() + 1)) {
            cout << "maxiEnv::setup - levels array should be one longer than times and curves\n";
            return 0;
        }
        pool().settle(slot_);
        std::vector<double> tab(6 * times.size() + 6);
        const int ns = times.empty() ? 0 : mxg_envgen_stages_host(levels.size(), levels.data(), times.data(), curves.data(), tab.data());
        if (ns < 0) {
            cout << "maxiEnv::setup - only one hold section allowed\n";
            return 0;
        }
        drop_table();
        if (ns > 0) {
            maxigpu::ps::check(mxg_init(-1), "mxg_init");
            shape_.d_stages = static_cast<double *>(mxg_malloc(sizeof(double) * 6 * ns));
            if (!shape_.d_stages) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
            maxigpu::ps::check(mxg_memcpy_h2d(shape_.d_stages, tab.data(), sizeof(double) * 6 * ns, nullptr), "h2d stages");
            shape_.nstages = ns;
        }
        shape_.loop = looping;
        shape_.retrigger = allowRetrigger;
        resetAndArm();
        return 1;
        }
        MAXIGPU_CATCH(return false)
    }
    void reset() {  // H:2402-2410
        pool().settle(slot_);
        slot_.sd[1] = 0.0;  // stages[phase].currentlevel
        slot_.si[3] = 0;    // stages[phase].counter
        slot_.si[0] = 0;    // phase
        slot_.si[1] = 1;    // TRIGGERED
    }
    void resetAndArm() {  // H:2413-2416
        reset();
        slot_.si[1] = 0;  // WAITING
    }
    void setupAR(const double attack, const double release) { setup({0, 1, 0}, {attack, release}, {1, 1}, false, false); }
    void setupASR(const double attack, const double release) { setup({0, 1, 1, 0}, {attack, maxiEnvGen::HOLD, release}, {1, 1, 1}, false, false); }
    void setupADSR(const double attack, const double decay, const double sustain, const double release) {
        setup({0, 1, sustain, sustain, 0}, {attack, decay, maxiEnvGen::HOLD, release}, {1, 1, 1, 1}, false, false);
    }
    void setRetrigger(const bool val) { pool().settle(slot_); shape_.retrigger = val; }
    bool getRetrigger() { return shape_.retrigger; }
    void setLoop(const bool val) { pool().settle(slot_); shape_.loop = val; }
    bool getLoop() { return shape_.loop; }

private:
    void copy_from(const maxiEnvGen &o) {  // its own copy of the stage table on the device, the same running state
        MAXIGPU_TRY {
        pool().settle(const_cast<maxiEnvGen &>(o).slot_);
        pool().settle(slot_);
        drop_table();
        if (o.shape_.d_stages && o.shape_.nstages > 0) {
            shape_.d_stages = static_cast<double *>(mxg_malloc(sizeof(double) * 6 * o.shape_.nstages));
            if (!shape_.d_stages) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
            maxigpu::ps::check(mxg_memcpy_d2d_async(shape_.d_stages, o.shape_.d_stages, sizeof(double) * 6 * o.shape_.nstages, nullptr), "d2d stages");
            maxigpu::ps::check(mxg_stream_sync(nullptr), "mxg_stream_sync");
            shape_.nstages = o.shape_.nstages;
        }
        shape_.loop = o.shape_.loop;
        shape_.retrigger = o.shape_.retrigger;
        slot_.sd = o.slot_.sd;
        slot_.si = o.slot_.si;
        }
        MAXIGPU_CATCH(return)
    }
};

// ---- maxiIFFT (L/maxiFFT.h:117-156; L/maxiFFT.cpp:141-192): one inverse transform per hop on the device -------------------------
class maxiIFFT {
public:
    enum fftModes { SPECTRUM = 0, COMPLEX = 1 };
    maxiIFFT() {}
    ~maxiIFFT() { release(); }
    maxiIFFT(const maxiIFFT &o) { copy_from(o); }
    maxiIFFT &operator=(const maxiIFFT &o) { if (this != &o) copy_from(o); return *this; }
    void setup(int _fftSize = 1024, int _hopSize = 512, int _windowSize = 0) {  // L/maxiFFT.cpp:141-152
        MAXIGPU_TRY {
        release();
        askedWindow_ = _windowSize;
        plan_ = mxg_ifft_plan_create(_fftSize, _hopSize, _windowSize);
        if (!plan_) maxigpu::ps::fatal(std::string("mxg_ifft_plan_create: ") + mxg_last_error());
        fftSize = _fftSize;
        hopSize = _hopSize;
        bins = fftSize / 2;
        pos = 0;
        hop_.assign(hopSize, 0.0f);
        d_in_ = static_cast<float *>(mxg_malloc(sizeof(float) * 2 * bins));
        d_buffer_ = static_cast<float *>(mxg_malloc(sizeof(float) * fftSize));
        d_signal_ = static_cast<float *>(mxg_malloc(sizeof(float) * hopSize));
        if (!d_in_ || !d_buffer_ || !d_signal_) maxigpu::ps::fatal(std::string("mxg_malloc: ") + mxg_last_error());
        maxigpu::ps::check(mxg_memset(d_buffer_, 0, sizeof(float) * fftSize, nullptr), "mxg_memset");  // buffer.resize(fftSize, 0)
        }
        MAXIGPU_CATCH(return)
    }
    float process(std::vector<float> &data1, std::vector<float> &data2, fftModes mode = maxiIFFT::SPECTRUM) {  // :154-192
        using maxigpu::ps::check;
        if (!plan_ || (int)data1.size() < bins || (int)data2.size() < bins) {
            maxigpu::ps::complain(!plan_ ? "maxiIFFT::process before setup()" : "maxiIFFT::process: fewer values than bins");
            return 0.0f;
        }
        if (maxigpu::ps::dead()) return 0.0f;
        if (0 == pos) {  // the spectrum is consumed here; the overlap-add buffer lives on the device
            check(mxg_memcpy_h2d(d_in_, data1.data(), sizeof(float) * bins, nullptr), "h2d spectrum");
            check(mxg_memcpy_h2d(d_in_ + bins, data2.data(), sizeof(float) * bins, nullptr), "h2d spectrum");
            if (mode == SPECTRUM)
                check(mxg_ifft_batch(plan_, d_in_, d_in_ + bins, 1, d_buffer_, d_signal_, nullptr, nullptr), "mxg_ifft_batch");
            else  // the reference's COMPLEX mode as it computes (L/fft.cpp:613-619: the inputs never reach the transform)
                check(mxg_ifft_batch_complex(plan_, d_in_, d_in_ + bins, 1, 1, d_buffer_, d_signal_, nullptr, nullptr), "mxg_ifft_batch_complex");
            check(mxg_memcpy_d2h(hop_.data(), d_signal_, sizeof(float) * hopSize, nullptr), "d2h hop");
        }
        const float nextValue = hop_[pos];
        if (hopSize == ++pos) pos = 0;
        return nextValue;
    }
    int getNumBins() { return bins; }

private:
    void release() {
        if (plan_) mxg_ifft_plan_destroy(plan_);
        if (d_in_) mxg_free(d_in_);
        if (d_buffer_) mxg_free(d_buffer_);
        if (d_signal_) mxg_free(d_signal_);
        plan_ = nullptr;
        d_in_ = d_buffer_ = d_signal_ = nullptr;
    }
    void copy_from(const maxiIFFT &o) {  // the overlap-add buffer lives on the device: copied there
        if (!o.plan_) {
            release();
            return;
        }
        setup(o.fftSize, o.hopSize, o.askedWindow_);
        maxigpu::ps::check(mxg_memcpy_d2d_async(d_buffer_, o.d_buffer_, sizeof(float) * fftSize, nullptr), "d2d overlap-add buffer");
        maxigpu::ps::check(mxg_stream_sync(nullptr), "mxg_stream_sync");
        hop_ = o.hop_;
        pos = o.pos;
    }
    int askedWindow_ = 0;
    mxg_ifft_plan *plan_ = nullptr;
    float *d_in_ = nullptr, *d_buffer_ = nullptr, *d_signal_ = nullptr;
    int fftSize = 0, hopSize = 0, bins = 0, pos = 0;
    std::vector<float> hop_;
};

// ---- the plugin API (src/maximilian.cpp:205-207; cpp/commandline/player.cpp:21-44) ------------------------------------
void setup();
void play(double *output);
