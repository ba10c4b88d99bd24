// runtime.hip -- library state, error plumbing and the memory/stream pass-throughs of the
// C-ABI (include/maxigpu.h).  No compute lives here.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "mxg_common.h"

namespace mxg {

namespace {
thread_local char g_err[512] = "";
std::mutex g_mu;
bool g_inited = false;
int g_device = -1;
hipStream_t g_stream = nullptr;
Settings g_settings;

struct Tune {
    const char *key;
    int value;
    int lo, hi;
};
Tune g_tune[] = {
    {"osc_vpl", 0, 0, 2},       {"osc_block", 256, 64, 1024},  {"osc_nt", 2, 0, 2},
    {"voice_vpl", 1, 1, 2},     {"voice_block", 256, 64, 1024}, {"voice_nt", 2, 0, 2},
    {"part_spin_limit", 1048576, 1, 16777216},  // time parts: polls (x s_sleep 8) before the writer gives up and reports ASYNC_PART_TIMEOUT
    {"part_fault", 0, 0, 1},  // fault injection for the tests: the writer waits for one signal more than will ever come
    {"rw_chunk", 0, 0, 32},  // samples per chunk of the pair-row read + write kernels (16-byte loads in flight per lane = half of it): 0 automatic, 8 / 16 / 32
    {"rw_store", 0, 0, 4},  // read + write bank kernels (filter2 ...): 16-byte pair-row streams: 0 automatic (on for blocks >= 64 MB), 1 off, 2 / 3 / 4 on with plain / write-through / non-temporal stores
    {"fft_exact", 1, 0, 1},  // 0: mxg_fft_mfcc_batch (fftSize 1024) runs its tolerance-mode kernel: true radix-8 butterflies with correctly rounded twiddles and FMAs, hardware sqrt -- NOT the reference's bits (tolerance in the header)
    {"time_parallel", 0, 0, 1},  // 1: small banks of linear filters (biquad, SVF, DC blocker, hoisted lores / hires) are cut along TIME and joined by a wavefront scan (scan.hip) -- reordered arithmetic: a tolerance mode, not bit-exact
    {"voice_store", 0, 0, 5},  // K2f store stream: 0 automatic, 1 plain 8 B, 2 nt 8 B, 3 / 4 / 5 pair rows (16 B) plain / sc1 / nt
    {"voice_xcd", 0, 0, 2},    // 0 automatic, 1 natural workgroup order, 2 XCD-contiguous
    {"voice_mix_store", 0, 0, 5},  // K2f mixdown form (mxg_voice_render_mix*): 0 automatic (voice_store's rule), 1 ... 5 as voice_store
    {"voice_diet", 0, 0, 2},  // K2f plain mode A: 1 = the round-5 instruction stream of the fast paths (comparison); 0 / 2 = the short one (voice_kernel DIET)
    {"voice_pace", 0, 0, 4096},  // K2f: the paced schedule (a chunk every P ticks of 10 ns): 0 automatic (controller at the store-bound sizes whose grid is resident at once), 1 never, >= 2 fixed P
    {"osc_pace", 0, 0, 4096},  // K1: the paced schedule (eight samples every P ticks of 10 ns, mxg_pace.h): 0 automatic (controller + the simplest launch, table-free waveforms at 90 112 ... 327 680 voices), 1 never, >= 2 fixed P
    {"smp_pace", 0, 0, 4096},  // the sample players: the paced schedule (mxg_pace.h): 0 automatic (play() at 45 056 ... 229 375 voices, controller), 1 never, >= 2 a fixed period for play and the *AtSpeed players
    {"osc_store", 0, 0, 5},  // K1 store stream (osc.hip pick<WF>): 0 automatic; one voice per lane: 1 plain 8 B, 2 nt 8 B, 3 / 4 / 5 pair rows (16 B) plain / sc1 / nt; two voices per lane: 1 plain, 2 nt, 3 sc1
    {"osc_xcd", 0, 0, 2},  // K1: 1 = every XCD renders one contiguous eighth of the bank (workgroup renumbering, mxg_common.h)
    {"osc_plan", 0, 0, 3},  // K1, large banks: the plan of launches (98 304-voice passes + a remainder launch): 0 automatic, 1 never, 2 / 3 always (natural / XCD-contiguous numbering)
    {"osc_passes", 0, 0, 64},  // K1: voice groups a wavefront renders one after the other (0 automatic; the grid covers 1 / passes of the bank)
    {"osc_mix_passes", 0, 0, 64},  // K1m: the same for the fused render + mixdown
    {"osc_split", 0, 0, 8},  // K1: time parts per voice group (0 = automatic: 2 for sinewave / coswave / sinebuf4 below 131 072 voices; up to 8 for the table oscillators on banks smaller than the machine; else 1)
    {"osc_mix_store", 0, 0, 2},  // K1m per-voice block: 0 automatic, 1 plain 8-byte stores, 2 pair rows of write-through 16-byte stores
    {"osc_mix_split", 0, 0, 4},  // K1m time parts (0 automatic: two below 2048 wavefronts)
    {"osc_mix_pc", 0, 0, 2},  // K1m: producer / consumer wavefront pairs (0 automatic: from 32 768 voices, whole blocks only; 1 off; 2 on)
    {"osc_mix_pcwin", 0, 0, 512},  // K1m, producer / consumer form: combine window (0 automatic = 512 where the waveform's table leaves room, 256 / 512)
    {"osc_mix_win", 0, 0, 256},  // K1m: samples per workgroup combine window (0 automatic: 256, 128 from 131 073 voices; 128 / 256)
    {"ifft_stream", 1, 0, 2},  // maxiIFFT: transform + hop buffer in one kernel (0: never; 1: where hop >= fftSize / 2; 2: wherever it fits)
    {"smp_pipe", 1, 0, 1},   // K5: the time-part kernel's loads of chunk k+1 issued before the stores of chunk k (0: chunk after chunk)
    {"tab_sides", 0, 0, 2},  // K1t: 0 automatic (= 1), 1 workgroups of 256 lanes / one round at a time / two per CU, 2 workgroups of 512 lanes / two rounds side by side
    {"smp_ring", 0, 0, 2},   // K5 time-part kernel: rows as rings, only new 16-byte pieces fetched (0 automatic: samples beyond 1 GiB, 1 off, 2 on)
    {"smp_split", 0, 0, 8},  // K5: time parts of a block-constant *AtSpeed launch (0 = automatic: ~4 wavefronts per SIMD)
    {"mix_block", 256, 64, 1024},
    {"mix_rows", 2, 1, 2},  // K3: sample rows per workgroup sharing one read of the gains (stereo, no bus output)
    {"fft_generic", 0, 0, 1},  // 1: force the generic per-stage FFT kernel also for fftSize 1024
    {"grain_chunked", 1, 0, 1},  // 0: serial-in-time K8 instead of the time-sharded K8a+K8b
    {"grain_lanes_k", 128, 16, 4096},  // K8b: target number of (stream, chunk) lanes, in units of 1024
    {"grain_line", 1, 0, 1},  // K8d: tile render for arbitrary increments (0: the (stream, chunk) walk K8b)
    {"grain_unit", 1, 0, 1},  // K8c: coalesced closed-form render when every grain has inc = +-1
    {"grain_fast_sched", 1, 0, 1},  // K8a: event-driven exact multi-step scheduler (0: one step at a time)
    {"grain_sync", 0, 0, 1},  // 1: mxg_granular_render reads its error word back before it returns (the round-1/2 behaviour); 0: deferred
    {"grain_spin_limit", 0, 0, 1 << 22},  // polls without progress before a streamed tile render gives up and the call is rendered again (0 = 2^22)
    {"grain_streamed", 1, 0, 1},  // K8c: scheduler lanes and tile renders in ONE launch (0: time slices on the auxiliary streams, grain_slices)
    {"grain_slices", 4, 1, 16},  // K8a/K8c: time slices of a maxiTimeStretch call (scheduling of slice i+1 overlaps render of slice i)
    {"mfcc_mfma_fullk", 0, 0, 1},  // K7b: contract over all numBins bins (1) instead of the bins that carry weight
    {"fused_layout", 0, 0, 2},   // K67: 0 automatic; 1 two frames in flight, two 4-wave workgroups per CU; 2 one frame in flight, one 12-wave workgroup per CU
    {"fused_mel", 0, 0, 3},  // K67 mel / log / DCT stage: 0 automatic (3, or 2 when the band sums are requested); 1 sparse walk + vector DCT (round-4 form); 2 sparse walk (band sums bit-exact) + DCT on the matrix pipe; 3 mel contraction AND DCT on the matrix pipe (v_mfma_f64_4x4x4, banded; band sums 1e-13)
    {"fused_waves16", 0, 0, 1},  // K67: 1 = the 16-waves-per-CU form of the fused FFT+MFCC kernel when applicable (measured slower: 1.72 vs 1.51 ms)
    {"mfcc_tiled", 1, 0, 1},  // K7a-t: stage spectra through LDS tiles (0: per-lane row loads, K7a)
};
}  // namespace

Settings &settings() { return g_settings; }

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return MXG_OK;
    return fail(e == hipErrorOutOfMemory ? MXG_ERR_NOMEM : MXG_ERR_HIP, "%s: %s", what,
                hipGetErrorString(e));
}

// Compute and synchronisation entry points: initialise on first use and report (once, then clear) a pending asynchronous device
// error -- the call that finds one returns it and does NOT run.  Everything else (allocation, copies, plan / queue / event
// creation, uploads) goes through ensure_init_only(): a pending error must not turn a malloc into NULL or a plan into a
// half-built object; it stays pending for the next compute or sync call.
int ensure_init_only() {
    if (g_inited) return MXG_OK;
    return mxg_init(-1);
}
int ensure_init() {
    if (int s = ensure_init_only()) return s;
    return async_error_poll();
}

hipStream_t resolve_stream(void *stream) { return stream ? (hipStream_t)stream : g_stream; }

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, g_device >= 0 ? g_device : 0) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

namespace {
struct ScratchBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};
std::map<std::pair<int, hipStream_t>, ScratchBuf> g_scratch;
}  // namespace

int scratch_get(ScratchSlot slot, hipStream_t st, size_t bytes, void **out, bool *fresh) {
    std::lock_guard<std::mutex> lk(g_mu);
    ScratchBuf &b = g_scratch[std::make_pair((int)slot, st)];
    if (fresh) *fresh = b.cap < bytes || !b.ptr;
    if (b.cap < bytes || !b.ptr) {
        if (b.ptr) MXG_HIP(hipFree(b.ptr));  // hipFree waits for the work that may still use the old buffer
        b.ptr = nullptr;
        b.cap = 0;
        MXG_HIP(hipMalloc(&b.ptr, bytes ? bytes : 8));
        b.cap = bytes ? bytes : 8;
    }
    *out = b.ptr;
    return MXG_OK;
}

// The starting period of a paced launch (mxg_pace.h), in ticks of the device's constant counter (s_memrealtime = HIP's wall clock: its
// rate is a device attribute, 100 MHz on MI355X): the time the memory system needs for `bytes` (one chunk of the whole grid) at 6.6 TB/s
// -- a rate it takes without back-pressure on every box seen; the controllers come down from there.  0 (= not paced) if the rate is
// unknown or the schedule would be too coarse for the controller's one-tick steps (fewer than 24 ticks per chunk).
unsigned pace_start_period(size_t bytes) {
    static const double ticks_per_second = [] {
        int dev = 0, khz = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) return 0.0;
        // (the bank sizes the launches pace at are those of the whole chip, 256 CUs: a partition of it, or another part, is not paced)
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus != 256) return 0.0;
        return (double)khz * 1e3;
    }();
    const double p = (double)bytes / 6.6e12 * ticks_per_second + 0.5;
    return (p >= 24.0 && p < 1e6) ? (unsigned)p : 0u;
}

// The words of a pace controller (mxg_pace.h) in per-stream scratch, zeroed when first handed out; null -- the launch is then simply not
// paced -- while `st` is being captured into a graph and the words do not exist yet (no allocation inside a capture; a graph captured
// after the first eager launch carries the pointer and its replays keep the controller going).
unsigned *pace_words(ScratchSlot slot, hipStream_t st, size_t nwords) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) return nullptr;
    if (cap != hipStreamCaptureStatusNone) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_scratch.find(std::make_pair((int)slot, st));
        return (it != g_scratch.end() && it->second.ptr && it->second.cap >= nwords * sizeof(unsigned)) ? (unsigned *)it->second.ptr : nullptr;
    }
    void *p = nullptr;
    bool fresh = false;
    if (scratch_get(slot, st, nwords * sizeof(unsigned), &p, &fresh) != MXG_OK) return nullptr;
    if (fresh && hipMemsetAsync(p, 0, nwords * sizeof(unsigned), st) != hipSuccess) return nullptr;
    return (unsigned *)p;
}

namespace {
int *g_async_host = nullptr;  // pinned, device-mapped: kernels store a code, the host polls it without synchronising
int *g_async_dev = nullptr;
bool g_part_dirty = false;    // a part time-out left counters non-zero: zero them before the next split launch
}  // namespace

int *async_error_word() { return g_async_dev; }

int async_error_status(int code) {
    if (code == ASYNC_OK) return MXG_OK;
    if (code == ASYNC_PART_TIMEOUT) {
        {
            std::lock_guard<std::mutex> lk(g_mu);  // (read under g_mu by part_sync_get)
            g_part_dirty = true;
        }
        return fail(MXG_ERR_HIP, "asynchronous device error: a time-split kernel (osc / sample *AtSpeed) timed out waiting for its "
                                 "sibling parts; the per-voice state of that launch was not stored");
    }
    switch (code - ASYNC_GRAIN_BASE) {
        case 1: return fail(MXG_ERR_INVALID, "mxg_granular_render: more than 8 grains alive in a stream");
        case 2: return fail(MXG_ERR_INVALID, "mxg_granular_render: d_rnd exhausted (R too small)");
        case 3: return fail(MXG_ERR_INVALID, "mxg_granular_render: internal spawn list overflow");
        case 4:
            return fail(MXG_ERR_INVALID,
                        "mxg_granular_render: d_gst holds a live grain this plan could not have made (another grain length, or "
                        "an index/position outside the window/sample); let live grains finish or clear d_gst first");
        case 5:
            return fail(MXG_ERR_INVALID,
                        "mxg_granular_render: a grain was born with a NaN/Inf step or one longer than the sample (|speed| too "
                        "large for this sample length); its reads would leave the buffer");
        case 6:  // (rounds 5 only: since round 6 such a call is rendered again, mxg_granular_retries counts them)
            return fail(MXG_ERR_HIP,
                        "mxg_granular_render: a tile render saw no scheduler progress for its whole polling budget (streamed form)");
        default: break;
    }
    return fail(MXG_ERR_HIP, "asynchronous device error %d", code);
}

int async_error_poll() {
    if (!g_async_host) return MXG_OK;
    if (__atomic_load_n(g_async_host, __ATOMIC_ACQUIRE) == ASYNC_OK) return MXG_OK;  // (the common case: no write to the shared word)
    const int code = __atomic_exchange_n(g_async_host, 0, __ATOMIC_ACQ_REL);  // one step: a code stored in between is not lost
    return async_error_status(code);
}

int part_sync_get(hipStream_t st, size_t wavefronts, int parts, PartSync *out) {
    const size_t bytes = (wavefronts ? wavefronts : 1) * sizeof(int);
    std::lock_guard<std::mutex> lk(g_mu);
    ScratchBuf &b = g_scratch[std::make_pair((int)SCR_PART_SYNC, st)];
    if (b.cap < bytes || !b.ptr) {
        if (b.ptr) MXG_HIP(hipFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
        const size_t cap = bytes < 65536 ? 65536 : bytes;
        MXG_HIP(hipMalloc(&b.ptr, cap));
        b.cap = cap;
        MXG_HIP(hipMemsetAsync(b.ptr, 0, cap, st));  // ordered before the launch that uses it; launches leave it zero
    }
    if (g_part_dirty) {  // after a reported time-out: every stream's counters start from zero again
        for (auto &kv : g_scratch)
            if (kv.first.first == (int)SCR_PART_SYNC && kv.second.ptr)
                // ON the stream the counters belong to: ordered after its earlier launches and before its next one (a memset on the
                // null stream is not ordered against these non-blocking streams and could land in the middle of the next kernel)
                MXG_HIP(hipMemsetAsync(kv.second.ptr, 0, kv.second.cap, kv.first.second));
        g_part_dirty = false;
    }
    out->ctrs = (int *)b.ptr;
    out->err = g_async_dev;
    out->spin_limit = tune_get("part_spin_limit");
    out->others = parts - 1 + (tune_get("part_fault") ? 1 : 0);
    return MXG_OK;
}

// ---- per-kernel event timing -------------------------------------------------------------------
namespace {
struct ProfSlot {
    const char *label;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.0;
    size_t count = 0;
};
std::vector<ProfSlot> g_prof;
std::vector<hipEvent_t> g_prof_free;
volatile bool g_prof_on = false;
std::mutex g_prof_mu;

hipEvent_t prof_event() {
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
// fold finished pairs into the slot's sum; `all` waits for every pair
void prof_drain(ProfSlot &p, bool all) {
    size_t done = 0;
    for (; done < p.pending.size(); done++) {
        auto &pr = p.pending[done];
        if (all) (void)hipEventSynchronize(pr.second);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) != hipSuccess) {
            (void)hipGetLastError();
            break;  // not finished yet: later pairs on the stream are not either
        }
        p.ms += (double)ms;
        p.count++;
        g_prof_free.push_back(pr.first);
        g_prof_free.push_back(pr.second);
    }
    p.pending.erase(p.pending.begin(), p.pending.begin() + done);
}
}  // namespace

KernelTimer::KernelTimer(const char *label, hipStream_t stream) : slot(-1), st(stream), e0(nullptr) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = 0; i < g_prof.size(); i++)
        if (g_prof[i].label == label || !strcmp(g_prof[i].label, label)) slot = (int)i;
    if (slot < 0) {
        g_prof.push_back(ProfSlot());
        g_prof.back().label = label;
        slot = (int)g_prof.size() - 1;
    }
    e0 = prof_event();
    if (e0) (void)hipEventRecord(e0, st);
}
KernelTimer::~KernelTimer() {
    if (slot < 0 || !e0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e1 = prof_event();
    if (!e1) return;
    (void)hipEventRecord(e1, st);
    ProfSlot &p = g_prof[(size_t)slot];
    p.pending.emplace_back(e0, e1);
    if (p.pending.size() >= 1024) prof_drain(p, false);
}

int tune_get(const char *key) {
    for (auto &t : g_tune)
        if (!strcmp(t.key, key)) return t.value;
    return 0;
}

}  // namespace mxg

using namespace mxg;

extern "C" {

namespace {
// The HIP runtime draws from / reseeds the process-wide libc PRNG while it initialises (measured: the first rand() after
// hipGetDeviceCount + hipStreamCreate is 2081652679 instead of 1804289383).  maxiOsc::noise (C:214-220) and the grain jitter
// (L/maxiGrains.h:352) ARE that stream in the reference, so a host that links this library must find it where the reference
// would: glibc's rand() shares random()'s state, so the initialisation runs on a throw-away state array and the caller's is
// switched back in afterwards, untouched.
struct LibcPrngGuard {
    char scratch[128];
    char *saved;
    LibcPrngGuard() { saved = initstate(1u, scratch, sizeof(scratch)); }
    ~LibcPrngGuard() { if (saved) setstate(saved); }
};
}  // namespace

int mxg_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    LibcPrngGuard prng;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(MXG_ERR_NO_DEVICE,
                    "mxg_init: no HIP device visible (%s); libmaxigpu has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device >= 0) {
        if (device >= count) return fail(MXG_ERR_INVALID, "mxg_init: device %d of %d", device, count);
        MXG_HIP(hipSetDevice(device));
        g_device = device;
    } else if (g_device < 0) {
        MXG_HIP(hipGetDevice(&g_device));
    }
    if (!g_stream) MXG_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (!g_async_host) {
        MXG_HIP(hipHostMalloc((void **)&g_async_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        g_async_host[0] = 0;
        g_async_host[1] = 0;  // (the word behind the error word: streamed granular launches that were rendered again)
        MXG_HIP(hipHostGetDevicePointer((void **)&g_async_dev, g_async_host, 0));
    }
    g_inited = true;
    return MXG_OK;
}

const char *mxg_last_error(void) { return g_err; }

int mxg_granular_retries(void) { return g_async_host ? __atomic_load_n(g_async_host + 1, __ATOMIC_ACQUIRE) : 0; }
const char *mxg_version(void) { return "maxigpu 0.1 (gfx950)"; }

int mxg_settings(size_t sampleRate, size_t channels, size_t bufferSize) {
    if (sampleRate == 0) return fail(MXG_ERR_INVALID, "mxg_settings: sampleRate 0");
    g_settings.sampleRate = sampleRate;
    g_settings.channels = channels;
    g_settings.bufferSize = bufferSize;
    return MXG_OK;
}
size_t mxg_sample_rate(void) { return g_settings.sampleRate; }

void *mxg_malloc(size_t bytes) {
    if (ensure_init_only()) return nullptr;
    LibcPrngGuard prng;  // (the first allocation initialises more of the runtime)
    void *p = nullptr;
    if (check_hip(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc")) return nullptr;
    return p;
}
int mxg_free(void *d_ptr) {
    if (!d_ptr) return MXG_OK;
    MXG_HIP(hipFree(d_ptr));
    return MXG_OK;
}
int mxg_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    if (int s = ensure_init_only()) return s;
    hipStream_t st = resolve_stream(stream);
    MXG_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
    MXG_HIP(hipStreamSynchronize(st));  // h_src is borrowed for the call only
    return MXG_OK;
}
int mxg_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (int s = ensure_init_only()) return s;
    hipStream_t st = resolve_stream(stream);
    MXG_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    MXG_HIP(hipStreamSynchronize(st));
    return async_error_poll();  // whatever the work before the copy reported
}
int mxg_memcpy_h2d_async(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    if (int s = ensure_init_only()) return s;
    MXG_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, resolve_stream(stream)));
    return MXG_OK;
}
int mxg_memcpy_d2h_async(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (int s = ensure_init_only()) return s;
    MXG_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, resolve_stream(stream)));
    return MXG_OK;
}
int mxg_memcpy_d2d_async(void *d_dst, const void *d_src, size_t bytes, void *stream) {
    if (int s = ensure_init_only()) return s;
    if (bytes) MXG_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, resolve_stream(stream)));
    return MXG_OK;
}
void *mxg_host_alloc(size_t bytes) {
    if (ensure_init_only()) return nullptr;
    void *p = nullptr;
    if (check_hip(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
    return p;
}
int mxg_host_free(void *h_ptr) {
    if (h_ptr) MXG_HIP(hipHostFree(h_ptr));
    return MXG_OK;
}
int mxg_memset(void *d_dst, int value, size_t bytes, void *stream) {
    if (int s = ensure_init_only()) return s;
    MXG_HIP(hipMemsetAsync(d_dst, value, bytes, resolve_stream(stream)));
    return MXG_OK;
}
void *mxg_stream_create(void) {
    if (ensure_init_only()) return nullptr;
    hipStream_t s = nullptr;
    if (check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"))
        return nullptr;
    return (void *)s;
}
int mxg_stream_destroy(void *stream) {
    if (!stream) return MXG_OK;
    {   // drop the scratch buffers that belonged to this stream
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_scratch.begin(); it != g_scratch.end();) {
            if (it->first.second == (hipStream_t)stream) {
                if (it->second.ptr) (void)hipFree(it->second.ptr);
                it = g_scratch.erase(it);
            } else {
                ++it;
            }
        }
    }
    MXG_HIP(hipStreamDestroy((hipStream_t)stream));
    return MXG_OK;
}
int mxg_stream_sync(void *stream) {
    if (int s = ensure_init()) return s;
    MXG_HIP(hipStreamSynchronize(resolve_stream(stream)));
    return async_error_poll();
}
int mxg_sync(void) {
    if (int s = ensure_init()) return s;
    MXG_HIP(hipDeviceSynchronize());
    return async_error_poll();
}
int mxg_last_async_error(void) { return async_error_poll(); }
void *mxg_event_create(void) {
    if (ensure_init_only()) return nullptr;
    hipEvent_t e = nullptr;
    if (check_hip(hipEventCreate(&e), "hipEventCreate")) return nullptr;
    return (void *)e;
}
int mxg_event_destroy(void *event) {
    if (event) MXG_HIP(hipEventDestroy((hipEvent_t)event));
    return MXG_OK;
}
int mxg_event_record(void *event, void *stream) {
    if (int s = ensure_init_only()) return s;
    MXG_REQUIRE(event, "null event");
    MXG_HIP(hipEventRecord((hipEvent_t)event, resolve_stream(stream)));
    return MXG_OK;
}
int mxg_event_sync(void *event) {
    MXG_REQUIRE(event, "null event");
    MXG_HIP(hipEventSynchronize((hipEvent_t)event));
    return MXG_OK;
}
int mxg_event_query(void *event) {  // 1 complete, 0 still running
    MXG_REQUIRE(event, "null event");
    hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return 0;
    }
    return check_hip(e, "hipEventQuery");
}
int mxg_stream_wait_event(void *stream, void *event) {
    if (int s = ensure_init_only()) return s;
    MXG_REQUIRE(event, "null event");
    MXG_HIP(hipStreamWaitEvent(resolve_stream(stream), (hipEvent_t)event, 0));
    return MXG_OK;
}
// cpp/commandline/player.cpp:25-44 `routing()`, restated without RtAudio: play() once per frame, then `channels`
// doubles copied to the interleaved buffer.  lastValues persists across calls like the callback's userData.
int mxg_host_render(void (*play)(double *), size_t channels, size_t nFrames, double *h_interleaved, double *h_lastValues) {
    MXG_REQUIRE(play && h_interleaved && h_lastValues && channels > 0, "null argument");
    double *buffer = h_interleaved;
    for (size_t i = 0; i < nFrames; i++) {
        play(h_lastValues);
        for (size_t j = 0; j < channels; j++) *buffer++ = h_lastValues[j];
    }
    return MXG_OK;
}
int mxg_event_elapsed_ms(void *start, void *stop, float *h_ms) {
    MXG_REQUIRE(start && stop && h_ms, "null argument");
    MXG_HIP(hipEventSynchronize((hipEvent_t)stop));
    MXG_HIP(hipEventElapsedTime(h_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MXG_OK;
}

int mxg_tune(const char *key, int value) {
    if (!key) return fail(MXG_ERR_INVALID, "mxg_tune: null key");
    for (auto &t : g_tune) {
        if (!strcmp(t.key, key)) {
            if (value < t.lo || value > t.hi || (t.lo == 64 && (value % 64)))
                return fail(MXG_ERR_INVALID, "mxg_tune: %s=%d out of range [%d,%d]", key, value,
                            t.lo, t.hi);
            int prev = t.value;
            t.value = value;
            return prev;
        }
    }
    return fail(MXG_ERR_INVALID, "mxg_tune: unknown key %s", key);
}

int mxg_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int prev = g_prof_on ? 1 : 0;
    g_prof_on = on != 0;
    return prev;
}
int mxg_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &p : g_prof) {
        prof_drain(p, true);
        p.ms = 0.0;
        p.count = 0;
    }
    return MXG_OK;
}
int mxg_prof_count(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof.size();
}
int mxg_prof_read(int index, const char **h_label, double *h_total_ms, size_t *h_launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (index < 0 || (size_t)index >= g_prof.size()) return fail(MXG_ERR_INVALID, "mxg_prof_read: index %d", index);
    ProfSlot &p = g_prof[(size_t)index];
    prof_drain(p, true);
    if (h_label) *h_label = p.label;
    if (h_total_ms) *h_total_ms = p.ms;
    if (h_launches) *h_launches = p.count;
    return MXG_OK;
}

int mxg_prof_overhead_ms(void *stream, int pairs, double *h_ms) {
    if (int s = ensure_init_only()) return s;
    MXG_REQUIRE(pairs > 0 && pairs <= 4096 && h_ms, "bad argument");
    hipStream_t st = resolve_stream(stream);
    MXG_HIP(hipStreamSynchronize(st));
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev((size_t)pairs);
    for (auto &p : ev) {
        MXG_HIP(hipEventCreate(&p.first));
        MXG_HIP(hipEventCreate(&p.second));
    }
    for (auto &p : ev) {  // exactly what a KernelTimer does around a launch, with no launch in between
        MXG_HIP(hipEventRecord(p.first, st));
        MXG_HIP(hipEventRecord(p.second, st));
    }
    MXG_HIP(hipStreamSynchronize(st));
    double sum = 0.0;
    for (auto &p : ev) {
        float ms = 0.f;
        MXG_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        sum += (double)ms;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    *h_ms = sum / pairs;
    return MXG_OK;
}

}  // extern "C"
