// mxg_common.h -- shared host/device plumbing for libmaxigpu.so (gfx950 only).
//
// Compile contract (see maximilian_amd/csrc/Makefile): hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off.  The last flag is part of the numerics contract: the reference is
// compiled without FMA contraction (x86-64 SSE2), so every a*b+c below must stay a separate
// v_mul_f64 + v_add_f64 or the bank outputs are no longer bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/maxigpu.h"

namespace mxg {

// H:55-58
#define MXG_TWOPI 6.283185307179586476925286766559
#define MXG_PI 3.1415926535897932384626433832795

// C:53 `float chandiv = 1;` -- multiplications by it are kept so the expression trees
// match; x*1.0 is exact, the compiler may fold it.
constexpr double kChandiv = 1.0;

struct Settings {
    size_t sampleRate = 44100;  // C:57
    size_t channels = 2;        // C:58
    size_t bufferSize = 1024;   // C:59
};
Settings &settings();

// error plumbing (thread-local message, negative status codes)
int fail(int status, const char *fmt, ...);
int check_hip(hipError_t e, const char *what);
int ensure_init();       // compute / sync entry points: init on first use + report a pending asynchronous device error
int ensure_init_only();  // allocation, copies, creation: init on first use, pending errors stay pending
hipStream_t resolve_stream(void *stream);
int tune_get(const char *key);
int device_cus();  // compute units of the device (256 on MI355X)
// The out[n*V + v] stream of the read+write / write-only bank kernels (knob rw_store: 0 automatic, 1 the 8-byte stores, 2 / 3 / 4 16-byte
// pair rows with plain / write-through / non-temporal stores): returns 0 = 8-byte stores, else the px_store flavour of emit_chunk (1 / 2 / 3).
// Pair rows need V even and a 16-byte aligned block.  Automatic, from tools/sweep_rw_store.py over rotating 4 GiB arenas
// (profiles/r04_rw_store.md), for blocks from 64 MB: kernels that READ a block and write one (maxiFilter, maxiEnv, filter2) take the
// write-through pair rows (65 536 voices: lores 117 -> 98 us, adsr in sustain 116 -> 96); kernels that only WRITE (maxiSample, maxiEnvGen)
// and maxiDelayline (whose ring traffic dominates) the non-temporal ones (131 072 voices: playAtSpeed 140 -> 120 us, maxiEnvGen 120 -> 106;
// write-through is SLOWER than 8-byte stores there) -- maxiEnvGen only from 98 304 voices: at 65 536 its state-machine path is faster
// with 8-byte stores (55 against 60-62 us).
enum RwFamily { RW_READ_WRITE, RW_WRITE_ONLY, RW_ENVGEN };
inline int rw_store_choice(size_t V, size_t N, const void *d_out, RwFamily fam = RW_READ_WRITE) {
    int rw = tune_get("rw_store");
    if (rw == 0) {
        const bool big = V * N * sizeof(double) >= ((size_t)64 << 20);
        if (!big) rw = 1;
        else if (fam == RW_READ_WRITE) rw = 3;
        else if (fam == RW_WRITE_ONLY) rw = 4;
        else rw = V >= 98304 ? 4 : 1;
    }
    const bool pairs_ok = V >= 2 && !(V & 1) && !(((uintptr_t)d_out) & 15);
    return (rw >= 2 && pairs_ok) ? rw - 1 : 0;
}

// Library-owned scratch, one grow-only buffer per (slot, stream): launches on different streams never share
// (or resize) each other's temporaries; launches on one stream are ordered by the stream.  Returns MXG_OK and
// a device pointer of at least `bytes`.
enum ScratchSlot { SCR_MIX_GAINS, SCR_OSC_MIX, SCR_GRAIN_ERR, SCR_GRAIN_SCHED, SCR_IFFT_OUT, SCR_IFFT_BUF, SCR_MFCC_RAW, SCR_CONVOLVE, SCR_GRAIN_MIX, SCR_PART_SYNC, SCR_OSCTAB_MARKS, SCR_OSCTAB_MARKS2, SCR_OSCTAB_CARRY, SCR_OSCTAB_SUM, SCR_VOICE_PACE, SCR_OSC_PACE, SCR_SMP_PACE, SCR_SLOTS };
int scratch_get(ScratchSlot slot, hipStream_t st, size_t bytes, void **out, bool *fresh = nullptr);  // *fresh: newly allocated (contents undefined)
unsigned pace_start_period(size_t bytes);  // a paced launch's starting period in ticks of the device's constant counter (runtime.hip); 0 = do not pace
unsigned *pace_words(ScratchSlot slot, hipStream_t st, size_t nwords);  // a pace controller's words (runtime.hip), or null inside a capture

// Optional per-kernel timing (mxg_prof_enable): a KernelTimer around a launch records two HIP events on the launch
// stream; mxg_prof_read sums the elapsed times per label.  Off by default: then it costs one load and a branch.
struct KernelTimer {
    KernelTimer(const char *label, hipStream_t st);
    ~KernelTimer();
    int slot;
    hipStream_t st;
    hipEvent_t e0;
};

#define MXG_HIP(call)                                         \
    do {                                                      \
        int _s = ::mxg::check_hip((call), #call);             \
        if (_s) return _s;                                    \
    } while (0)

#define MXG_REQUIRE(cond, msg)                                               \
    do {                                                                     \
        if (!(cond)) return ::mxg::fail(MXG_ERR_INVALID, "%s: %s", __func__, msg); \
    } while (0)

// ---- small banks of linear filters, parallel in time (scan.hip; tolerance mode, knob "time_parallel") ---------------------
bool scan_applies(size_t V, size_t N);
int scan_filter_launch(int kind, size_t V, size_t N, const double *in, const double *coef, double *st, double *out, hipStream_t s);

// ---- asynchronous device errors ------------------------------------------------------------------------------------
// A kernel that detects a failure the host cannot see at launch time (a time part that never got its signals, a grain render
// with an exhausted rand() queue ...) stores a code in ONE word of pinned, device-mapped host memory.  Nothing synchronises for
// it: the NEXT C-ABI call of any kind finds the word set, returns MXG_ERR_HIP / MXG_ERR_INVALID with the message, and clears it
// (mxg_last_async_error() does the same on request) -- so a failure is reported once, loudly, at the first call after it.
enum AsyncError {
    ASYNC_OK = 0,
    ASYNC_GRAIN_BASE = 100,   // + the granular render's code 1..5 (grains.hip)
    ASYNC_PART_TIMEOUT = 16,  // part_wait gave up: the state of that launch was NOT stored
};
int *async_error_word();  // device-visible address of the word (valid after mxg_init)
int async_error_poll();   // host: MXG_OK, or the pending error as a status (+ message), cleared
int async_error_status(int code);  // the status + message of one code

// ---- time parts: who may overwrite the state ----------------------------------------------------------------------
// A kernel cut into gridDim.y time parts reads its per-voice state in EVERY part and stores the new state from ONE of them (the
// writer).  Nothing orders the workgroups of a launch: with little work per part the writer can be done before another part of the
// same voices has even started, and that part would then start from the NEW state.  So every other part signals, per wavefront,
// once its state loads have returned, and the writer waits for those signals before it stores.  Workgroups are dispatched in
// order (x fastest, then y) and the writer is ALWAYS the last part (the last dispatched, with the most work in front of its
// store), so it never waits for work that has not been dispatched yet and the wait is normally over before it begins.  It is
// bounded anyway (`spin_limit` polls, knob part_spin_limit): on a time-out the writer reports ASYNC_PART_TIMEOUT through the
// async error word and does NOT store -- the launch's state is lost, loudly, instead of a late part silently starting from
// the new state -- and leaves the counter as it is; the host zeroes the counters again before the next split launch after an
// error.  Otherwise the writer leaves the counter at zero.
// CONTRACT after ASYNC_PART_TIMEOUT: the counters of that stream are not epoch-based, so split launches that were already
// enqueued behind the failed one start from stale counts and may store early.  A caller that sees MXG_ERR_ASYNC with this code must
// synchronise the stream and treat every bank state rendered on it since the failed launch as invalid (re-upload or re-create the
// banks); include/maxigpu.h says so at mxg_last_async_error.  (Epoch-based counters -- the writer waiting for a per-launch target instead
// of resetting to zero -- would close that window, but the target would be a kernel argument frozen into a captured hipGraph: a replayed
// split launch would then find its wait already satisfied, SILENTLY.  The reset-to-zero form is what keeps these launches replayable.)  The time-out itself means a device that made no progress for
// spin_limit x ~0.5 us on work that was dispatched BEFORE the writer -- it has only ever been seen under fault injection (knob part_fault).
struct PartSync {
    int *ctrs = nullptr;   // one counter per wavefront of gridDim.x (zero between launches)
    int *err = nullptr;    // the async error word
    int spin_limit = 1 << 20;
    int others = 0;        // signals the writer waits for (parts - 1; the fault-injection knob part_fault adds one that never comes)
};
// counters for a launch of `wavefronts` wavefronts per part in `parts` time parts on stream st
int part_sync_get(hipStream_t st, size_t wavefronts, int parts, PartSync *out);

__device__ __forceinline__ int *part_counter(const PartSync &ps) {
    return ps.ctrs ? ps.ctrs + ((size_t)blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6)) : nullptr;
}
__device__ __forceinline__ void part_signal(int *ctr) {
    if (!ctr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the state is in registers
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((unsigned long long)__ballot(1)) - 1))
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true: every other part has read its state -- store yours.  false: timed out (reported), do not store.
__device__ __forceinline__ bool part_wait(int *ctr, const PartSync &ps) {
    if (!ctr) return true;
    int ok = 1;
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((unsigned long long)__ballot(1)) - 1)) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ps.others && spins < ps.spin_limit) {
            __builtin_amdgcn_s_sleep(8);
            spins++;
        }
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ps.others) {
            __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = 0;
            if (ps.err) __hip_atomic_store(ps.err, (int)ASYNC_PART_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    ok = __shfl(ok, __ffsll((unsigned long long)__ballot(1)) - 1);
    __builtin_amdgcn_wave_barrier();
    return ok != 0;
}

// ---- XCD-aware workgroup numbering ---------------------------------------------------------------------------------
// Workgroups are handed to the eight XCDs round-robin (workgroup b runs on XCD b mod 8, each with its own 4 MB L2).  With
// xcd != 0 and a grid that is a multiple of 8, workgroup b takes the unit range that gives every XCD one CONTIGUOUS eighth of
// the bank (of a row) instead of every eighth piece.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nblk, int xcd) {
    if (!xcd || (nblk & 7)) return b;
    return (b & 7) * (nblk >> 3) + (b >> 3);
}

// ---- device store helpers ---------------------------------------------------------------
// Store flavours of the out[n*V + v] streams (measured with csrc/calib.hip, profiles/r03_write_ceiling.md): 0 plain,
// 1 non-temporal, 2 write-through (`sc1`: the line leaves the XCD's L2 with the store instead of by eviction -- the fastest
// 16-byte stream while a block is within a few times the Infinity Cache, no gain beyond).
#ifndef MXG_STORE_CLOBBER
#define MXG_STORE_CLOBBER 1  // A/B (tools/build_ab.sh): 1 = the write-through store asm carries a "memory" clobber
#endif
#if MXG_STORE_CLOBBER
#define MXG_STORE_CLOBBER_LIST : "memory"
#else
#define MXG_STORE_CLOBBER_LIST
#endif
template <int ST>
__device__ __forceinline__ void store1(double *p, double v) {
    if constexpr (ST == 1)
        __builtin_nontemporal_store(v, p);
    else if constexpr (ST == 2)
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) MXG_STORE_CLOBBER_LIST);  // (see store2)
    else
        *p = v;
}
typedef double double2v __attribute__((ext_vector_type(2)));
template <int ST>
__device__ __forceinline__ void store2(double *p, double a, double b) {
    double2v v = {a, b};
    if constexpr (ST == 1)
        __builtin_nontemporal_store(v, reinterpret_cast<double2v *>(p));
    else if constexpr (ST == 2)  // (the wait states a following write of the four data registers needs: hipcc does not add them after asm)
        // The "memory" clobber is the MEASURED default (MXG_STORE_CLOBBER 1 above): without it hipcc may hoist the next samples' LDS table
        // reads over this store -- a whole chunk's reads in flight instead of two samples' -- which looked right and measured slower on
        // K1's store-bound loop (round 4 A/B builds; the switch stays for such experiments).  `volatile` keeps the store itself and its
        // order among the other volatile asm statements.
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) MXG_STORE_CLOBBER_LIST);
    else
        *reinterpret_cast<double2v *>(p) = v;
}
// A 16-byte store at a WAVE-UNIFORM base plus a per-lane byte offset.  For the write-through flavour this is a buffer store the compiler
// KNOWS about (`__builtin_amdgcn_raw_buffer_store_b128`, aux = sc1) instead of store2's inline asm: hipcc does not count an asm store in
// its vmcnt bookkeeping, so in a loop that also LOADS every `s_waitcnt vmcnt(n)` it places is short by the asm stores in flight -- and the
// wait for a chunk's loads then drains the stores issued after them (found late in round 4: filter_pairs_kernel waited for vmcnt(0)
// once per chunk).  Loops without loads (K1) are unaffected and keep the asm form.  `ubase` must be uniform (kernel argument + uniform
// index): a divergent one would cost a waterfall loop.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int ST>
__device__ __forceinline__ void store2_at(double *ubase, unsigned off, double a, double b) {
    if constexpr (ST == 2) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, 0xffffffff, 0x00020000);
        const double2v v = {a, b};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)off, 0, 16 /* sc1 */);
    } else {
        store2<ST>(reinterpret_cast<double *>(reinterpret_cast<char *>(ubase) + off), a, b);
    }
}
// The exchange inside a lane pair (lanes 2k, 2k+1) that both pair-row streams are made of: every lane gives (x, y) and gets
//     a = even lane ? its own x : the partner's y,        b = odd lane ? its own y : the partner's x.
// Four VALU instructions: one v_cndmask_b32_dpp per dword (the DPP operand is the partner's value, quad_perm [1,0,3,2]; the lane parity
// is the select mask, in VCC: VOP2 is the only encoding that takes a DPP operand on gfx950, and it reads VCC).  Left to hipcc the same
// exchange is a v_mov_b32_dpp plus a v_cndmask_b32_e64 per dword -- 8 instructions per 16-byte store, 10 per 16-byte load with the
// selects around it -- which is 10 % of the VALU-bound oscillators (sinewave, sinebuf4: profiles/r04_heavy_osc.md).
// Both lanes of a pair must be live.  (s_mov + s_nop: the two wait states between a VALU write of a VGPR and a DPP read of it, which
// hipcc's hazard recognizer does not insert in front of inline asm.)
// ASM = false: the exchange written with __builtin_amdgcn_update_dpp and selects, as in rounds 1-3 -- kept for the kernels whose
// time is their store stream's and whose loop was tuned in that form (K1's sinebuf: osc.hip).
template <bool ASM = true>
__device__ __forceinline__ void pair_exchange(double x, double y, double &a, double &b) {
    const int x0 = __double2loint(x), x1 = __double2hiint(x), y0 = __double2loint(y), y1 = __double2hiint(y);
    int a0, a1, b0, b1;
    if constexpr (ASM) {
    const unsigned long long even = 0x5555555555555555ull;  // (workgroups are whole wavefronts: lane parity = thread parity)
    asm("s_mov_b64 vcc, %8\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "s"(even)
        : "vcc", "scc");
    } else {
    const bool odd = (threadIdx.x & 1) != 0;
    constexpr int kQuadXor1 = 0xB1;  // quad_perm [1,0,3,2]: the partner lane
    // (the exchanges are evaluated by every lane BEFORE the selects: a DPP read under a divergent branch would see a masked partner)
    const int py0 = __builtin_amdgcn_update_dpp(0, y0, kQuadXor1, 0xf, 0xf, true);
    const int py1 = __builtin_amdgcn_update_dpp(0, y1, kQuadXor1, 0xf, 0xf, true);
    const int px0 = __builtin_amdgcn_update_dpp(0, x0, kQuadXor1, 0xf, 0xf, true);
    const int px1 = __builtin_amdgcn_update_dpp(0, x1, kQuadXor1, 0xf, 0xf, true);
    a0 = odd ? py0 : x0, a1 = odd ? py1 : x1;
    b0 = odd ? y0 : px0, b1 = odd ? y1 : px1;
    }
    a = __hiloint2double(a1, a0);
    b = __hiloint2double(b1, b0);
}

// Two samples of one voice per lane -> one 16-byte store per lane: lanes 2k and 2k+1 (voices v, v+1) swap one value, so that
// the even lane holds sample n of both voices (16 contiguous bytes of row n) and the odd lane sample n+1 of both (row n+1).
// One wave store then covers 512 contiguous bytes in each of two rows with 16 bytes per lane.  `o` is the lane's own
// pointer: out + (n + (lane & 1)) * V + (v & ~1); both lanes of a pair must be live (V even).
// (a = sample n of voice 2k: the even lane's own r0 / for the odd lane, whose row is n+1, the partner's r1; b = voice 2k+1 likewise)
template <int ST, bool ASM = true>
__device__ __forceinline__ void store_pair_rows(double *o, double r0, double r1) {
    double a, b;
    pair_exchange<ASM>(r0, r1, a, b);
    store2<ST>(o, a, b);
}

// store_pair_rows at a wave-uniform base + per-lane byte offset (store2_at: the compiler-visible write-through store)
template <int ST, bool ASM = true>
__device__ __forceinline__ void store_pair_rows_at(double *ubase, unsigned off, double r0, double r1) {
    double a, b;
    pair_exchange<ASM>(r0, r1, a, b);
    store2_at<ST>(ubase, off, a, b);
}

// One chunk of U consecutive samples of the lane's voice to the out[n*V + v] stream; `op` = out + n*V + v, advanced by U rows.
// PX: as 16-byte pair rows (store_pair_rows) -- V even, out 16-byte aligned, both lanes of every pair live on the same chunk, lane parity =
// voice parity; px_store (wave-uniform): 1 / 2 / 3 = plain / write-through / non-temporal stores.  !PX: 8-byte stores.
template <bool PX, int U>
__device__ __forceinline__ void emit_chunk(double *&op, size_t V, const double (&o)[U], int px_store) {
    if constexpr (PX) {
        double *pp = op + ((threadIdx.x & 1) ? V - 1 : 0);  // this lane's 16 bytes of row n + (lane & 1): out + (n + odd) * V + (v & ~1)
        if (px_store == 2) {
#pragma unroll
            for (int j = 0; j < U / 2; j++) store_pair_rows<2>(pp + (size_t)(2 * j) * V, o[2 * j], o[2 * j + 1]);
        } else if (px_store == 3) {
#pragma unroll
            for (int j = 0; j < U / 2; j++) store_pair_rows<1>(pp + (size_t)(2 * j) * V, o[2 * j], o[2 * j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < U / 2; j++) store_pair_rows<0>(pp + (size_t)(2 * j) * V, o[2 * j], o[2 * j + 1]);
        }
        op += (size_t)U * V;
    } else {
#pragma unroll
        for (int i = 0; i < U; i++) {
            *op = o[i];
            op += V;
        }
    }
}

// The mirror image for a READ stream in[n*V + v]: the lane loads 16 bytes -- two voices of its row n + (lane & 1), from
// in + (n + (lane & 1)) * V + (v & ~1) -- and the pair makes the same exchange, after which every lane holds samples n (r0) and n + 1
// (r1) of ITS voice (even lane: raw = (x[n][2k], x[n][2k+1]) keeps .x and gets the partner's .x = x[n+1][2k]; odd lane: raw =
// (x[n+1][2k], x[n+1][2k+1]) keeps .y and gets the partner's .y).  The load (`raw`) and the swap are separate so that a kernel can
// request a chunk ahead and swap when it consumes.
__device__ __forceinline__ void pair_rows_swap(const double2v raw, double &r0, double &r1) { pair_exchange(raw.x, raw.y, r0, r1); }

}  // namespace mxg
