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
int ensure_init();
hipStream_t resolve_stream(void *stream);
int tune_get(const char *key);

// Library-owned scratch, one grow-only buffer per (slot, stream): launches on different streams never share
// (or resize) each other's temporaries; launches on one stream are ordered by the stream.  Returns MXG_OK and
// a device pointer of at least `bytes`.
enum ScratchSlot { SCR_MIX_GAINS, SCR_OSC_MIX, SCR_GRAIN_ERR, SCR_GRAIN_SCHED, SCR_IFFT_OUT, SCR_IFFT_BUF, SCR_MFCC_RAW, SCR_CONVOLVE, SCR_GRAIN_MIX, SCR_PART_SYNC, SCR_SLOTS };
int scratch_get(ScratchSlot slot, hipStream_t st, size_t bytes, void **out);
// the per-wavefront counters of the time-part kernels (part_signal / part_wait below): zero when handed out for the first time,
// and every launch leaves them zero again
int part_counters_get(hipStream_t st, size_t wavefronts, int **out);

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

// ---- time parts: who may overwrite the state ----------------------------------------------------------------------
// A kernel cut into gridDim.y time parts reads its per-voice state in EVERY part and stores the new state from ONE of them (the
// writer).  Nothing orders the workgroups of a launch: with little work per part the writer can be done before another part of the
// same voices has even started, and that part would then start from the NEW state.  So every other part signals, per wavefront,
// once its state loads have returned, and the writer waits for those signals before it stores.  Workgroups are dispatched in
// order (x fastest, then y) and the writer is the part with the most work in front of its store, so the wait is normally over
// before it begins; it is bounded anyway (a kernel never hangs on a stale counter).  The writer leaves the counter at zero.
__device__ __forceinline__ int *part_counter(int *ctrs) {
    return ctrs ? ctrs + ((size_t)blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6)) : nullptr;
}
__device__ __forceinline__ void part_signal(int *ctr) {
    if (!ctr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the state is in registers
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((unsigned long long)__ballot(1)) - 1))
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void part_wait(int *ctr, int others) {
    if (!ctr) return;
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((unsigned long long)__ballot(1)) - 1)) {
        for (int spins = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < others && spins < (1 << 20); spins++)
            __builtin_amdgcn_s_sleep(8);
        __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- device store helpers ---------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void store1(double *p, double v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
typedef double double2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void store2(double *p, double a, double b) {
    double2v v = {a, b};
    if constexpr (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<double2v *>(p));
    else
        *reinterpret_cast<double2v *>(p) = v;
}

}  // namespace mxg
