// mxg_gate.h -- a gate / trigger signal shared by a whole bank, handed to the lanes with v_readlane
// (used by the maxiEnv, fused-voice and maxiEnvGen kernels).
#pragma once
#include "mxg_common.h"

namespace mxg {
namespace {

// A gate shared by the whole bank, 64 chunks of U samples at a time: lane j of every wavefront keeps the U gate
// values of chunk j of the group and their class (gate_of_chunk), and the kernels pick chunk cc's with v_readlane.
// This replaces U clamped scalar loads + the class arithmetic per chunk (about 140 scalar instructions, a quarter of
// the steady-state voice kernel's issue slots at one wavefront per SIMD) by one readlane per chunk.
// The readlanes need all 64 lanes of a wavefront alive, so the surplus lanes of the bank's last wavefront do not
// exit: they shadow voice V-1 -- the same loads, the same arithmetic and the same stores of the same values to the same
// addresses as the lane that owns it.
__device__ __forceinline__ size_t live_voice(size_t gid, size_t V) { return gid < V ? gid : V - 1; }

template <int U, typename T>
struct GateGroup {
    T g[U];
    int cls;  // +1: pred holds for every gate of the chunk, -1: for none, 0: mixed or not a full chunk
};
// A block of up to 64*U samples (the default 512) needs one group, loaded in the kernel prologue; longer launches
// reload at every group boundary and pay one drain of the store stream there (loads and stores share one in-order
// counter, and a prefetch consumed 64 chunks later cannot be expressed as a counted wait).
template <int U, typename T, typename Pred>
__device__ __forceinline__ void gate_group_load(GateGroup<U, T> &G, const T *__restrict__ trig, size_t N, size_t grp,
                                                Pred pred) {
    const size_t c = grp * 64 + (threadIdx.x & 63);
    int on = 0;
#pragma unroll
    for (int i = 0; i < U; i++) {
        const size_t m = c * U + i;
        G.g[i] = trig[m < N ? m : N - 1];  // clamped: a surplus lane re-reads the last gate, its class is 0
        on += pred(G.g[i]) ? 1 : 0;
    }
    G.cls = (c * U + U <= N) ? (on == U ? 1 : (on == 0 ? -1 : 0)) : 0;
}
__device__ __forceinline__ int lane_value(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
__device__ __forceinline__ double lane_value(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                            __builtin_amdgcn_readlane(__double2loint(x), lane));
}

}  // namespace
}  // namespace mxg
