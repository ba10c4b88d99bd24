// mxg_lanefold.h -- transposing lane reductions: 16 wave-wide vectors (one per sample) folded to one vector that holds
// 16 sums.  Used by K1m (osc.hip: the fused maxiMix::stereo mixdown of a voice bank) and by the granular unit kernel
// (grains.hip: the mixdown of a stream tile).  Issue costs measured on gfx950 (tools/ubench): any 32-bit VALU op incl. a
// DPP move 4.35 clk per wave64 at one wave per SIMD, an fp64 add 4.35, v_permlane16/32_swap 16, ds_bpermute 24.
#pragma once
#include "mxg_common.h"

namespace mxg {
namespace {

constexpr int kMixChunk = 16;

template <typename T> struct Fold;
template <> struct Fold<double> {
    static __device__ __forceinline__ double join(double x, double y) { return x + y; }
    static __device__ __forceinline__ void split(double v, unsigned &lo, unsigned &hi) { lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v); }
    static __device__ __forceinline__ double make(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
};

// DPP exchange: t = x with the lanes of `BANKS_T` replaced by ctrl(y); u = y with the other banks replaced by ctrl(x)
template <int CTRL, int BANKS_T>
__device__ __forceinline__ void dpp_exchange(unsigned x, unsigned y, unsigned &t, unsigned &u) {
    t = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)y, CTRL, 0xf, BANKS_T, false);
    u = (unsigned)__builtin_amdgcn_update_dpp((int)y, (int)x, CTRL, 0xf, 0xf ^ BANKS_T, false);
}
// banks BANKS_T of the result carry sample B (its lanes added to their mirror partners), the others sample A
template <int CTRL, int BANKS_T>
__device__ __forceinline__ double fold_dpp(double a, double b) {
    unsigned al, ah, bl, bh, tl, th, ul, uh;
    Fold<double>::split(a, al, ah);
    Fold<double>::split(b, bl, bh);
    dpp_exchange<CTRL, BANKS_T>(al, bl, tl, ul);
    dpp_exchange<CTRL, BANKS_T>(ah, bh, th, uh);
    return Fold<double>::make(tl, th) + Fold<double>::make(ul, uh);
}
template <int CTRL, int BANKS_T>
__device__ __forceinline__ int fold_dpp(int a, int b) {
    unsigned t, u;
    dpp_exchange<CTRL, BANKS_T>((unsigned)a, (unsigned)b, t, u);
    return (int)t == (int)u ? (int)t : -1;  // both operands must name the same sample
}
// lanes with `bit` set keep sample B, the others sample A; each adds the partner lane (quad_perm QP) of its own sample
template <int QP>
__device__ __forceinline__ double fold_quad(double a, double b, bool bit) {
    const double keep = bit ? b : a, send = bit ? a : b;
    unsigned lo, hi;
    Fold<double>::split(send, lo, hi);
    const double got = Fold<double>::make((unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, QP, 0xf, 0xf, true),
                                          (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, QP, 0xf, 0xf, true));
    return keep + got;
}
template <int QP>
__device__ __forceinline__ int fold_quad(int a, int b, bool bit) {
    const int keep = bit ? b : a, send = bit ? a : b;
    const int got = __builtin_amdgcn_update_dpp(0, send, QP, 0xf, 0xf, true);
    return keep == got ? keep : -1;
}
constexpr int kDppRowMirror = 0x140, kDppRowHalfMirror = 0x141;
constexpr int kDppQuadXor1 = 0xB1 /* [1,0,3,2] */, kDppQuadXor2 = 0x4E /* [2,3,0,1] */;
// 16 vectors (one per sample) -> one vector: lane l holds, for sample slot(l), the sum over the 16 lanes of its row
template <typename T>
__device__ __forceinline__ T fold_chunk(const T (&v)[kMixChunk], int lane) {
    T l1[8], l2[4], l3[2];
#pragma unroll
    for (int j = 0; j < 8; j++) l1[j] = fold_dpp<kDppRowMirror, 0xC>(v[2 * j], v[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; j++) l2[j] = fold_dpp<kDppRowHalfMirror, 0xA>(l1[2 * j], l1[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 2; j++) l3[j] = fold_quad<kDppQuadXor2>(l2[2 * j], l2[2 * j + 1], (lane & 2) != 0);
    return fold_quad<kDppQuadXor1>(l3[0], l3[1], (lane & 1) != 0);
}

// ---- variant 0: cross-row levels first with v_permlane32/16_swap, then mirror / half-mirror, then a quad reduction:
// every quad of lanes ends up with the wave-wide sum of one sample (no row sums for the workgroup pass to add)
__device__ __forceinline__ double fold32(double a, double b) {
    unsigned al, ah, bl, bh;
    Fold<double>::split(a, al, ah);
    Fold<double>::split(b, bl, bh);
    auto lo = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    return Fold<double>::make(lo[0], hi[0]) + Fold<double>::make(lo[1], hi[1]);
}
__device__ __forceinline__ int fold32(int a, int b) {
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    return (int)r[0] == (int)r[1] ? (int)r[0] : -1;
}
__device__ __forceinline__ double fold16(double a, double b) {
    unsigned al, ah, bl, bh;
    Fold<double>::split(a, al, ah);
    Fold<double>::split(b, bl, bh);
    auto lo = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
    return Fold<double>::make(lo[0], hi[0]) + Fold<double>::make(lo[1], hi[1]);
}
__device__ __forceinline__ int fold16(int a, int b) {
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    return (int)r[0] == (int)r[1] ? (int)r[0] : -1;
}
__device__ __forceinline__ double quad_sum(double v) {
    unsigned lo, hi;
    Fold<double>::split(v, lo, hi);
    double o = Fold<double>::make((unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, kDppQuadXor1, 0xf, 0xf, true),
                                  (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, kDppQuadXor1, 0xf, 0xf, true));
    v = v + o;
    Fold<double>::split(v, lo, hi);
    o = Fold<double>::make((unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, kDppQuadXor2, 0xf, 0xf, true),
                           (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, kDppQuadXor2, 0xf, 0xf, true));
    return v + o;
}
template <typename T>
__device__ __forceinline__ T fold_chunk_swap(const T (&v)[kMixChunk]) {
    T l1[8], l2[4], l3[2];
#pragma unroll
    for (int j = 0; j < 8; j++) l1[j] = fold32(v[2 * j], v[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; j++) l2[j] = fold16(l1[2 * j], l1[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 2; j++) l3[j] = fold_dpp<kDppRowMirror, 0xC>(l2[2 * j], l2[2 * j + 1]);
    return fold_dpp<kDppRowHalfMirror, 0xA>(l3[0], l3[1]);
}

// ---- the [16 samples][64 voices] mixdown tile of K1m (osc.hip) and K2f's mixdown form (voice.hip) ---------------------------------
// Voice quarter q, sample s, voice j of the quarter at double q * kTileQuarter + s * kTileRow + j: the row stride of 144 B moves
// consecutive samples by nine 16-byte bank groups (see osc.hip, K1m).
constexpr int kTileRow = 18;                 // doubles per (quarter, sample) row: 16 voices + 2 of padding (144 B)
constexpr int kTileQuarter = 16 * kTileRow;  // 288 doubles = 2304 B
constexpr int kTileWave = 4 * kTileQuarter;  // 1152 doubles = 9 KB per wavefront
constexpr int kPcRing = 3;                   // tiles per producer / consumer pair
__device__ __forceinline__ int lds_flag_load(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// One combine window of a producer / consumer workgroup (8 wavefronts): the four consumer rows s_part[4][WIN][2] of `span` samples are
// added left to right into the workgroup's row `prow`; both barriers are met by all eight wavefronts.
template <int WIN>
__device__ __forceinline__ void mixpc_window_close(const double *s_part, double *prow, int span) {
    __syncthreads();
    for (int i = threadIdx.x; i < span * 2; i += blockDim.x)
        prow[i] = ((s_part[i] + s_part[WIN * 2 + i]) + s_part[2 * WIN * 2 + i]) + s_part[3 * WIN * 2 + i];
    __syncthreads();
}

// The CONSUMER wavefront of a pair over a whole block of N samples (the arithmetic of osc_mixpc_kernel's consumer branch, the same
// order of additions: C:503-509 per voice, the user's `mix +=` over voices as a fixed tree).  ring: the pair's kPcRing tiles; f_prod /
// f_cons: the pair's counters; gl / gr: the gains of the 16 voices this lane sums; wg_rows: this workgroup's rows [N][2].
template <int WIN, int SLEEP>
__device__ __forceinline__ void mixpc_consume(size_t N, double *ring, int *f_prod, int *f_cons, const double (&gl)[16], const double (&gr)[16],
                                              double *s_part, double *my_part, double *s_dump, double *wg_rows) {
    const int lane = threadIdx.x & 63, ts = lane & 15, tq = lane >> 4;
    int k = 0;
    for (size_t n0 = 0; n0 < N; n0 += WIN) {
        const int span = (int)((N - n0) < (size_t)WIN ? (N - n0) : (size_t)WIN);
        for (int c0 = 0; c0 < span; c0 += kMixChunk, k++) {
            const int cnt = (span - c0) < kMixChunk ? (span - c0) : kMixChunk;
            while (lds_flag_load(f_prod) <= k) __builtin_amdgcn_s_sleep(SLEEP);
            asm volatile("" ::: "memory");
            const double2v *tr = reinterpret_cast<const double2v *>(ring + (k % kPcRing) * kTileWave + tq * kTileQuarter + ts * kTileRow);
            double2v xv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) xv[j] = tr[j];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            lds_flag_store(f_cons, k + 1);  // (behind the tile reads in the LDS queue)
            double pl[8], pr[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                pl[j] = xv[j].x * gl[2 * j] + xv[j].y * gl[2 * j + 1];
                pr[j] = xv[j].x * gr[2 * j] + xv[j].y * gr[2 * j + 1];
            }
            const double sl = ((pl[0] + pl[1]) + (pl[2] + pl[3])) + ((pl[4] + pl[5]) + (pl[6] + pl[7]));
            const double sr2 = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
            double t = fold32(sl, sr2);
            t = fold16(t, t);
            // lanes 0-15 hold the left sums of samples c0 + lane, lanes 32-47 the right sums; the others drop theirs in a scratch row
            double *dst = ((lane & 16) == 0 && ts < cnt) ? my_part + (c0 + ts) * 2 + (lane >> 5) : s_dump + (threadIdx.x & 255);
            *dst = t;
        }
        mixpc_window_close<WIN>(s_part, wg_rows + n0 * 2, span);
    }
}

// mix[i] = sum over workgroups of partial[g][i], i = n*2 + ch.  A workgroup owns 64 consecutive elements (one coalesced
// 512-B row segment per load); its 16 waves each add the groups g = w, w+16, ... in order (independent loads, all in
// flight), then the 16 wave sums are combined left to right: a fixed order for a fixed number of workgroups.  blockIdx.y = block k of
// a batch (the mix queue's fold: partial[k][g][i] -> mix[k][i]).
constexpr int kPartWaves = 16;
__global__ __launch_bounds__(64 * kPartWaves) void mix_partials_kernel(size_t ngroups, size_t count,
                                                                       const double *__restrict__ partial,
                                                                       double *__restrict__ mix) {
    __shared__ double s_red[kPartWaves][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    double s = 0.0;
    partial += (size_t)blockIdx.y * ngroups * count;
    mix += (size_t)blockIdx.y * count;
    if (i < count) {
        const double *p = partial + i;
#pragma unroll 8
        for (size_t g = w; g < ngroups; g += kPartWaves) s += p[g * count];
    }
    s_red[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < count) {
        double t = s_red[0][lane];
#pragma unroll
        for (int k = 1; k < kPartWaves; k++) t += s_red[k][lane];
        mix[i] = t;
    }
}

// The same fold for a SMALL, fixed number of groups (the granular tile renders: one row per 64 streams), one lane per element and no
// workgroup step: all NG loads of a lane are in flight together, and the additions are the ones mix_partials_kernel makes in the
// order it makes them (wave w's chain 0.0 + p[w] + p[w + 16] + ..., then the sixteen chains left to right) -- the same bits.
// 2048 streams x 70 560 samples: 15.0 -> ~8 us (the 1024-lane workgroups of the general kernel were mostly launch cost there).
template <int NG>
__global__ __launch_bounds__(256) void mix_partials_flat_kernel(size_t count, const double *__restrict__ partial, double *__restrict__ mix) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double v[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) v[g] = partial[(size_t)g * count + i];
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kPartWaves; w++) {
        double s = 0.0;
#pragma unroll
        for (int g = w; g < NG; g += kPartWaves) s += v[g];
        t = w == 0 ? s : t + s;
    }
    mix[i] = t;
}
// ngroups x count -> count on stream st with whichever of the two kernels fits
inline void mix_partials_launch(hipStream_t st, size_t ngroups, size_t count, const double *partial, double *mix) {
    const dim3 g((unsigned)((count + 255) / 256)), b(256);
    if (ngroups == 16) {
        hipLaunchKernelGGL((mix_partials_flat_kernel<16>), g, b, 0, st, count, partial, mix);
    } else if (ngroups == 32) {
        hipLaunchKernelGGL((mix_partials_flat_kernel<32>), g, b, 0, st, count, partial, mix);
    } else if (ngroups == 48) {
        hipLaunchKernelGGL((mix_partials_flat_kernel<48>), g, b, 0, st, count, partial, mix);
    } else if (ngroups == 64) {
        hipLaunchKernelGGL((mix_partials_flat_kernel<64>), g, b, 0, st, count, partial, mix);
    } else {
        hipLaunchKernelGGL(mix_partials_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * kPartWaves), 0, st, ngroups, count, partial, mix);
    }
}

}  // namespace
}  // namespace mxg
