// osc.hip -- maxiOsc voice bank on gfx950.
//
// Path: maxiOsc::{sinewave,coswave,phasor,saw,triangle,square,pulse,impulse,sinebuf,
// sinebuf4,sawn,phasorBetween} (reference src/maximilian.cpp:228-373, cited per function
// as C:line).  The reference advances ONE oscillator by ONE sample per call; here one
// wavefront lane owns one voice (or two adjacent voices) and walks the whole block of N
// samples with its phase in VGPRs; the 514-point sine table / 1001-point transition table
// are staged once per workgroup into LDS.  Every sample is stored straight to
// out[n*V + v]: a wavefront store covers 512 B (VPL=1) or 1 KiB (VPL=2) of one row, so
// the only mandatory HBM traffic is the 8 B/sample output stream (kernel K1).
//
// Numerics: the expression trees are the reference's, op for op, in fp64, compiled with
// -ffp-contract=off.  +,-,*,/ and floor are IEEE-exact on gfx950, so every waveform except
// sinewave/coswave is bit-identical to the reference; those two go through mxg_sincos.h.
#include <type_traits>

#include "mxg_common.h"
#include "maxi_tables.h"
#include "mxg_sincos.h"
#include "mxg_osc.h"
#include "mxg_lanefold.h"

namespace mxg {

namespace {

// Device-resident copies of the two static tables (with their guard elements).
__device__ const double MAXI_SINE_TAB_D[MAXI_SINE_TAB_LEN] = MAXI_SINE_TAB_INIT;
__device__ const double MAXI_TRANS_TAB_D[MAXI_TRANS_TAB_LEN] = MAXI_TRANS_TAB_INIT;

__device__ const double MXG_SINTAB_D[MXG_SINTAB_LEN] = {MXG_SINTAB_VALUES};

template <int WF>
constexpr bool uses_sine() {
    return WF == MXG_OSC_SINEBUF || WF == MXG_OSC_SINEBUF4;
}
// the table a waveform's tick reads from LDS: sineBuffer (sinebuf / sinebuf4), transition (sawn), sin / cos of k*pi/256 (sinewave / coswave)
static_assert((kSineOddOff & 1) == 1 && kSineOddOff >= MAXI_SINE_TAB_LEN, "parity copy of the sine table");
template <int WF>
constexpr int tab_len() {
    return (WF == MXG_OSC_SINEBUF4 && MXG_SB4_PAIRS) ? kSineOddOff + MAXI_SINE_TAB_LEN + 1 : uses_sine<WF>() ? MAXI_SINE_TAB_LEN
                           : (WF == MXG_OSC_SAWN ? MAXI_TRANS_TAB_LEN
                                                 : ((WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) ? MXG_SINTAB_LEN : 1));
}
template <int WF>
__device__ __forceinline__ void load_tab(double *s_tab) {  // (the caller synchronises)
    if constexpr (uses_sine<WF>()) {
        for (int i = threadIdx.x; i < MAXI_SINE_TAB_LEN; i += blockDim.x) {
            s_tab[i] = MAXI_SINE_TAB_D[i];
            if constexpr (WF == MXG_OSC_SINEBUF4 && MXG_SB4_PAIRS) s_tab[kSineOddOff + i] = MAXI_SINE_TAB_D[i];  // the parity copy (osc_tick)
        }
    } else if constexpr (WF == MXG_OSC_SAWN) {
        for (int i = threadIdx.x; i < MAXI_TRANS_TAB_LEN; i += blockDim.x) s_tab[i] = MAXI_TRANS_TAB_D[i];
    } else if constexpr (WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) {
        for (int i = threadIdx.x; i < MXG_SINTAB_LEN; i += blockDim.x) s_tab[i] = MXG_SINTAB_D[i];
    }
}

// K1: one lane = VPL adjacent voices; the N-sample recurrence runs in registers.
// ST: store flavour (mxg_common.h: 0 plain, 1 nt, 2 sc1).  PX (VPL = 1): two samples per lane pair leave as one 16-byte store per
// lane (store_pair_rows) -- every SIMD keeps its own wavefront of voices, and the store stream is the 16-byte one that the
// calibration kernels measure fastest.
template <int WF, bool FPS, int VPL, int ST, bool PX>
__global__ void osc_kernel(size_t V, size_t N, const double *__restrict__ freq,
                           const double *__restrict__ p1, const double *__restrict__ p2,
                           double *__restrict__ phase_io, double *__restrict__ hold_io,
                           double *__restrict__ out, double sr, PartSync psync, int xcd, int p1ps) {
    // (p1ps, FPS only: p1 is [N][V] too -- a pulse width / start phase per sample, for the per-sample engine's derived arguments)
    // All LDS in ONE array (a second __shared__ object perturbs hipcc's waitcnt placement).
    __shared__ __attribute__((aligned(16))) double s_tab[tab_len<WF>()];
    if constexpr (tab_len<WF>() > 1) {
        load_tab<WF>(s_tab);
        __syncthreads();
    }
    const size_t v0 = ((size_t)xcd_block(blockIdx.x, gridDim.x, xcd) * blockDim.x + threadIdx.x) * VPL;
    if (v0 >= V) return;

    double ph[VPL], hd[VPL];
    OscPre q[VPL];
#pragma unroll
    for (int j = 0; j < VPL; j++) {
        ph[j] = phase_io[v0 + j];
        hd[j] = hold_io[v0 + j];
        double a = p1 ? p1[v0 + j] : 0.0, b = p2 ? p2[v0 + j] : 0.0;
        if constexpr (!FPS) {
            q[j] = osc_pre<WF>(freq[v0 + j], sr, a, b);
        } else {
            q[j].p1 = a;
            q[j].p2 = b;
        }
    }
    // Time split (gridDim.y parts): 65 536 voices are one wavefront per SIMD, and a lone wavefront issues one VALU
    // instruction per ~4.4 clk whatever it is (tools/ubench: fp64 4.35 clk at one wave per SIMD, 2.27 at four).  For the
    // waveforms whose OUTPUT is the expensive part (sinewave / coswave: ~100 fp64 ops per sample against 3 for the phase
    // ramp) part p first advances the recurrence over the samples of parts 0..p-1 without producing them -- the same
    // additions in the same order, so the same bits -- and then renders its own stretch; the last part stores the state.
    // The last part stores the state; the others tell it when they have read theirs (part_signal / part_wait, mxg_common.h).
    int *const part_ctr = gridDim.y > 1 ? part_counter(psync) : nullptr;
    if (blockIdx.y + 1 != gridDim.y) part_signal(part_ctr);
    const size_t plen = (N + gridDim.y - 1) / gridDim.y;
    const size_t nA = blockIdx.y * plen < N ? blockIdx.y * plen : N;
    const size_t nB = nA + plen < N ? nA + plen : N;
    if constexpr (!FPS) {
#pragma unroll 4
        for (size_t n = 0; n < nA; n++) {
#pragma unroll
            for (int j = 0; j < VPL; j++) osc_skip<WF>(ph[j], hd[j], q[j], s_tab, s_tab);
        }
    }
    double *o = out + nA * V + v0;
    const double *fp = freq + v0;
    const double *pp = (FPS && p1ps) ? p1 + v0 : nullptr;
#ifndef MXG_OSC_UNROLL
#define MXG_OSC_UNROLL 4
#endif
    // sinewave / coswave with 0 <= inc <= 1 and the phase in [0, 2] on every lane of the wavefront (any audio frequency from a
    // fresh or carried bank): the argument of sin / cos stays in [0, 4 pi], so the table routine needs neither its range test nor
    // a sign (mxg_sincos.h, TRUST)
    bool trust = false;
    if constexpr ((WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) && !FPS) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < VPL; j++) ok = ok && q[j].inc >= 0.0 && q[j].inc <= 1.0 && ph[j] >= 0.0 && ph[j] <= 2.0;
        trust = __all(ok);
    }
    auto run = [&](auto trust_tag) {
        constexpr bool kTrust = decltype(trust_tag)::value;
        size_t n = nA;
        if constexpr (PX && VPL == 1 && !FPS) {
            double *op = out + (nA + (threadIdx.x & 1)) * V + (v0 & ~(size_t)1);
#pragma unroll 2
            for (; n + 2 <= nB; n += 2) {
                const double r0 = osc_tick<WF, kTrust>(ph[0], hd[0], q[0], s_tab, s_tab);
                const double r1 = osc_tick<WF, kTrust>(ph[0], hd[0], q[0], s_tab, s_tab);
                store_pair_rows<ST>(op, r0, r1);
                op += 2 * V;
            }
            o = out + n * V + v0;
        }
#pragma unroll MXG_OSC_UNROLL
        for (; n < nB; n++) {
            double r[VPL];
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                if constexpr (FPS) q[j] = osc_pre<WF>(fp[j], sr, pp ? pp[j] : q[j].p1, q[j].p2);
                r[j] = osc_tick<WF, kTrust>(ph[j], hd[j], q[j], s_tab, s_tab);
            }
            if constexpr (VPL == 2)
                store2<ST>(o, r[0], r[1]);
            else
                store1<ST>(o, r[0]);
            o += V;
            if constexpr (FPS) {
                fp += V;
                if (pp) pp += V;
            }
        }
    };
    if constexpr ((WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) && !FPS) {
        if (trust) run(std::true_type{}); else run(std::false_type{});
    } else {
        run(std::false_type{});
    }
    if (blockIdx.y + 1 == gridDim.y && part_wait(part_ctr, psync)) {
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            phase_io[v0 + j] = ph[j];
            hold_io[v0 + j] = hd[j];
        }
    }
}

// ---- K1m: K1 + fused maxiMix::stereo partial sums ---------------------------------------------------
// Same per-voice recurrence and (optional) per-voice store as K1; in addition every wavefront reduces its 64 voices'
// panned samples (in*sqrt(1-x), in*sqrt(x), C:503-509) per sample, so the mixdown never re-reads the 268 MB block from
// HBM.  The reduction is a transposing butterfly in registers: each level takes two vectors, exchanges complementary
// lanes between them and adds, so the number of vectors halves while every vector carries twice as many samples.
// Measured issue costs on gfx950 (tools/ubench): any 32-bit VALU op incl. a DPP move 4.35 clk per wave64, an fp64
// add/mul 4.35, v_permlane16/32_swap 16, ds_bpermute 24.  So the four transposing levels for a chunk of 16 samples use
// the exchanges that stay inside a row of 16 lanes (DPP), largest level first where the pairs are most numerous:
//   level 1  row_mirror      (lane i <-> 15-i, bank-masked DPP moves)   16 vectors -> 8     5 ops per fold
//   level 2  row_half_mirror (lane i <-> 7-i)                            8 -> 4            5 ops
//   level 3  quad_perm xor 2 (select + DPP)                              4 -> 2            7 ops
//   level 4  quad_perm xor 1                                             2 -> 1            7 ops
// = 81 VALU ops per channel and 16 samples (the first version exchanged across rows with v_permlane swaps: 48 of them,
// 768 clk per chunk, VALU-bound at 48 us against 42 us for K1).  After level 4 lane l holds, for one of the 16 samples
// (which one: the same network run once on the sample indices, `slot`), the sum over ITS ROW of 16 voices; the four row
// sums are not combined in registers at all: every lane stores its (L, R) pair to LDS [wave][row][sample] and the
// workgroup's combine pass -- which has to add the four waves anyway -- adds 16 terms instead of 4, in a fixed order.
// A small second kernel sums the per-workgroup partials => deterministic (but not the reference's sequential order:
// tolerance on the mix).

// the lane-exchange helpers (Fold, fold_dpp, fold_quad, fold_chunk, fold32/16, quad_sum, fold_chunk_swap) live in mxg_lanefold.h

// STORE: 0 = mix only (no per-voice block), 1 = plain 8-byte stores, 2 = pair rows of write-through 16-byte stores (as K1, V even and
// `out` 16-byte aligned).  VAR 0: permlane-swap butterfly, 4 LDS rows per window (one per wave).  VAR 1: all-DPP butterfly, 16
// rows (wave x row; A/B only).  VAR 2 (round 3): the CROSS-ROW half of the reduction on the matrix pipe.  v_mfma_f64_4x4x4 (four
// 4 x 4 x 4 blocks, 16 cycles) computes D[b][i][j] += sum_k A[b][i][k] B[b][k][j]; the operand layout, probed on the device
// (tools/ubench/mfma_probe.hip): lane l = 16 k + 4 b + c supplies A[b][i = c][k] and B[b][k][j = c], and lane 16 i + 4 b + j receives
// D[b][i][j] -- k is the ROW of 16 lanes.  Give it B = the lane's own product (x * gain: no data movement) and A = [c == s & 3]
// for sample s, accumulate four samples into one register, and lane (row i, column q) holds the sum of sample 4 g + i over lanes
// q, q + 16, q + 32, q + 48: the two levels that cost 48 v_permlane swaps of 16 clk per chunk and channel run beside the
// oscillator's VALU work (two 16-cycle MFMAs per sample on an otherwise idle pipe), and the transposition comes for free.  What
// is left is the sum over the 16 lanes of a row for four vectors: row_mirror, row_half_mirror, quad sum (21 VALU per chunk and
// channel instead of ~110).  The MFMA adds the four products in its own order: covered by the mix tolerance (the sum was
// tree-ordered already).  (v_mfma_f64_16x16x4 does the same job with a 16-row selector at 64 cycles per instruction: measured
// 85 us -- at one wavefront per SIMD the in-order issue waits for the busy matrix pipe.)  Time parts (gridDim.y, round 3): the kernel is VALU-issue bound at one wavefront per SIMD (43 us
// of arithmetic for a 65 536 x 512 block against ~41 us of stores), and a second resident wavefront nearly doubles the issue
// rate -- so a block is cut into two time parts like K1's sinewave: part p advances the phase over the samples before it with
// osc_skip (the same additions: the same bits), renders its stretch and mixes it into its own rows of the partial buffer; the
// last part stores the state (part_signal / part_wait).
template <int WF, int STORE, int VAR, int WIN>
__global__ __launch_bounds__(256) void osc_mix_kernel(size_t V, size_t N, const double *__restrict__ freq,
                                                      const double *__restrict__ p1, const double *__restrict__ p2,
                                                      double *__restrict__ phase_io, double *__restrict__ hold_io,
                                                      double *__restrict__ out, const double *__restrict__ pan,
                                                      double *__restrict__ partial, double sr, PartSync psync) {
    constexpr int kTab = tab_len<WF>();
    constexpr int kTabPad = (kTab + 1) & ~1;  // the (L, R) pairs below are 16-byte stores
    constexpr int kRows = VAR == 1 ? 16 : 4;  // LDS rows the workgroup pass adds per output
    constexpr int kMixWin = WIN;
    __shared__ __attribute__((aligned(16))) double s_all[kTabPad + kRows * kMixWin * 2];
    double *s_tab = s_all;
    load_tab<WF>(s_tab);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *s_part = s_all + kTabPad;                                     // [kRows][kMixWin][2]
    double *my_part = s_part + (VAR != 1 ? wave : wave * 4 + (lane >> 4)) * (kMixWin * 2);
    // The lane exchanges need all 64 lanes alive, and a per-sample `if (live)` costs an exec-mask region per sample:
    // the surplus lanes of the bank's last wavefront shadow a live voice instead (same loads, same arithmetic, same stores
    // of the same values to the same addresses) and enter the mix with zero gains -- voice V-1, or with pair rows the last
    // PAIR of voices, parity kept, so that they exchange among themselves.
    const size_t vraw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = vraw < V;
    const size_t v = live ? vraw : (STORE == 2 ? V - 2 + (vraw & 1) : V - 1);
    double ph = phase_io[v], hd = hold_io[v];
    double x = pan[v];
    if (x > 1) x = 1;  // C:504
    if (x < 0) x = 0;  // C:505
    double gl = live ? sqrt(1.0 - x) : 0.0, gr = live ? sqrt(x) : 0.0;
    OscPre q = osc_pre<WF>(freq[v], sr, p1 ? p1[v] : 0.0, p2 ? p2[v] : 0.0);
    // Consume the prologue loads HERE: otherwise hipcc's waitcnt pass keeps them "pending" at the loop back-edge and
    // drains the asynchronous output stores with s_waitcnt vmcnt(0) every chunk.
    asm volatile("" : "+v"(ph), "+v"(hd), "+v"(gl), "+v"(gr));
    asm volatile("" : "+v"(q.inc), "+v"(q.k), "+v"(q.p1), "+v"(q.p2));
    // time parts: this part renders [nA, nB); part lengths are whole mix chunks, so a chunk never straddles two parts
    int *const part_ctr = gridDim.y > 1 ? part_counter(psync) : nullptr;
    if (blockIdx.y + 1 != gridDim.y) part_signal(part_ctr);
    const size_t plen = ((N + gridDim.y - 1) / gridDim.y + kMixChunk - 1) / kMixChunk * kMixChunk;
    const size_t nA = blockIdx.y * plen < N ? blockIdx.y * plen : N;
    const size_t nB = nA + plen < N ? nA + plen : N;
#pragma unroll 4
    for (size_t n = 0; n < nA; n++) osc_skip<WF>(ph, hd, q, s_tab, s_tab);
    // which sample of a chunk this lane ends up holding: the same network on the sample indices
    int slot;
    {
        int idx[kMixChunk];
#pragma unroll
        for (int i = 0; i < kMixChunk; i++) idx[i] = i;
        if constexpr (VAR == 2) {  // register r of the MFMA result holds sample 4 r + (lane >> 4); then the in-row network
            const int q = lane >> 4;
            slot = fold_dpp<kDppRowHalfMirror, 0xA>(fold_dpp<kDppRowMirror, 0xC>(q, 4 + q), fold_dpp<kDppRowMirror, 0xC>(8 + q, 12 + q));
        } else {
            slot = VAR == 0 ? fold_chunk_swap<int>(idx) : fold_chunk<int>(idx, lane);
        }
        if (VAR != 1 && (lane & 3) != 0) slot = -1;  // one lane per quad stores
    }
    double *o = out + nA * V + v;
    double *op = out + (nA + (threadIdx.x & 1)) * V + (v & ~(size_t)1);  // pair rows: this lane's 16 bytes of row n + (lane & 1)
    for (size_t n0 = nA; n0 < nB; n0 += kMixWin) {
        const int span = (int)((nB - n0) < (size_t)kMixWin ? (nB - n0) : (size_t)kMixWin);
        for (int c0 = 0; c0 < span; c0 += kMixChunk) {
            const int cnt = (span - c0) < kMixChunk ? (span - c0) : kMixChunk;
            auto chunk = [&](auto full_tag) {
                constexpr bool kFull = decltype(full_tag)::value;
                double L[kMixChunk], R[kMixChunk];
                if constexpr (VAR == 2) {
                    double DL[4] = {0.0, 0.0, 0.0, 0.0}, DR[4] = {0.0, 0.0, 0.0, 0.0};
                    const int l4 = lane & 3;
#pragma unroll
                    for (int i = 0; i < kMixChunk; i += 2) {
                        double r0 = 0.0, r1 = 0.0;
                        if (kFull || i < cnt) r0 = osc_tick<WF>(ph, hd, q, s_tab, s_tab);  // ragged last chunk: the state must not advance past N
                        if (kFull || i + 1 < cnt) r1 = osc_tick<WF>(ph, hd, q, s_tab, s_tab);
                        if constexpr (STORE == 2) {
                            if constexpr (kFull) {
                                store_pair_rows<2>(op, r0, r1);
                                op += 2 * V;
                            } else {
                                if (i < cnt) o[0] = r0;
                                if (i + 1 < cnt) o[V] = r1;
                                o += 2 * V;
                            }
                        } else if constexpr (STORE == 1) {
                            if (kFull || i < cnt) o[0] = r0;
                            if (kFull || i + 1 < cnt) o[V] = r1;
                            o += 2 * V;
                        }
                        const double s0 = l4 == (i & 3) ? 1.0 : 0.0, s1 = l4 == ((i + 1) & 3) ? 1.0 : 0.0;
                        DL[i >> 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(s0, r0 * gl, DL[i >> 2], 0, 0, 0);  // two[0] = input*sqrt(1.0-x)   C:506
                        DR[i >> 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(s0, r0 * gr, DR[i >> 2], 0, 0, 0);  // two[1] = input*sqrt(x)       C:507
                        DL[i >> 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(s1, r1 * gl, DL[i >> 2], 0, 0, 0);
                        DR[i >> 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(s1, r1 * gr, DR[i >> 2], 0, 0, 0);
                    }
                    if constexpr (kFull && STORE == 2) o += (size_t)kMixChunk * V;
                    const double2v pr2 = {
                        quad_sum(fold_dpp<kDppRowHalfMirror, 0xA>(fold_dpp<kDppRowMirror, 0xC>(DL[0], DL[1]), fold_dpp<kDppRowMirror, 0xC>(DL[2], DL[3]))),
                        quad_sum(fold_dpp<kDppRowHalfMirror, 0xA>(fold_dpp<kDppRowMirror, 0xC>(DR[0], DR[1]), fold_dpp<kDppRowMirror, 0xC>(DR[2], DR[3])))};
                    if (slot >= 0 && slot < cnt) *reinterpret_cast<double2v *>(my_part + (c0 + slot) * 2) = pr2;
                    return;
                }
                if constexpr (kFull && STORE == 2) {
#pragma unroll
                    for (int i = 0; i < kMixChunk; i += 2) {
                        const double r0 = osc_tick<WF>(ph, hd, q, s_tab, s_tab);
                        const double r1 = osc_tick<WF>(ph, hd, q, s_tab, s_tab);
                        store_pair_rows<2>(op, r0, r1);
                        op += 2 * V;
                        L[i] = r0 * gl;      // two[0] = input*sqrt(1.0-x)   C:506
                        R[i] = r0 * gr;      // two[1] = input*sqrt(x)       C:507
                        L[i + 1] = r1 * gl;
                        R[i + 1] = r1 * gr;
                    }
                    o += (size_t)kMixChunk * V;
                } else {
#pragma unroll
                    for (int i = 0; i < kMixChunk; i++) {
                        double r = 0.0;
                        if (kFull || i < cnt) {  // ragged last chunk: the state must not advance past N
                            r = osc_tick<WF>(ph, hd, q, s_tab, s_tab);
                            if constexpr (STORE != 0) {
                                *o = r;
                                o += V;
                            }
                        }
                        L[i] = r * gl;  // two[0] = input*sqrt(1.0-x)   C:506
                        R[i] = r * gr;  // two[1] = input*sqrt(x)       C:507
                    }
                }
                double2v pr;
                if constexpr (VAR == 0)
                    pr = (double2v){quad_sum(fold_chunk_swap<double>(L)), quad_sum(fold_chunk_swap<double>(R))};
                else
                    pr = (double2v){fold_chunk<double>(L, lane), fold_chunk<double>(R, lane)};
                if (slot >= 0 && slot < cnt) *reinterpret_cast<double2v *>(my_part + (c0 + slot) * 2) = pr;
            };
            if (cnt == kMixChunk) chunk(std::true_type{}); else chunk(std::false_type{});
        }
        // 4 waves x 4 rows of this window -> one partial per workgroup: 16 terms, added in the order wave 0 row 0..3, wave 1 ...
        __syncthreads();
        double *prow = partial + (size_t)blockIdx.x * N * 2 + n0 * 2;
        for (int i = threadIdx.x; i < span * 2; i += blockDim.x) {
            double t = s_part[i];
#pragma unroll
            for (int k = 1; k < kRows; k++) t += s_part[k * (kMixWin * 2) + i];
            prow[i] = t;
        }
        __syncthreads();
    }
    if (blockIdx.y + 1 == gridDim.y && part_wait(part_ctr, psync)) {
        phase_io[v] = ph;
        hold_io[v] = hd;
    }
}

typedef void (*osc_mix_fn)(size_t, size_t, const double *, const double *, const double *, double *, double *,
                           double *, const double *, double *, double, PartSync);
// store: 0 none, 1 plain, 2 pair rows (sc1); var / window A/B forms only for the bench waveform
template <int WF>
osc_mix_fn pick_mix(int store, int var) {
    if (WF == MXG_OSC_SINEBUF) {
        switch (var) {
            case 1: return store ? osc_mix_kernel<WF, 1, 0, 128> : osc_mix_kernel<WF, 0, 0, 128>;
            case 2: return store ? osc_mix_kernel<WF, 1, 1, 128> : osc_mix_kernel<WF, 0, 1, 128>;
            case 3: return store ? osc_mix_kernel<WF, 1, 1, 512> : osc_mix_kernel<WF, 0, 1, 512>;
            default: break;
        }
    }
    if (var == 4)  // the matrix-pipe form
        return store == 2 ? osc_mix_kernel<WF, 2, 2, 256> : (store == 1 ? osc_mix_kernel<WF, 1, 2, 256> : osc_mix_kernel<WF, 0, 2, 256>);
    return store == 2 ? osc_mix_kernel<WF, 2, 0, 256> : (store == 1 ? osc_mix_kernel<WF, 1, 0, 256> : osc_mix_kernel<WF, 0, 0, 256>);
}
osc_mix_fn pick_mix_wf(int wf, int store, int var) {
    switch (wf) {
        case 0: return pick_mix<0>(store, var);
        case 1: return pick_mix<1>(store, var);
        case 2: return pick_mix<2>(store, var);
        case 3: return pick_mix<3>(store, var);
        case 4: return pick_mix<4>(store, var);
        case 5: return pick_mix<5>(store, var);
        case 6: return pick_mix<6>(store, var);
        case 7: return pick_mix<7>(store, var);
        case 8: return pick_mix<8>(store, var);
        case 9: return pick_mix<9>(store, var);
        case 10: return pick_mix<10>(store, var);
        case 11: return pick_mix<11>(store, var);
    }
    return nullptr;
}

typedef void (*osc_fn)(size_t, size_t, const double *, const double *, const double *, double *,
                       double *, double *, double, PartSync, int, int);

// store: 0 plain 8 B, 1 nt 8 B, 2 pair rows (16 B) plain, 3 pair rows sc1, 4 pair rows nt      (one voice per lane)
//        0 plain 16 B, 1 nt 16 B, 2 sc1 16 B                                                  (two voices per lane)
template <int WF>
osc_fn pick(bool fps, int vpl, int store) {
    if (fps) return osc_kernel<WF, true, 1, 0, false>;
    if (vpl == 2) return store == 2 ? osc_kernel<WF, false, 2, 2, false> : (store == 1 ? osc_kernel<WF, false, 2, 1, false> : osc_kernel<WF, false, 2, 0, false>);
    switch (store) {
        case 1: return osc_kernel<WF, false, 1, 1, false>;
        case 2: return osc_kernel<WF, false, 1, 0, true>;
        case 3: return osc_kernel<WF, false, 1, 2, true>;
        case 4: return osc_kernel<WF, false, 1, 1, true>;
        default: return osc_kernel<WF, false, 1, 0, false>;
    }
}

osc_fn pick_wf(int wf, bool fps, int vpl, int store) {
    switch (wf) {
        case 0: return pick<0>(fps, vpl, store);
        case 1: return pick<1>(fps, vpl, store);
        case 2: return pick<2>(fps, vpl, store);
        case 3: return pick<3>(fps, vpl, store);
        case 4: return pick<4>(fps, vpl, store);
        case 5: return pick<5>(fps, vpl, store);
        case 6: return pick<6>(fps, vpl, store);
        case 7: return pick<7>(fps, vpl, store);
        case 8: return pick<8>(fps, vpl, store);
        case 9: return pick<9>(fps, vpl, store);
        case 10: return pick<10>(fps, vpl, store);
        case 11: return pick<11>(fps, vpl, store);
    }
    return nullptr;
}

}  // namespace

}  // namespace mxg

extern "C" int mxg_osc_render(int waveform, size_t V, size_t N, const double *d_freq, int fps,
                              const double *d_p1, const double *d_p2, double *d_phase,
                              double *d_outhold, double *d_out, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(waveform >= 0 && waveform <= 11, "unknown waveform");
    MXG_REQUIRE(d_freq && d_phase && d_outhold && d_out, "null device pointer");
    MXG_REQUIRE(waveform != MXG_OSC_PULSE || d_p1, "pulse needs d_p1 (duty)");
    MXG_REQUIRE(waveform != MXG_OSC_PHASORBETWEEN || (d_p1 && d_p2),
                "phasorBetween needs d_p1/d_p2 (start/end phase)");
    MXG_REQUIRE(fps >= 0 && fps <= 2 && (fps != 2 || d_p1), "fps is 0, 1 (d_freq [N][V]) or 2 (d_freq and d_p1 [N][V])");
    if (V == 0 || N == 0) return MXG_OK;
    // ---- the store stream --------------------------------------------------------------------------------------------
    // Knobs osc_vpl, osc_store, osc_xcd (0 = automatic each) name it; left alone, it goes by
    // waveform class and bank size, from the rotated-destination sweep of tools/sweep_osc_store.py (profiles/r03_osc_store.md;
    // MI355X, 512-sample blocks, fraction of the 8 TB/s peak on 8 B per sample):
    //   pair rows = ONE voice per lane, two samples of a lane pair exchanged into one write-through (sc1) 16-byte store per lane;
    //   2v        = TWO voices per lane, write-through 16-byte stores (half the wavefronts: only where one lane can carry two voices'
    //               arithmetic without exposing its latency).
    //   table forms (sinebuf, sawn):  < 49 152 voices plain 8-byte stores (the block lives in the caches; 32 768: 31 us against 37);
    //       < 98 304 pair rows (65 536: 51 -> 40.5 us, 0.66 -> 0.83); from there 2v (98 304: 79 -> 67; 131 072: 101 -> 88 us, 0.67 -> 0.77);
    //   ramps (phasor, saw, triangle, square, pulse, impulse, phasorBetween): 2v already from 65 536 voices (saw: 49 -> 40 us);
    //   VALU-heavy forms (sinewave, coswave, sinebuf4): pair rows at every size (65 536 sinewave: 69 -> 65, sinebuf4 60 -> 57);
    //   from 262 144 voices XCD-contiguous workgroup numbering (262 144 sinebuf: 215 -> 192 us; 1 048 576: 818 -> 755 us, 0.66 -> 0.71).
    const bool pairs_ok = !fps && !(V & 1) && !(((uintptr_t)d_out) & 15);
    const size_t out_bytes = V * N * sizeof(double);
    int vpl = tune_get("osc_vpl"), store = tune_get("osc_store") - 1, xcd = tune_get("osc_xcd") - 1;  // (knob value 0 = automatic)
    const bool automatic = vpl == 0 && store < 0;
    if (automatic) {
        vpl = 1;
        store = 0;
        const bool heavy = waveform == MXG_OSC_SINEWAVE || waveform == MXG_OSC_COSWAVE || waveform == MXG_OSC_SINEBUF4;
        const bool table = waveform == MXG_OSC_SINEBUF || waveform == MXG_OSC_SAWN;
        if (pairs_ok) {
            if (heavy) {
                if (out_bytes >= ((size_t)32 << 20)) store = 3;
            } else if (out_bytes >= ((size_t)192 << 20)) {
                if (V >= (table ? 98304u : 65536u)) {
                    vpl = 2;
                    store = 2;
                } else {
                    store = 3;
                }
            }
        }
        if (xcd < 0) xcd = V >= 262144 ? 1 : 0;
    } else {
        if (vpl == 0) vpl = 1;
        if (store < 0) {  // (the round-2 rule for 8-byte stores, knob osc_nt: non-temporal by block size)
            const int nt_knob = tune_get("osc_nt");
            store = (nt_knob == 1 || (nt_knob == 2 && out_bytes > ((size_t)300 << 20) && out_bytes <= ((size_t)1200 << 20))) ? 1 : 0;
        }
        if (xcd < 0) xcd = 0;
    }
    if (!pairs_ok) vpl = 1;
    int block = tune_get("osc_block");
    if (vpl == 2 && store > 2) store = 0;
    if (vpl == 1 && store >= 2 && !pairs_ok) store = store == 4 ? 1 : 0;  // pair rows need whole pairs
    osc_fn fn = pick_wf(waveform, fps != 0, vpl, store);
    size_t lanes = (V + vpl - 1) / vpl;
    // time parts.  (a) Where the output dominates the recurrence (sinewave, coswave, sinebuf4) and the bank is too small to give every
    // SIMD two wavefronts by itself: two parts.  (b) SMALL banks (fewer wavefronts than the 1024 SIMDs) of the waveforms whose phase
    // skip is much cheaper than their tick (the table oscillators: sinebuf, sinebuf4, sawn, sinewave, coswave): a block is one chain of
    // N dependent steps per wavefront, 27-30 us for 512 samples however few voices there are; cut into up to eight parts it is 14.6 us
    // at 1024 voices, 17.0 at 4096, 20.1 at 16 384 (sinebuf; profiles/r03_small_osc_banks.md).  Same bits (the skip is the same
    // additions); the other waveforms' tick IS their recurrence, parts would only repeat it.
    int split = tune_get("osc_split");
    if (split == 0) {
        split = 1;
        const size_t waves = (lanes + 63) / 64;
        const bool heavy = waveform == MXG_OSC_SINEWAVE || waveform == MXG_OSC_COSWAVE || waveform == MXG_OSC_SINEBUF4;
        const bool table = heavy || waveform == MXG_OSC_SINEBUF || waveform == MXG_OSC_SAWN;
        if (!fps && heavy) split = waves >= 2048 ? 1 : 2;  // (sinewave at 65 536 voices: 60-62 us in two parts; one part 65-69, four 64)
        if (!fps && table && waves < 1024) {
            int want = 1;
            while (want < 8 && (size_t)(2 * want) * waves <= 1024 && (size_t)(2 * want) * 64 <= N) want *= 2;  // (a part renders >= 64 samples)
            if (want > split) split = want;
        }
    }
    if (fps) split = 1;
    // every part must render at least one sample: the last part's ticks leave the member `output` of the final sample
    while (split > 1 && (size_t)(split - 1) * ((N + split - 1) / split) >= N) split--;
    dim3 grid((unsigned)((lanes + block - 1) / block), (unsigned)split), blk((unsigned)block);
    PartSync psync;
    if (split > 1)
        if (int s = part_sync_get(resolve_stream(stream), (size_t)grid.x * ((block + 63) / 64), split, &psync)) return s;
    KernelTimer kt("osc_kernel", resolve_stream(stream));
    hipLaunchKernelGGL(fn, grid, blk, 0, resolve_stream(stream), V, N, d_freq, d_p1, d_p2, d_phase,
                       d_outhold, d_out, (double)settings().sampleRate, psync, xcd, fps == 2 ? 1 : 0);
    return check_hip(hipGetLastError(), "osc_kernel launch");
}

extern "C" int mxg_osc_render_mix(int waveform, size_t V, size_t N, const double *d_freq, const double *d_p1,
                                  const double *d_p2, double *d_phase, double *d_outhold, double *d_out,
                                  const double *d_pan, double *d_mix, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(waveform >= 0 && waveform <= 11, "unknown waveform");
    MXG_REQUIRE(d_freq && d_phase && d_outhold && d_pan && d_mix, "null device pointer");
    MXG_REQUIRE(waveform != MXG_OSC_PULSE || d_p1, "pulse needs d_p1 (duty)");
    MXG_REQUIRE(waveform != MXG_OSC_PHASORBETWEEN || (d_p1 && d_p2), "phasorBetween needs d_p1/d_p2");
    if (N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const int block = 256;
    const size_t nblocks = (V + block - 1) / block;
    const size_t need = N * nblocks * 2 + 2;
    double *partial = nullptr;  // per-stream scratch: [nblocks][N][2] per-workgroup sums
    if (int s = scratch_get(SCR_OSC_MIX, st, sizeof(double) * need, (void **)&partial)) return s;
    if (V) {
        // per-voice block: pair rows of write-through 16-byte stores where whole pairs exist (knob osc_mix_store: 0 automatic,
        // 1 plain 8-byte stores, 2 pair rows); time parts (knob osc_mix_split: 0 automatic = one)
        const int var = tune_get("osc_mix_var");
        int store = 0;
        if (d_out) {
            const bool pairs_ok = !(V & 1) && !(((uintptr_t)d_out) & 15) && V >= 2;
            const int knob = tune_get("osc_mix_store");
            store = (pairs_ok && var == 0 && (knob == 2 || (knob == 0 && V * N * sizeof(double) >= ((size_t)32 << 20)))) ? 2 : 1;
        }
        int split = tune_get("osc_mix_split");
        if (split == 0) {
            split = 1;  // (measured, MI355X 65 536 x 512 rotated: 1 part 51.1 us, 2 parts 53.6, 3 parts 54.8 -- the lane folds' permlane swaps do
            // not overlap across wavefronts).  SMALL banks of the table oscillators are another matter, as in mxg_osc_render: fewer
            // wavefronts than SIMDs, one chain of N dependent steps each -- up to eight parts
            const bool table = waveform == MXG_OSC_SINEWAVE || waveform == MXG_OSC_COSWAVE || waveform == MXG_OSC_SINEBUF4 ||
                               waveform == MXG_OSC_SINEBUF || waveform == MXG_OSC_SAWN;
            const size_t waves = nblocks * 4;
            if (table && waves < 1024)
                while (split < 8 && (size_t)(2 * split) * waves <= 1024 && (size_t)(2 * split) * 64 <= N) split *= 2;  // (a part renders >= 64 samples)
        }
        if (var != 0) split = 1;
        while (split > 1 && (size_t)(split - 1) * (((N + split - 1) / split + kMixChunk - 1) / kMixChunk * kMixChunk) >= N) split--;
        PartSync psync;
        if (split > 1)
            if (int s2 = part_sync_get(st, nblocks * 4, split, &psync)) return s2;
        osc_mix_fn fn = pick_mix_wf(waveform, store, var);
        if (var == 3 && waveform == MXG_OSC_SINEBUF)
            MXG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 0));
        KernelTimer kt("osc_mix_kernel", st);
        // a bank of one workgroup (<= 256 voices): its "partial" row IS the mix -- written in place, no second kernel (4-5 us of an
        // 18 us call at 64 voices)
        hipLaunchKernelGGL(fn, dim3((unsigned)nblocks, (unsigned)split), dim3(block), 0, st, V, N, d_freq, d_p1, d_p2, d_phase,
                           d_outhold, d_out, d_pan, nblocks == 1 ? d_mix : partial, (double)settings().sampleRate, psync);
        if (nblocks == 1) return check_hip(hipGetLastError(), "osc_mix_kernel launch");
    }
    KernelTimer kt2("mix_partials_kernel", st);
    hipLaunchKernelGGL(mix_partials_kernel, dim3((unsigned)((N * 2 + 63) / 64)), dim3(64 * kPartWaves), 0, st, nblocks, N * 2,
                       partial, d_mix);
    return check_hip(hipGetLastError(), "osc_mix_kernel launch");
}

// ---- maxiOsc::noise (C:214-220) -----------------------------------------------------------------
//     float r = rand()/(float)RAND_MAX;  output = r*2-1;
// rand() is one process-wide serial stream: the caller supplies the draws (see maxigpu.h), the
// kernel does the reference's float arithmetic.  (float)RAND_MAX = 2^31 exactly; the int -> float
// conversion rounds to nearest even as cvtsi2ss does; r*2-1 is evaluated in float (int operands
// convert to float), then widened to the double member.  Pure streaming: 4 B in, 8 B out.
namespace mxg {
namespace {
__global__ void osc_noise_kernel(size_t count, size_t V, size_t N, const int32_t *__restrict__ rnd,
                                 double *__restrict__ outhold, double *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float r = (float)rnd[i] / 2147483648.0f;
        const double o = (double)(r * 2.0f - 1.0f);
        out[i] = o;
        if (outhold && i >= count - V) outhold[i - (count - V)] = o;
    }
}
}  // namespace
}  // namespace mxg

extern "C" int mxg_osc_noise(size_t V, size_t N, const int32_t *d_rand, double *d_outhold, double *d_out,
                             void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_rand && d_out, "null device pointer");
    if (N == 0 || V == 0) return MXG_OK;
    const size_t count = V * N;
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(osc_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, resolve_stream(stream), count, V,
                       N, d_rand, d_outhold, d_out);
    return check_hip(hipGetLastError(), "osc_noise_kernel launch");
}
