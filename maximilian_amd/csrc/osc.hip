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
#include "mxg_pace.h"

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
static_assert(tab_copy<MXG_OSC_SINEBUF, 1>() >= MAXI_SINE_TAB_LEN, "sinebuf's table copy");
static_assert(tab_copy<MXG_OSC_SAWN, 1>() >= MAXI_TRANS_TAB_LEN, "sawn's table copy");
static_assert(kSb4Copy >= MAXI_SINE_TAB_LEN && (kSb4Copy - 2) * 8 > 255 * 8, "sinebuf4's table copies: disjoint, and out of a ds_read2_b64's reach");
template <int WF, int FL = 0>
constexpr int tab_len() {
    return (WF == MXG_OSC_SINEBUF4 && MXG_SB4_PAIRS) ? kSineOddOff + MAXI_SINE_TAB_LEN + 1
           : WF == MXG_OSC_SINEBUF4                  ? 4 * kSb4Copy
           : uses_sine<WF>()                         ? tab_copy<WF, FL>() + MAXI_SINE_TAB_LEN
                           : (WF == MXG_OSC_SAWN ? tab_copy<WF, FL>() + MAXI_TRANS_TAB_LEN
                                                 : ((WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) ? MXG_SINTAB_WIDE_LEN : 1));
}
template <int WF, int FL = 0>
__device__ __forceinline__ void load_tab(double *s_tab) {  // (the caller synchronises)
    if constexpr (uses_sine<WF>()) {
        for (int i = threadIdx.x; i < MAXI_SINE_TAB_LEN; i += blockDim.x) {
            s_tab[i] = MAXI_SINE_TAB_D[i];
            if constexpr (WF == MXG_OSC_SINEBUF4 && MXG_SB4_PAIRS) s_tab[kSineOddOff + i] = MAXI_SINE_TAB_D[i];  // the parity copy (osc_tick)
            if constexpr (tab_copy<WF, FL>() != 0) s_tab[tab_copy<WF, FL>() + i] = MAXI_SINE_TAB_D[i];  // sinebuf: the copy (osc_tick)
            if constexpr (WF == MXG_OSC_SINEBUF4 && !MXG_SB4_PAIRS) {  // the four copies (osc_tick)
                s_tab[kSb4Copy + i] = MAXI_SINE_TAB_D[i];
                s_tab[2 * kSb4Copy + i] = MAXI_SINE_TAB_D[i];
                s_tab[3 * kSb4Copy + i] = MAXI_SINE_TAB_D[i];
            }
        }
    } else if constexpr (WF == MXG_OSC_SAWN) {
        for (int i = threadIdx.x; i < MAXI_TRANS_TAB_LEN; i += blockDim.x) {
            s_tab[i] = MAXI_TRANS_TAB_D[i];
            if constexpr (tab_copy<WF, FL>() != 0) s_tab[tab_copy<WF, FL>() + i] = MAXI_TRANS_TAB_D[i];
        }
    } else if constexpr (WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) {
        for (int i = threadIdx.x; i < MXG_SINTAB_WIDE_LEN; i += blockDim.x) s_tab[i] = MXG_SINTAB_D[i & (MXG_SINTAB_LEN - 1)];  // entries 0 .. 1024
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
                           double *__restrict__ out, double sr, PartSync psync, int xcd, int p1ps, int passes, size_t v_begin,
                           size_t v_end, size_t P, unsigned *__restrict__ pace_ctl, unsigned pace_arg, unsigned *__restrict__ trial) {
    // P: the row pitch of `out` in doubles (>= V; mxg_osc_render_pitch -- a bank whose natural pitch V * 8 is a multiple of 2 MB puts
    // the same column of every row on the same HBM channel: a caller that pads its rows by a few hundred bytes removes that)
    // [v_begin, v_end): the voices of the bank this launch renders (V stays the bank's size = the row pitch of `out`): a large bank
    // is rendered as a few launches of the grid shapes that stream best (osc_plan below)
    // (p1ps, FPS only: p1 is [N][V] too -- a pulse width / start phase per sample, for the per-sample engine's derived arguments)
    // passes (round 4): the grid covers 1 / passes of the bank and every wavefront renders `passes` voice groups one after the other
    // (group stride = the grid's width): banks beyond the machine's 1024 wavefronts keep the access pattern of the 65 536-voice bank --
    // every resident wavefront walking down the SAME rows at the same time, one contiguous row slice per pass -- instead of 2, 3, ...
    // wavefronts per SIMD drifting apart over rows that are megabytes long (profiles/r04_osc_grid.md)
    // All LDS in ONE array (a second __shared__ object perturbs hipcc's waitcnt placement).
    // Which form of the tick and of the loop a waveform gets is decided by MEASUREMENT (profiles/r04_heavy_osc.md): the VALU-bound
    // ones (sinewave, coswave, sinebuf4) and sawn take the lean forms -- 32-bit trip count, two table copies, three-instruction wrap:
    // sawn 52 -> 46 us, sinebuf4 57 -> 52, sinewave 61 -> 58 -- while sinebuf and the table-free waveforms, whose time is the store
    // stream's, keep the loop they were tuned with: every one of these changes made sinebuf SLOWER (41 -> 46 us with all three).
#ifndef MXG_K1_LEAN_MASK
#define MXG_K1_LEAN_MASK ((1 << MXG_OSC_SINEWAVE) | (1 << MXG_OSC_COSWAVE) | (1 << MXG_OSC_SINEBUF4) | (1 << MXG_OSC_SAWN))  // A/B: per-waveform bit mask
#endif
    constexpr bool kLean = ((MXG_K1_LEAN_MASK >> WF) & 1) != 0;
    constexpr int kFL = kLean ? kTickLean : 0;
    __shared__ __attribute__((aligned(16))) double s_tab[tab_len<WF, kFL>()];
    // (round 6: the first pass's per-voice loads are REQUESTED before the table is staged -- the barrier behind the staging would
    // otherwise put their round trip behind the table's; a lane past the range requests the range's last voices, which it never uses)
    constexpr bool kPre = tab_len<WF, kFL>() > 1;  // (a waveform without a table has no barrier in front of its loads: measured 0.3 us slower with this)
    double ph0[VPL] = {}, hd0[VPL] = {}, f0[VPL] = {};
    if constexpr (kPre) {
        const size_t w0 = v_begin + ((size_t)xcd_block(blockIdx.x, gridDim.x, xcd) * blockDim.x + threadIdx.x) * VPL;
        const size_t wc = (w0 + VPL <= v_end) ? w0 : (v_end >= v_begin + VPL ? v_end - VPL : v_begin);
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            const size_t w = wc + j < v_end ? wc + j : v_end - 1;
            ph0[j] = phase_io[w];
            hd0[j] = hold_io[w];
            f0[j] = FPS ? 0.0 : freq[w];
        }
    }
    if constexpr (tab_len<WF, kFL>() > 1) {
        load_tab<WF, kFL>(s_tab);
        __syncthreads();
    }
    for (int pass = 0; pass < passes; pass++) {
    const size_t v0 = v_begin + (((size_t)pass * gridDim.x + xcd_block(blockIdx.x, gridDim.x, xcd)) * blockDim.x + threadIdx.x) * VPL;
    if (v0 >= v_end) return;

    double ph[VPL], hd[VPL];
    OscPre q[VPL];
#pragma unroll
    for (int j = 0; j < VPL; j++) {
        const bool pre = kPre && pass == 0 && v0 + VPL <= v_end;  // (the values requested above are this lane's own)
        ph[j] = pre ? ph0[j] : phase_io[v0 + j];
        hd[j] = pre ? hd0[j] : hold_io[v0 + j];
        double a = p1 ? p1[v0 + j] : 0.0, b = p2 ? p2[v0 + j] : 0.0;
        if constexpr (!FPS) {
            q[j] = osc_pre<WF>(pre ? f0[j] : freq[v0 + j], sr, a, b);
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
    double *o = out + nA * P + v0;
    const double *fp = freq + v0;
    const double *pp = (FPS && p1ps) ? p1 + v0 : nullptr;
#ifndef MXG_OSC_UNROLL
#define MXG_OSC_UNROLL 4
#endif

    // sinewave / coswave with 0 <= inc < 1 and the phase in [0, 2) on every lane of the wavefront (any audio frequency from a
    // fresh or carried bank): the argument of sin / cos stays in [0, 4 pi), so the table routine needs neither its range test nor
    // a sign nor an index wrap (mxg_sincos.h, TRUST: the LDS table holds the period twice), and the phase wrap is v_fract_f64
    bool trust = false;
    if constexpr ((WF == MXG_OSC_SINEWAVE || WF == MXG_OSC_COSWAVE) && !FPS) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < VPL; j++)  // (sign bits clear: a -0.0 phase or increment takes the general form)
            ok = ok && q[j].inc >= 0.0 && q[j].inc < 1.0 && ph[j] >= 0.0 && ph[j] < 2.0 && (__double2hiint(ph[j]) | __double2hiint(q[j].inc)) >= 0;
        trust = __all(ok);
        if (trust) {
#pragma unroll
            for (int j = 0; j < VPL; j++) q[j].sk = sintab_k();  // the two vector-register coefficients, loaded once (mxg_sincos.h)
        }
    }
    // PACE (mxg_pace.h): where the launch asks for it (one part, one pass, a store-bound waveform at the size where every SIMD holds one
    // wavefront) eight samples start every P ticks of the 100 MHz counter
    // (trial != null: this launch holds BOTH forms, the free-running pair rows and the paced 8-byte stream; the trial's words say which runs)
    PaceTrial tr;
    tr.start(trial, pace_arg, (unsigned)N);
    Pace pc;
    pc.start(trial ? nullptr : pace_ctl, trial ? tr.period : pace_arg);
    auto run = [&](auto trust_tag) {
        constexpr bool kTrust = decltype(trust_tag)::value;
        size_t n = nA;
        if constexpr (PX && VPL == 1 && !FPS) {
            double *op = out + (nA + (threadIdx.x & 1)) * P + (v0 & ~(size_t)1);
            if constexpr (!kLean) {
                if (pc.P) {  // the paced schedule (mxg_pace.h): eight samples per slot, as non-temporal 8-byte stores (what the paced
                             // stream wants: profiles/r06_pace.md; this kernel's own pair rows are the free-running form)
                    double *o8 = out + nA * P + v0;
                    for (; n + 8 <= nB; n += 8) {
                        pc.wait(true);
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            store1<1>(o8, osc_tick<WF, kTrust, kFL>(ph[0], hd[0], q[0], s_tab, s_tab));
                            o8 += P;
                        }
                    }
                    op += (n - nA) * P;
                }
#pragma unroll 2
                for (; n + 2 <= nB; n += 2) {
                    const double r0 = osc_tick<WF, kTrust, kFL>(ph[0], hd[0], q[0], s_tab, s_tab);
                    const double r1 = osc_tick<WF, kTrust, kFL>(ph[0], hd[0], q[0], s_tab, s_tab);
                    store_pair_rows<ST, false>(op, r0, r1);
                    op += 2 * P;
                }
            } else {
                // (a 32-bit trip count: gfx950 has no scalar 64-bit less-than, and hipcc then tests a size_t bound with two VALU
                // instructions per iteration -- in a loop whose VALU-bound forms have ~35 per sample)
                const unsigned pairs = (unsigned)((nB - nA) >> 1);
                for (unsigned k = 0; k < pairs; k++) {
                    const double r0 = osc_tick<WF, kTrust, kFL>(ph[0], hd[0], q[0], s_tab, s_tab);
                    const double r1 = osc_tick<WF, kTrust, kFL>(ph[0], hd[0], q[0], s_tab, s_tab);
                    store_pair_rows<ST>(op, r0, r1);
                    op += 2 * P;
                }
                n = nA + 2 * (size_t)pairs;
            }
            o = out + n * P + v0;
        }
        if constexpr (VPL == 1 && !FPS && !PX && !kLean) {
            if (pc.P) {  // the paced schedule, 8-byte store streams
                for (; n + 8 <= nB; n += 8) {
                    pc.wait(true);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        store1<ST>(o, osc_tick<WF, kTrust, kFL>(ph[0], hd[0], q[0], s_tab, s_tab));
                        o += P;
                    }
                }
            }
        }
#pragma unroll MXG_OSC_UNROLL
        for (; n < nB; n++) {
            double r[VPL];
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                if constexpr (FPS) q[j] = osc_pre<WF>(fp[j], sr, pp ? pp[j] : q[j].p1, q[j].p2);
                r[j] = osc_tick<WF, kTrust, kFL>(ph[j], hd[j], q[j], s_tab, s_tab);
            }
            if constexpr (VPL == 2)
                store2<ST>(o, r[0], r[1]);
            else
                store1<ST>(o, r[0]);
            o += P;
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
    if (threadIdx.x == 0) {
        pc.finish(trial ? nullptr : pace_ctl, pace_arg, blockIdx.x, gridDim.x, WF == MXG_OSC_SINEBUF);
        tr.finish(trial, pace_arg, blockIdx.x, gridDim.x);
    }
    }  // passes (time parts are launched with one pass only: the part counters are per wavefront of the grid)
}

// ---- K1m: K1 + fused maxiMix::stereo partial sums ---------------------------------------------------
// Same per-voice recurrence and (optional) per-voice store as K1; in addition every workgroup forms the panned sum of its
// 256 voices (in*sqrt(1-x), in*sqrt(x), C:503-509; the sum over voices is the user's `mix +=` loop, 15.polysynth/main.cpp:67)
// per sample, so the mixdown never re-reads the 268 MB block from HBM.
//
// Round 4: the sum over the 64 voices of a wavefront goes through an LDS TRANSPOSE instead of a lane butterfly.  Lanes are
// voices while the oscillator ticks; every lane drops its RAW sample into a [16 samples][64 voices] tile of the wavefront
// (one ds_write_b64 per sample: 16 contiguous lanes cover 128 bytes, conflict-free), and once 16 samples are in, the
// wavefront re-reads the tile the other way round: lane (s = lane & 15, q = lane >> 4) fetches sample s of the 16 voices
// 16 q .. 16 q + 15 (8 x ds_read_b128), multiplies them with THOSE voices' gains -- which it keeps in registers, 16 + 16
// doubles -- and adds the 16 products per channel in a fixed tree.  Two exchange steps fold the four voice quarters
// (v_permlane32_swap, then a ds_bpermute by 16 lanes).  Per 16 samples and lane that is 62 fp64 operations, 16 LDS writes,
// 8 LDS reads and ~10 exchange instructions: ~6 instructions per voice-sample on top of the oscillator's tick, where the
// register butterfly it replaces (rounds 1-3: DPP mirrors, permlane swaps, or the matrix pipe for the cross-row half) cost
// ~40 and left the kernel VALU-bound at 48-50 us against K1's 41 us.
// Tile layout: voice quarter q, sample s, voice j of the quarter at byte q * 2304 + s * 144 + j * 8.  The row stride of 144 B
// moves consecutive samples by nine 16-byte bank groups, so the 16 lanes the LDS serves per cycle of a ds_read_b128 (each lane
// group of MI355X_MICROARCH.md's table holds every s exactly once) fall on 16 different groups; 2304 B = 9 x 256 keeps the
// quarters bank-neutral.
// The workgroup's four wavefront sums per sample are added left to right once per window of WIN samples, and the
// per-workgroup rows [workgroup][N][2] are the kernel's output: mix_partials_kernel (or the mix queue, which does it once per
// batch on its own stream: comm.hip) adds them in workgroup order.  A fixed order everywhere => deterministic, but not the
// reference's sequential voice order: tolerance on the mix, per-voice signals bit-exact.
//
// STORE: 0 = mix only (no per-voice block), 1 = plain 8-byte stores, 2 = pair rows of write-through 16-byte stores (as K1, V
// even and `out` 16-byte aligned).  Time parts (gridDim.y) as in K1: part p advances the phase over the samples before it with
// osc_skip (the same additions: the same bits), renders its stretch and mixes it into its own rows of the partial buffer; the
// last part stores the state (part_signal / part_wait).
// (tile geometry kTileRow / kTileQuarter / kTileWave: mxg_lanefold.h -- shared with K2f's mixdown form, voice.hip)

template <int WF, int STORE, int WIN>
__global__ __launch_bounds__(256) void osc_mix_kernel(size_t V, size_t N, const double *__restrict__ freq,
                                                      const double *__restrict__ p1, const double *__restrict__ p2,
                                                      double *__restrict__ phase_io, double *__restrict__ hold_io,
                                                      double *__restrict__ out, const double *__restrict__ pan,
                                                      double *__restrict__ partial, double sr, PartSync psync, int passes) {
    constexpr int kTab = tab_len<WF, kTickLean>();
    constexpr int kTabPad = (kTab + 1) & ~1;  // the tiles are read with 16-byte loads
    __shared__ __attribute__((aligned(16))) double s_all[kTabPad + 4 * kTileWave + 4 * WIN * 2 + 256];
    double *s_tab = s_all;
    load_tab<WF, kTickLean>(s_tab);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *tile = s_all + kTabPad + wave * kTileWave;
    double *s_part = s_all + kTabPad + 4 * kTileWave;  // [4 waves][WIN][2]
    double *my_part = s_part + wave * (WIN * 2);
    double *s_dump = s_part + 4 * WIN * 2;  // [256]: where the lanes that hold no sum drop their value
    // The transposed read needs all 64 lanes alive, and a per-sample `if (live)` costs an exec-mask region per sample:
    // the surplus lanes of the bank's last wavefront shadow a live voice instead (same loads, same arithmetic, same stores
    // of the same values to the same addresses) and enter the mix with zero gains -- voice V-1, or with pair rows the last
    // PAIR of voices, parity kept, so that they exchange among themselves.
    // passes: as K1 -- the workgroup renders `passes` groups of 256 voices one after the other (group stride = the grid's width)
    for (int pass = 0; pass < passes; pass++) {
    const size_t wg = (size_t)pass * gridDim.x + blockIdx.x;
    if (wg * blockDim.x >= V) break;
    if (pass) __syncthreads();  // (the last window's combine has read s_part; the tile is handed the next gains)
    const size_t vraw = wg * blockDim.x + threadIdx.x;
    const bool live = vraw < V;
    const size_t v = live ? vraw : (STORE == 2 ? V - 2 + (vraw & 1) : V - 1);
    double ph = phase_io[v], hd = hold_io[v];
    double x = pan[v];
    if (x > 1) x = 1;  // C:504
    if (x < 0) x = 0;  // C:505
    // the gains of the 16 voices this lane sums, handed round through the (still unused) tile
    const int ts = lane & 15, tq = lane >> 4;
    double gl[16], gr[16];
    tile[lane] = live ? sqrt(1.0 - x) : 0.0;  // two[0] = input*sqrt(1.0-x)   C:506
    tile[64 + lane] = live ? sqrt(x) : 0.0;   // two[1] = input*sqrt(x)       C:507
    __syncthreads();  // (the table and the gains)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        gl[j] = tile[16 * tq + j];
        gr[j] = tile[64 + 16 * tq + j];
    }
    OscPre q = osc_pre<WF>(freq[v], sr, p1 ? p1[v] : 0.0, p2 ? p2[v] : 0.0);
    // Consume the prologue loads HERE: otherwise hipcc's waitcnt pass keeps them "pending" at the loop back-edge and
    // drains the asynchronous output stores with s_waitcnt vmcnt(0) every chunk.
    asm volatile("" : "+v"(ph), "+v"(hd));
    asm volatile("" : "+v"(q.inc), "+v"(q.k), "+v"(q.p1), "+v"(q.p2));
#pragma unroll
    for (int j = 0; j < 16; j++) asm volatile("" : "+v"(gl[j]), "+v"(gr[j]));
    __syncthreads();  // every lane has its gains: the tile is free
    // time parts: this part renders [nA, nB); part lengths are whole mix chunks, so a chunk never straddles two parts
    int *const part_ctr = gridDim.y > 1 ? part_counter(psync) : nullptr;
    if (blockIdx.y + 1 != gridDim.y) part_signal(part_ctr);
    const size_t plen = ((N + gridDim.y - 1) / gridDim.y + kMixChunk - 1) / kMixChunk * kMixChunk;
    const size_t nA = blockIdx.y * plen < N ? blockIdx.y * plen : N;
    const size_t nB = nA + plen < N ? nA + plen : N;
#pragma unroll 4
    for (size_t n = 0; n < nA; n++) osc_skip<WF>(ph, hd, q, s_tab, s_tab);
    double *tw = tile + tq * kTileQuarter + ts;  // this VOICE's column: + i * kTileRow for sample i of the chunk
    const double2v *tr = reinterpret_cast<const double2v *>(tile + tq * kTileQuarter + ts * kTileRow);  // sample ts, quarter tq
    double *o = out + nA * V + v;
    double *op = out + (nA + (threadIdx.x & 1)) * V + (v & ~(size_t)1);  // pair rows: this lane's 16 bytes of row n + (lane & 1)
    for (size_t n0 = nA; n0 < nB; n0 += WIN) {
        const int span = (int)((nB - n0) < (size_t)WIN ? (nB - n0) : (size_t)WIN);
        // the voice sum of one chunk: the tile read back transposed, 16 products per channel added in a fixed tree, the four voice
        // quarters folded (lanes 0-31 <- left sums of quarters (0, 2) / (1, 3), lanes 32-63 the right sums; then the two rows of 16)
        auto tile_load = [&](double2v (&xv)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) xv[k] = tr[k];
        };
        // (pure VALU: products, the tree, v_permlane32_swap and v_permlane16_swap -- nothing that queues behind LDS reads in flight)
        auto tile_products = [&](const double2v (&xv)[8], double &sl, double &sr2) {
            double pl[8], pr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                pl[k] = xv[k].x * gl[2 * k] + xv[k].y * gl[2 * k + 1];
                pr[k] = xv[k].x * gr[2 * k] + xv[k].y * gr[2 * k + 1];
            }
            sl = ((pl[0] + pl[1]) + (pl[2] + pl[3])) + ((pl[4] + pl[5]) + (pl[6] + pl[7]));
            sr2 = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
        };
        auto tile_fold = [&](double sl, double sr2) {
            const double t = fold32(sl, sr2);
            return fold16(t, t);  // lane l <- t[l] + t[l ^ 16]
        };
        auto tile_value = [&](const double2v (&xv)[8]) {
            double sl, sr2;
            tile_products(xv, sl, sr2);
            return tile_fold(sl, sr2);
        };
        // lanes 0-15 hold the left sums of samples c0 + lane, lanes 32-47 the right sums; the other lanes (and samples beyond cnt) write
        // into a scratch row instead -- a select on the address, not a branch around the store
        auto tile_put = [&](double t, int c0, int cnt) {
            double *dst = ((lane & 16) == 0 && ts < cnt) ? my_part + (c0 + ts) * 2 + (lane >> 5) : s_dump + threadIdx.x;
            *dst = t;
        };
        auto tile_sum = [&](const double2v (&xv)[8], int c0, int cnt) { tile_put(tile_value(xv), c0, cnt); };
        int c_first = 0;
        // Table oscillators with pair rows (or no per-voice block): the chunks of a window as a software pipeline.  While the
        // (bank-conflicted) table reads of chunk c + 1 drain, the wavefront adds up chunk c -- a lone wavefront otherwise sits out
        // ~60 clk per sample on s_waitcnt (SQ_WAIT_ANY, profiles/r04_k1m_sq.md).  LDS operations complete in order, so the tile
        // reads of chunk c are issued BEFORE the table reads of chunk c + 1 and the sums need only them.
        if constexpr (osc_has_pipe<WF>() && STORE != 1) {
            const int full = span / kMixChunk;
            if (full > 0) {
                OscPipe<kMixChunk> P;
                osc_pipe_phase<WF, kMixChunk>(ph, q, P);
                __builtin_amdgcn_sched_barrier(0);
                osc_pipe_fetch<WF, kMixChunk>(P, s_tab);
                double t_prev = 0.0;  // the sums of chunk c - 1: written with chunk c's tile, so that no LDS write sits behind the table reads
                // The eight pair-row stores of a chunk are SPREAD over the iteration (S(j) below): the four wavefronts of a CU run this
                // code in step, and eight 1-KiB stores back to back from each of them is a 32 KB burst on the CU's store path with the
                // arithmetic standing still behind it (K1 with its stores in groups of four: 49 us against 41, profiles/r04_k1_chunk_ab.md).
                auto body = [&](int c, auto last_tag) {
                    constexpr bool kLast = decltype(last_tag)::value;
                    double r[kMixChunk];
                    auto S = [&](int j) {
                        if constexpr (STORE == 2) {
                            store_pair_rows<2>(op, r[2 * j], r[2 * j + 1]);
                            op += 2 * V;
                        }
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    osc_pipe_finish<WF, kMixChunk>(P, r, hd);  // (waits for the table reads of this chunk)
                    __builtin_amdgcn_sched_barrier(0);
                    S(0);
#pragma unroll
                    for (int i = 0; i < kMixChunk; i++) tw[i * kTileRow] = r[i];
                    tile_put(t_prev, (c - 1) * kMixChunk, c > 0 ? kMixChunk : 0);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    double2v xv[8];
                    tile_load(xv);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();  // (the next chunk's tile writes stay behind these reads)
                    __builtin_amdgcn_sched_barrier(0);
                    S(1);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!kLast) {
                        osc_pipe_phase_half<WF, kMixChunk, 0>(ph, q, P);
                        __builtin_amdgcn_sched_barrier(0);
                        S(2);
                        __builtin_amdgcn_sched_barrier(0);
                        osc_pipe_phase_half<WF, kMixChunk, 1>(ph, q, P);
                        __builtin_amdgcn_sched_barrier(0);
                        S(3);
                        __builtin_amdgcn_sched_barrier(0);
                        osc_pipe_fetch<WF, kMixChunk>(P, s_tab);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        S(2);
                        S(3);
                    }
                    S(4);
                    __builtin_amdgcn_sched_barrier(0);
                    double vl, vr;
                    tile_products(xv, vl, vr);
                    __builtin_amdgcn_sched_barrier(0);
                    S(5);
                    __builtin_amdgcn_sched_barrier(0);
                    t_prev = tile_fold(vl, vr);
                    __builtin_amdgcn_sched_barrier(0);
                    S(6);
                    S(7);
                };
                for (int c = 0; c + 1 < full; c++) body(c, std::false_type{});
                body(full - 1, std::true_type{});
                tile_put(t_prev, (full - 1) * kMixChunk, kMixChunk);
                if constexpr (STORE == 2) o += (size_t)full * kMixChunk * V;
                c_first = full * kMixChunk;
            }
        }
        for (int c0 = c_first; c0 < span; c0 += kMixChunk) {
            const int cnt = (span - c0) < kMixChunk ? (span - c0) : kMixChunk;
            auto chunk = [&](auto full_tag) {
                constexpr bool kFull = decltype(full_tag)::value;
                if constexpr (kFull && STORE == 2) {
                    // the 16 ticks first (osc_tick_chunk: every table read of the chunk in flight at once -- at one wavefront per SIMD the
                    // LDS latency of a tick is otherwise exposed sample by sample), stores and tile writes after them
                    double r[kMixChunk];
#ifndef MXG_K1M_CHUNK
#define MXG_K1M_CHUNK 1  // A/B: 0 = tick by tick
#endif
#if MXG_K1M_CHUNK
                    osc_tick_chunk<WF, kMixChunk>(ph, hd, q, s_tab, s_tab, r);
                    __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
                    for (int i = 0; i < kMixChunk; i++) r[i] = osc_tick<WF, false, kTickLean>(ph, hd, q, s_tab, s_tab);
#endif
#pragma unroll
                    for (int i = 0; i < kMixChunk; i += 2) {
                        store_pair_rows<2>(op, r[i], r[i + 1]);
                        op += 2 * V;
                        tw[i * kTileRow] = r[i];
                        tw[(i + 1) * kTileRow] = r[i + 1];
                    }
                    o += (size_t)kMixChunk * V;
                } else if constexpr (kFull) {
                    double r[kMixChunk];
                    osc_tick_chunk<WF, kMixChunk>(ph, hd, q, s_tab, s_tab, r);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < kMixChunk; i++) {
                        if constexpr (STORE != 0) {
                            *o = r[i];
                            o += V;
                        }
                        tw[i * kTileRow] = r[i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kMixChunk; i++) {
                        double r = 0.0;
                        if (kFull || i < cnt) {  // ragged last chunk: the state must not advance past N
                            r = osc_tick<WF, false, kTickLean>(ph, hd, q, s_tab, s_tab);
                            if constexpr (STORE != 0) {
                                *o = r;
                                o += V;
                            }
                        }
                        tw[i * kTileRow] = r;
                    }
                    if constexpr (STORE == 2) op += (size_t)kMixChunk * V;  // (only the bank's last chunk comes here)
                }
                // the wavefront's LDS operations execute in order: the transposed reads below see the writes above
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double2v xv[8];
                tile_load(xv);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // (the next chunk's writes stay behind these reads)
                tile_sum(xv, c0, cnt);
            };
            if (cnt == kMixChunk) chunk(std::true_type{}); else chunk(std::false_type{});
        }
        // 4 wavefront sums of this window -> one partial row per workgroup, added in the order wave 0, 1, 2, 3
        __syncthreads();
        double *prow = partial + wg * N * 2 + n0 * 2;
        for (int i = threadIdx.x; i < span * 2; i += blockDim.x)
            prow[i] = ((s_part[i] + s_part[WIN * 2 + i]) + s_part[2 * WIN * 2 + i]) + s_part[3 * WIN * 2 + i];
        __syncthreads();
    }
    if (blockIdx.y + 1 == gridDim.y && part_wait(part_ctr, psync)) {
        phase_io[v] = ph;
        hold_io[v] = hd;
    }
    }  // passes (time parts are launched with one pass only)
}

// ---- K1m, producer / consumer form (round 4) ----------------------------------------------------------------------------
// One wavefront per SIMD cannot hide anything from itself: in K1 the wavefront spends more than half of its time waiting for the
// store path to take its next store (the kernel is HBM-bound: that IS the roofline), and every instruction the voice sum adds to
// that wavefront is time on top (K1 41 us, the fused kernel above 50 us, its arithmetic alone 39 us: profiles/r04_k1m_sq.md).  Here
// the two jobs run on DIFFERENT wavefronts of the same SIMD: a workgroup is four PRODUCERS (K1's instruction stream -- tick, pair-row
// store -- plus one ds_write_b64 per sample into a tile) and four CONSUMERS (the transposed tile read, 16 + 16 multiply-adds per
// lane, the fold: the arithmetic of the fused kernel's second half), consumer w + 4 serving producer w.  The consumer's instructions
// issue while its producer waits for the store path, so the block costs what K1 costs plus the tile writes.
// Hand-off: per pair a ring of kPcRing tiles in LDS and two counters, `prod` (chunks written) and `cons` (chunks read), single
// writer each.  The LDS executes a wavefront's operations in order, so the counter store that follows the tile writes is seen after
// them, and the counter store that follows the tile READS is performed after them; the polls are relaxed workgroup-scope loads with
// an s_sleep between them.  No barrier inside a window; the two barriers per window (combine of the four consumer rows) are shared
// by all eight wavefronts.  The sums, their order and therefore the rows' bits are the fused kernel's.
// (kPcRing, lds_flag_load / lds_flag_store: mxg_lanefold.h)
#ifndef MXG_PC_FL
#define MXG_PC_FL kTickLean  // A/B: the producers' tick flavour (0 = K1's own sinebuf tick: one table copy, the generic wrap)
#endif
#ifndef MXG_PC_ASMX
#define MXG_PC_ASMX 1  // A/B: 0 = K1's compiler-scheduled pair exchange in the producers
#endif
#ifndef MXG_PC_CSLEEP
#define MXG_PC_CSLEEP 4  // A/B: the consumer's poll interval (units of 64 clocks)
#endif
#ifndef MXG_PC_PRIO
#define MXG_PC_PRIO 2  // A/B: the producers' issue priority
#endif
constexpr int kPcFL = MXG_PC_FL;

template <int WF, int STORE, int WIN>
__global__ __launch_bounds__(512) void osc_mixpc_kernel(size_t V, size_t N, const double *__restrict__ freq,
                                                        const double *__restrict__ p1, const double *__restrict__ p2,
                                                        double *__restrict__ phase_io, double *__restrict__ hold_io,
                                                        double *__restrict__ out, const double *__restrict__ pan,
                                                        double *__restrict__ partial, double sr, int passes) {
    constexpr int kTab = tab_len<WF, kPcFL>();
    constexpr int kTabPad = (kTab + 1) & ~1;
    __shared__ __attribute__((aligned(16))) double s_all[kTabPad + 4 * kPcRing * kTileWave + 4 * WIN * 2 + 256 + 8];
    double *s_tab = s_all;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave < 4;
    const int pw = wave & 3;  // the pair
    // (round 6, as in K1: the first pass's per-voice loads are REQUESTED before the table is staged -- behind the two barriers of the
    // prologue they would be a round trip to memory of their own)
    double ph_pre = 0, hd_pre = 0, f_pre = 0, a_pre = 0, b_pre = 0;
    if (producer && (size_t)blockIdx.x * 256 < V) {
        const size_t vr = (size_t)blockIdx.x * 256 + (size_t)pw * 64 + lane;
        const size_t vv = vr < V ? vr : (STORE == 2 ? V - 2 + (vr & 1) : V - 1);
        ph_pre = phase_io[vv];
        hd_pre = hold_io[vv];
        f_pre = freq[vv];
        a_pre = p1 ? p1[vv] : 0.0;
        b_pre = p2 ? p2[vv] : 0.0;
    }
    load_tab<WF, kPcFL>(s_tab);
    double *ring = s_all + kTabPad + pw * (kPcRing * kTileWave);
    double *s_part = s_all + kTabPad + 4 * kPcRing * kTileWave;  // [4 pairs][WIN][2]
    double *my_part = s_part + pw * (WIN * 2);
    double *s_dump = s_part + 4 * WIN * 2;                       // [256]
    int *flags = reinterpret_cast<int *>(s_dump + 256);          // [4 pairs][2]: prod, cons
    int *f_prod = flags + 2 * pw, *f_cons = flags + 2 * pw + 1;
    const int ts = lane & 15, tq = lane >> 4;
    for (int pass = 0; pass < passes; pass++) {
        const size_t wg = (size_t)pass * gridDim.x + blockIdx.x;
        if (wg * 256 >= V) break;
        if (pass) __syncthreads();
        if (threadIdx.x < 8) flags[threadIdx.x] = 0;
        const size_t vraw = wg * 256 + (size_t)pw * 64 + lane;
        const bool live = vraw < V;
        const size_t v = live ? vraw : (STORE == 2 ? V - 2 + (vraw & 1) : V - 1);
        if (producer) {
            double x = pan[v];
            if (x > 1) x = 1;  // C:504
            if (x < 0) x = 0;  // C:505
            ring[lane] = live ? sqrt(1.0 - x) : 0.0;  // two[0] = input*sqrt(1.0-x)   C:506
            ring[64 + lane] = live ? sqrt(x) : 0.0;   // two[1] = input*sqrt(x)       C:507
        }
        __syncthreads();  // (the table, the gains, the counters)
        double gl[16], gr[16];
        if (!producer) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                gl[j] = ring[16 * tq + j];
                gr[j] = ring[64 + 16 * tq + j];
            }
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" : "+v"(gl[j]), "+v"(gr[j]));
        }
        __syncthreads();  // every consumer has its gains: the ring is free
        if (producer) {
            double ph = pass ? phase_io[v] : ph_pre, hd = pass ? hold_io[v] : hd_pre;
            OscPre q = osc_pre<WF>(pass ? freq[v] : f_pre, sr, pass ? (p1 ? p1[v] : 0.0) : a_pre, pass ? (p2 ? p2[v] : 0.0) : b_pre);
            asm volatile("" : "+v"(ph), "+v"(hd));
            asm volatile("" : "+v"(q.inc), "+v"(q.k), "+v"(q.p1), "+v"(q.p2));
            double *o = out + v;
            double *op = out + (size_t)(threadIdx.x & 1) * V + (v & ~(size_t)1);
            int k = 0;
            int seen = 0;  // `cons` as last read: requested at the start of a chunk, looked at one chunk later (no wait on the way)
            __builtin_amdgcn_s_setprio(MXG_PC_PRIO);  // the store-bound stream is the critical one: the consumer takes the issue slots it leaves
            for (size_t n0 = 0; n0 < N; n0 += WIN) {
                const int span = (int)((N - n0) < (size_t)WIN ? (N - n0) : (size_t)WIN);
                for (int c0 = 0; c0 < span; c0 += kMixChunk, k++) {
                    const int cnt = (span - c0) < kMixChunk ? (span - c0) : kMixChunk;
                    double *tw = ring + (k % kPcRing) * kTileWave + tq * kTileQuarter + ts;
                    // the consumer must be done with the tile this chunk overwrites: it normally was a chunk ago (`seen`)
                    if (seen < k - kPcRing + 1)
                        while ((seen = lds_flag_load(f_cons)) < k - kPcRing + 1) __builtin_amdgcn_s_sleep(1);
                    const int seen_next = lds_flag_load(f_cons);
                    asm volatile("" ::: "memory");
                    if (cnt == kMixChunk && STORE == 2) {
#pragma unroll
                        for (int i = 0; i < kMixChunk; i += 2) {
                            const double r0 = osc_tick<WF, false, kPcFL>(ph, hd, q, s_tab, s_tab);
                            const double r1 = osc_tick<WF, false, kPcFL>(ph, hd, q, s_tab, s_tab);
                            store_pair_rows<2, MXG_PC_ASMX != 0>(op, r0, r1);
                            op += 2 * V;
#ifndef MXG_PC_NOTILE  // (A/B: the producer without its tile writes)
                            tw[i * kTileRow] = r0;
                            tw[(i + 1) * kTileRow] = r1;
#endif
                        }
                        o += (size_t)kMixChunk * V;
                    } else {
#pragma unroll
                        for (int i = 0; i < kMixChunk; i++) {
                            double r = 0.0;
                            if (i < cnt) {  // ragged last chunk: the state must not advance past N
                                r = osc_tick<WF, false, kPcFL>(ph, hd, q, s_tab, s_tab);
                                if constexpr (STORE != 0) {
                                    *o = r;
                                    o += V;
                                }
                            }
                            tw[i * kTileRow] = r;
                        }
                        if constexpr (STORE == 2) op += (size_t)kMixChunk * V;
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    lds_flag_store(f_prod, k + 1);  // (behind the tile writes in the LDS queue)
                    seen = seen_next;
                }
                __syncthreads();
                double *prow = partial + wg * N * 2 + n0 * 2;
                for (int i = threadIdx.x; i < span * 2; i += blockDim.x)
                    prow[i] = ((s_part[i] + s_part[WIN * 2 + i]) + s_part[2 * WIN * 2 + i]) + s_part[3 * WIN * 2 + i];
                __syncthreads();
            }
            __builtin_amdgcn_s_setprio(0);
            phase_io[v] = ph;
            hold_io[v] = hd;
        } else {
            int k = 0;
            for (size_t n0 = 0; n0 < N; n0 += WIN) {
                const int span = (int)((N - n0) < (size_t)WIN ? (N - n0) : (size_t)WIN);
                for (int c0 = 0; c0 < span; c0 += kMixChunk, k++) {
                    const int cnt = (span - c0) < kMixChunk ? (span - c0) : kMixChunk;
                    while (lds_flag_load(f_prod) <= k) __builtin_amdgcn_s_sleep(MXG_PC_CSLEEP);
                    asm volatile("" ::: "memory");
                    const double2v *tr =
                        reinterpret_cast<const double2v *>(ring + (k % kPcRing) * kTileWave + tq * kTileQuarter + ts * kTileRow);
                    double2v xv[8];
#ifdef MXG_PC_NOCONSUME  // (A/B: the consumer only keeps the counters moving)
                    lds_flag_store(f_cons, k + 1);
                    continue;
#endif
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
                    double *dst = ((lane & 16) == 0 && ts < cnt) ? my_part + (c0 + ts) * 2 + (lane >> 5) : s_dump + (threadIdx.x & 255);
                    *dst = t;
                }
                __syncthreads();
                double *prow = partial + wg * N * 2 + n0 * 2;
                for (int i = threadIdx.x; i < span * 2; i += blockDim.x)
                    prow[i] = ((s_part[i] + s_part[WIN * 2 + i]) + s_part[2 * WIN * 2 + i]) + s_part[3 * WIN * 2 + i];
                __syncthreads();
            }
        }
    }
}

typedef void (*osc_mixpc_fn)(size_t, size_t, const double *, const double *, const double *, double *, double *, double *,
                             const double *, double *, double, int);
// win: the combine window.  512 (a whole block of the usual length in ONE window: the producers meet a barrier only at the end of the
// block, instead of waiting for the consumers' last chunk twice per block) where the waveform's table leaves room for [4][512][2] sums in
// the 160 KB; 256 otherwise.
template <int WF>
constexpr bool mixpc_fits_512() {
    return ((((tab_len<WF, kPcFL>() + 1) & ~1) + 4 * kPcRing * kTileWave + 4 * 512 * 2 + 256 + 8) * sizeof(double)) <= 160 * 1024;
}
template <int WF>
osc_mixpc_fn pick_mixpc(int store, int win) {
    if constexpr (mixpc_fits_512<WF>()) {
        if (win == 512)
            return store == 2 ? osc_mixpc_kernel<WF, 2, 512> : (store == 1 ? osc_mixpc_kernel<WF, 1, 512> : osc_mixpc_kernel<WF, 0, 512>);
    }
    return store == 2 ? osc_mixpc_kernel<WF, 2, 256> : (store == 1 ? osc_mixpc_kernel<WF, 1, 256> : osc_mixpc_kernel<WF, 0, 256>);
}
osc_mixpc_fn pick_mixpc_wf(int wf, int store, int win) {
    switch (wf) {
        case 0: return pick_mixpc<0>(store, win);
        case 1: return pick_mixpc<1>(store, win);
        case 2: return pick_mixpc<2>(store, win);
        case 3: return pick_mixpc<3>(store, win);
        case 4: return pick_mixpc<4>(store, win);
        case 5: return pick_mixpc<5>(store, win);
        case 6: return pick_mixpc<6>(store, win);
        case 7: return pick_mixpc<7>(store, win);
        case 8: return pick_mixpc<8>(store, win);
        case 9: return pick_mixpc<9>(store, win);
        case 10: return pick_mixpc<10>(store, win);
        case 11: return pick_mixpc<11>(store, win);
    }
    return nullptr;
}

typedef void (*osc_mix_fn)(size_t, size_t, const double *, const double *, const double *, double *, double *,
                           double *, const double *, double *, double, PartSync, int);
// store: 0 none, 1 plain, 2 pair rows (sc1); win: samples per workgroup combine (128 where three workgroups must share a CU)
template <int WF>
osc_mix_fn pick_mix(int store, int win) {
    if (win == 128)
        return store == 2 ? osc_mix_kernel<WF, 2, 128> : (store == 1 ? osc_mix_kernel<WF, 1, 128> : osc_mix_kernel<WF, 0, 128>);
    return store == 2 ? osc_mix_kernel<WF, 2, 256> : (store == 1 ? osc_mix_kernel<WF, 1, 256> : osc_mix_kernel<WF, 0, 256>);
}
osc_mix_fn pick_mix_wf(int wf, int store, int win) {
    switch (wf) {
        case 0: return pick_mix<0>(store, win);
        case 1: return pick_mix<1>(store, win);
        case 2: return pick_mix<2>(store, win);
        case 3: return pick_mix<3>(store, win);
        case 4: return pick_mix<4>(store, win);
        case 5: return pick_mix<5>(store, win);
        case 6: return pick_mix<6>(store, win);
        case 7: return pick_mix<7>(store, win);
        case 8: return pick_mix<8>(store, win);
        case 9: return pick_mix<9>(store, win);
        case 10: return pick_mix<10>(store, win);
        case 11: return pick_mix<11>(store, win);
    }
    return nullptr;
}

typedef void (*osc_fn)(size_t, size_t, const double *, const double *, const double *, double *,
                       double *, double *, double, PartSync, int, int, int, size_t, size_t, size_t, unsigned *, unsigned, unsigned *);

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

namespace mxg {
namespace {
// one launch of K1 over the voices [v_begin, v_end) of a bank
constexpr size_t kPacedFrom = 90112, kPacedFromSinebuf = 122880, kPacedTo = 327680;  // K1 on the controlled schedule: bank sizes (mxg_osc_render_pitch)
struct OscLaunch {
    int waveform = 0;
    size_t V = 0, N = 0, P = 0;  // P: row pitch of `out` in doubles
    const double *freq = nullptr, *p1 = nullptr, *p2 = nullptr;
    int fps = 0;
    double *phase = nullptr, *hold = nullptr, *out = nullptr;
    hipStream_t st = nullptr;
    bool pairs_ok = false;
    size_t v_begin = 0, v_end = 0;
    int vpl = 1, store = 0, xcd = 0;  // store: osc.hip pick<WF> numbering (0 plain 8 B ...)
    int split = 0, passes = 0, block = 256;  // 0 = automatic
    bool paced = false;                      // the controlled schedule (mxg_pace.h): one voice per lane, one part, one pass
    bool trial = false;                      // free-running or paced, decided by a trial on the device (mxg_pace.h, PaceTrial)
};

// The store stream of ONE launch over `count` voices, by waveform class and size (MI355X, 512-sample blocks, destination rotated;
// profiles/r03_osc_store.md, profiles/r04_osc_grid.md; fraction of the 8 TB/s peak on 8 B per sample):
//   pair rows = ONE voice per lane, two samples of a lane pair exchanged into one write-through (sc1) 16-byte store per lane;
//   2v        = TWO voices per lane, write-through 16-byte stores (half the wavefronts).
//   table forms (sinebuf, sawn):  < 49 152 voices plain 8-byte stores (the block lives in the caches);
//       < 81 920 pair rows (65 536: 51 -> 40 us, 0.66 -> 0.84); from there 2v (81 920: 61 us; 98 304: 63.5, 0.79; 131 072: 88, 0.76);
//   ramps (phasor, saw, triangle, square, pulse, impulse, phasorBetween): 2v already from 65 536 voices (saw: 49 -> 40 us);
//   VALU-heavy forms (sinewave, coswave, sinebuf4): pair rows at every size (65 536 sinewave: 69 -> 65, sinebuf4 60 -> 57), from
//   262 144 voices XCD-contiguous workgroup numbering.
void osc_single_rule(OscLaunch &L, size_t count) {
    const int wf = L.waveform;
    const bool heavy = wf == MXG_OSC_SINEWAVE || wf == MXG_OSC_COSWAVE || wf == MXG_OSC_SINEBUF4;
    const bool table = wf == MXG_OSC_SINEBUF || wf == MXG_OSC_SAWN;
    const size_t bytes = count * L.N * sizeof(double);
    L.vpl = 1;
    L.store = 0;
    L.xcd = 0;
    L.passes = 0;
    if (L.pairs_ok) {
        if (heavy) {
            if (bytes >= ((size_t)32 << 20)) L.store = 3;
            L.xcd = count >= 262144 ? 1 : 0;
        } else if (bytes >= ((size_t)192 << 20)) {
            // (tools/sweep_osc_mid.py, late round 4: between the 65 536- and the 98 304-voice shapes -- 2.1 to 2.7 wavefronts of 128
            // voices per CU -- sinebuf is fastest as TWO passes of half the range with one voice per lane and non-temporal 8-byte
            // stores: 69 632 ... 81 920 voices 57-59 us against 64-67; sawn keeps its pair rows -- the lean loop -- up to 98 304:
            // 64.6 against 81.0 us at 81 920)
            if (wf == MXG_OSC_SINEBUF && count >= 67584 && count < 88064) {
                L.store = 1;
                L.passes = 2;
            } else if (count >= (wf == MXG_OSC_SAWN ? 98304u : table ? 81920u : 65536u)) {
                L.vpl = 2;
                L.store = 2;
                L.xcd = count >= 262144 ? 1 : 0;
            } else {
                L.store = 3;
            }
        }
    }
}

int osc_launch(const OscLaunch &L) {
    int vpl = L.vpl, store = L.store, block = L.block ? L.block : 256;
    const bool fps = L.fps != 0;
    const size_t count = L.v_end - L.v_begin;
    if (count == 0) return MXG_OK;
    const bool pairs_ok = L.pairs_ok && !(L.v_begin & 1) && !(count & 1);
    if (!pairs_ok) vpl = 1;
    if (vpl == 2 && store > 2) store = 0;
    if (vpl == 1 && store >= 2 && !pairs_ok) store = store == 4 ? 1 : 0;  // pair rows need whole pairs
    osc_fn fn = pick_wf(L.waveform, fps, vpl, store);
    const size_t lanes = (count + vpl - 1) / vpl;
    // time parts.  (a) Where the output dominates the recurrence (sinewave, coswave, sinebuf4) and the bank is too small to give every
    // SIMD two wavefronts by itself: two parts.  (b) SMALL banks (fewer wavefronts than the 1024 SIMDs) of the waveforms whose phase
    // skip is much cheaper than their tick (the table oscillators: sinebuf, sinebuf4, sawn, sinewave, coswave): a block is one chain of
    // N dependent steps per wavefront, 27-30 us for 512 samples however few voices there are; cut into up to eight parts it is 14.6 us
    // at 1024 voices, 17.0 at 4096, 20.1 at 16 384 (sinebuf; profiles/r03_small_osc_banks.md).  Same bits (the skip is the same
    // additions); the other waveforms' tick IS their recurrence, parts would only repeat it.
    int split = L.split;
    if (split == 0) {
        split = 1;
        const size_t waves = (lanes + 63) / 64;
        const int wf = L.waveform;
        const bool heavy = wf == MXG_OSC_SINEWAVE || wf == MXG_OSC_COSWAVE || wf == MXG_OSC_SINEBUF4;
        const bool table = heavy || wf == MXG_OSC_SINEBUF || wf == MXG_OSC_SAWN;
        if (!fps && heavy) split = waves >= 2048 ? 1 : 2;  // (sinewave at 65 536 voices: 60-62 us in two parts; one part 65-69, four 64)
        if (!fps && table && waves < 1024) {
            int want = 1;
            while (want < 8 && (size_t)(2 * want) * waves <= 1024 && (size_t)(2 * want) * 64 <= L.N) want *= 2;  // (a part renders >= 64 samples)
            if (want > split) split = want;
        }
    }
    if (fps) split = 1;
    // every part must render at least one sample: the last part's ticks leave the member `output` of the final sample
    while (split > 1 && (size_t)(split - 1) * ((L.N + split - 1) / split) >= L.N) split--;
    // passes: voice groups a wavefront renders one after the other (the grid covers 1 / passes of the range)
    int passes = L.passes > 0 ? L.passes : 1;
    if (split > 1) passes = 1;
    size_t nblk = (lanes + block - 1) / block;
    if ((size_t)passes > nblk) passes = (int)nblk;
    nblk = (nblk + passes - 1) / passes;
    dim3 grid((unsigned)nblk, (unsigned)split), blk((unsigned)block);
    PartSync psync;
    if (split > 1)
        if (int s = part_sync_get(L.st, (size_t)grid.x * ((block + 63) / 64), split, &psync)) return s;
    // the paced schedule (mxg_pace.h): L.paced = the controller (mxg_osc_render_pitch decides where); knob osc_pace >= 2 = a fixed period in
    // ticks of 10 ns per 8 samples (sweeps).  One part, one pass, one voice per lane, a waveform whose time is its store stream's.
    unsigned *pace_ctl = nullptr;
    unsigned pace_arg = 0;
    {
        const int knob = tune_get("osc_pace");
        const bool lean = ((MXG_K1_LEAN_MASK >> L.waveform) & 1) != 0;
        if (split == 1 && passes == 1 && !fps && vpl == 1 && !lean) {
            if (knob >= 2) {
                pace_arg = (unsigned)knob;
            } else if (L.paced) {
                pace_arg = pace_start_period(count * 8 * 8);  // the chip's 8 rows
                unsigned *base = pace_words(SCR_OSC_PACE, L.st, 16 * kPaceWords);
                if (base && pace_arg) pace_ctl = base + kPaceWords * (L.waveform & 15);
                else pace_arg = 0;  // (inside a graph capture before the first eager launch, or a device whose counter's rate is unknown: not paced)
            }
        }
    }
    // the trial (mxg_pace.h, PaceTrial): sinebuf at the headline's size, as the automatic rule launches it (pair rows, write-through, one
    // voice per lane, one part, one pass) -- the same kernel also holds the paced 8-byte stream, and eight words on the device decide
    // between them, and on the period, by the launches' measured durations
    unsigned *trial = nullptr;
    if (L.trial && !pace_arg && split == 1 && passes == 1 && !fps && vpl == 1 && store == 3 && tune_get("osc_pace") == 0) {
        unsigned *base = pace_words(SCR_OSC_PACE, L.st, 16 * kPaceWords);
        if (base && pace_start_period(count * 8 * 8)) {
            trial = base + kPaceWords * 14;
            pace_arg = pace_start_period(count * 8 * 8);  // (the candidates are 31/32, 29/32, 27/32 of it: 62, 58, 54 at 65 536 voices)
        }
    }
    KernelTimer kt("osc_kernel", L.st);
    hipLaunchKernelGGL(fn, grid, blk, 0, L.st, L.V, L.N, L.freq, L.p1, L.p2, L.phase, L.hold, L.out, (double)settings().sampleRate, psync,
                       L.xcd, L.fps == 2 ? 1 : 0, passes, L.v_begin, L.v_end, L.P ? L.P : L.V, pace_ctl, pace_arg, trial);
    return check_hip(hipGetLastError(), "osc_kernel launch");
}
}  // namespace
}  // namespace mxg

// (diagnostics, not in maxigpu.h: K1's pace controllers of a stream -- one per waveform, kPaceWords words each)
extern "C" int mxg_debug_osc_pace(void *stream, unsigned *host) {
    hipStream_t st = mxg::resolve_stream(stream);
    unsigned *base = nullptr;
    bool fresh = false;
    if (int s = mxg::scratch_get(mxg::SCR_OSC_PACE, st, 16 * mxg::kPaceWords * sizeof(unsigned), (void **)&base, &fresh)) return s;
    if (fresh) MXG_HIP(hipMemsetAsync(base, 0, 16 * mxg::kPaceWords * sizeof(unsigned), st));
    MXG_HIP(hipStreamSynchronize(st));
    return mxg::check_hip(hipMemcpy(host, base, 16 * mxg::kPaceWords * sizeof(unsigned), hipMemcpyDeviceToHost), "pace words");
}

extern "C" int mxg_osc_render(int waveform, size_t V, size_t N, const double *d_freq, int fps,
                              const double *d_p1, const double *d_p2, double *d_phase,
                              double *d_outhold, double *d_out, void *stream) {
    return mxg_osc_render_pitch(waveform, V, N, d_freq, fps, d_p1, d_p2, d_phase, d_outhold, d_out, V * sizeof(double), stream);
}

extern "C" int mxg_osc_render_pitch(int waveform, size_t V, size_t N, const double *d_freq, int fps,
                                    const double *d_p1, const double *d_p2, double *d_phase,
                                    double *d_outhold, double *d_out, size_t out_pitch_bytes, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(out_pitch_bytes % sizeof(double) == 0 && out_pitch_bytes >= V * sizeof(double),
                "out_pitch_bytes must be a multiple of 8 and at least V * 8");
    const size_t P = out_pitch_bytes / sizeof(double);
    MXG_REQUIRE(waveform >= 0 && waveform <= 11, "unknown waveform");
    MXG_REQUIRE(d_freq && d_phase && d_outhold && d_out, "null device pointer");
    MXG_REQUIRE(waveform != MXG_OSC_PULSE || d_p1, "pulse needs d_p1 (duty)");
    MXG_REQUIRE(waveform != MXG_OSC_PHASORBETWEEN || (d_p1 && d_p2),
                "phasorBetween needs d_p1/d_p2 (start/end phase)");
    MXG_REQUIRE(fps >= 0 && fps <= 2 && (fps != 2 || d_p1), "fps is 0, 1 (d_freq [N][V]) or 2 (d_freq and d_p1 [N][V])");
    MXG_REQUIRE(N <= 0xffffffffull, "a block holds at most 2^32 - 1 samples per voice");
    if (V == 0 || N == 0) return MXG_OK;
    // the store stream: knobs osc_vpl, osc_store, osc_xcd, osc_passes, osc_split (0 = automatic each); left alone, osc_single_rule / the plan
    // of launches below
    const bool pairs_ok = !fps && !(V & 1) && !(P & 1) && !(((uintptr_t)d_out) & 15);  // (16-byte pair rows: every row 16-byte aligned)
    hipStream_t st = resolve_stream(stream);
    OscLaunch L0;
    L0.waveform = waveform; L0.V = V; L0.P = P; L0.N = N; L0.freq = d_freq; L0.fps = fps; L0.p1 = d_p1; L0.p2 = d_p2; L0.phase = d_phase;
    L0.hold = d_outhold; L0.out = d_out; L0.st = st; L0.pairs_ok = pairs_ok;
    int vpl = tune_get("osc_vpl"), store = tune_get("osc_store") - 1, xcd = tune_get("osc_xcd") - 1;  // (knob value 0 = automatic)
    const bool automatic = vpl == 0 && store < 0;
    const bool heavy = waveform == MXG_OSC_SINEWAVE || waveform == MXG_OSC_COSWAVE || waveform == MXG_OSC_SINEBUF4;
    // (Row pitches that are multiples of 2 MB -- 262 144 voices and its multiples -- take the passes badly: 524 288 voices 0.56, 1 048 576
    // 0.64 against 0.68-0.71 for one launch over the whole bank; 262 144 itself wants the XCD-contiguous numbering, 0.80 against 0.73.
    // profiles/r04_osc_grid.md.)
    // ---- the paced schedule (round 6, mxg_pace.h; profiles/r06_pace.md) -------------------------------------------------------------------
    // Between 90 112 and 327 680 voices the TABLE-FREE store-bound waveforms (phasor, saw, triangle, square, pulse, impulse, phasorBetween)
    // stream best as the SIMPLEST launch -- one voice per lane, one pass, non-temporal 8-byte stores -- on the controlled schedule: eight
    // samples every P ticks, P on the knee of the memory system's rate.  Measured with the controller (saw / square, us per 512-sample
    // block, against the plan of launches below, which round 4 tuned for the free-running kernel): 98 304 voices 63.6-65.3 -> 60.0-61.9,
    // 131 072 88.2-89.3 -> 78.1-78.7 (0.76 -> 0.86 of 8 TB/s), 196 608 135-141 -> 114.4-115.0 (0.88), 262 144 180-183 -> 153.3-155.6.
    // Not below (69 632 ... 81 920 voices: the tuned launches are as fast or faster, 42-54 us against 51-56), not beyond (393 216 mixed;
    // 524 288: 385 -> 630 us, the grid is no longer resident at once).  sinebuf from 122 880 voices, under the TOLERANT rule of the
    // controller: its launches go late now and then at ANY period (one in ten or twenty), which the strict rule reads as the knee and
    // parks far above it (131 072 voices: P = 128, 87.8 us); with eight late launches of 32 for a tick up and up to four for a tick
    // down it holds P = 107-108, the optimum of the fixed-period sweep: 131 072 voices 88.2-89.9 -> 79.4-79.6 us (0.76 -> 0.85), 196 608
    // 126.0-126.6 -> 119.9-120.2, 262 144 168.1-168.3 -> 160.0-160.3; at 98 304 voices the plan is as fast (63.6 against 63.4-64.0), and at
    // 65 536 K1's own stream (pair rows, write-through) already sits where the paced one ends up (40.8-41.2 us against 41.1-42.8).
    // Knob osc_pace: 0 automatic, 1 never, >= 2 a fixed period.
    const bool lean_wf = ((MXG_K1_LEAN_MASK >> waveform) & 1) != 0 || (waveform == MXG_OSC_SINEBUF && V < kPacedFromSinebuf);
    if (automatic && !fps && !lean_wf && xcd < 0 && tune_get("osc_pace") == 0 && tune_get("osc_passes") == 0 && tune_get("osc_split") == 0 &&
        tune_get("osc_plan") == 0 && V >= kPacedFrom && V <= kPacedTo && pace_start_period(V * 8 * 8) &&
        pace_words(SCR_OSC_PACE, st, 16 * kPaceWords)) {  // (no schedule to be had -- a capture before the stream's first eager launch, a device
                                                            // that is not the whole chip: the plan below is the faster free-running launch)
        OscLaunch A = L0;
        A.v_begin = 0; A.v_end = V; A.vpl = 1; A.store = 1; A.xcd = 0; A.split = 1; A.passes = 1; A.block = 256; A.paced = true;
        return osc_launch(A);
    }
    int plan = tune_get("osc_plan");  // 0 automatic, 1 never, 2 / 3 always (main launch with natural / XCD-contiguous numbering)
    if (plan == 0) plan = (P % 262144) ? 2 : (V == 262144 ? 3 : 1);  // (the PITCH decides: padded rows take the passes well)
    if (automatic && pairs_ok && !heavy && xcd < 0 && tune_get("osc_passes") == 0 && tune_get("osc_split") == 0 && plan != 1 &&
        V * N * sizeof(double) >= ((size_t)352 << 20)) {
        // ---- large banks of the store-bound waveforms: a plan of launches (round 4, profiles/r04_osc_grid.md) ----------------------------
        // What streams best is a grid of exactly THREE wavefronts per CU, two voices per lane, every wavefront walking down the same
        // rows at the same time: 768 wavefronts = 98 304 voices per pass at 0.79-0.80 of the HBM peak, pass after pass (196 608 voices in
        // two passes 126 us, 393 216 in four 251 us) -- where one wavefront per 128 voices over the whole bank gives 1536, 3072 ...
        // wavefronts that drift apart over megabyte rows (0.62-0.69), and uneven grids (683, 640 wavefronts) leave CUs idle.  So: as
        // many 98 304-voice passes as fit, in one launch; the remainder as a second launch in the best shape for ITS size (the
        // single-launch rules below), unless a 131 072-voice tail (1024 wavefronts, 0.75-0.77) is cheaper than 98 304 + a small rest.
        const size_t kPass = 98304;
        size_t k = V / kPass;
        size_t rest = V - k * kPass;
        if (k > 0 && rest > 0 && rest < 40960 && rest + kPass <= 131072) {  // e.g. 131 072 = one 1024-wavefront launch, not 98 304 + 32 768
            k--;
            rest += kPass;
        }
        if (k > 0) {
            OscLaunch A = L0;
            A.v_begin = 0; A.v_end = k * kPass; A.vpl = 2; A.store = 2; A.xcd = plan == 3 ? 1 : 0; A.split = 1; A.passes = (int)k; A.block = 256;
            if (int s2 = osc_launch(A)) return s2;
        }
        if (rest > 0) {
            OscLaunch B = L0;
            B.v_begin = k * kPass; B.v_end = V;
            osc_single_rule(B, rest);
            if (int s2 = osc_launch(B)) return s2;
        }
        return MXG_OK;
    }
    OscLaunch A = L0;
    A.v_begin = 0; A.v_end = V;
    if (automatic) {
        osc_single_rule(A, V);
        if (xcd >= 0) A.xcd = xcd;
        A.trial = waveform == MXG_OSC_SINEBUF && !fps && V >= 57344 && V < 73728;  // (every SIMD one wavefront: the headline's shape)
    } else {
        if (vpl == 0) vpl = 1;
        if (store < 0) {  // (the round-2 rule for 8-byte stores, knob osc_nt: non-temporal by block size)
            const size_t out_bytes = V * N * sizeof(double);
            const int nt_knob = tune_get("osc_nt");
            store = (nt_knob == 1 || (nt_knob == 2 && out_bytes > ((size_t)300 << 20) && out_bytes <= ((size_t)1200 << 20))) ? 1 : 0;
        }
        A.vpl = vpl; A.store = store; A.xcd = xcd < 0 ? 0 : xcd;
    }
    A.block = tune_get("osc_block");
    A.split = tune_get("osc_split");
    if (const int kp = tune_get("osc_passes")) A.passes = kp;  // (else what the automatic rule chose, or one)
    return osc_launch(A);
}

namespace mxg {
namespace {
// K1m launch: render (+ optional per-voice block) and the per-workgroup mix rows d_rows[(V + 255) / 256][N][2]
int osc_mix_launch(int waveform, size_t V, size_t N, const double *d_freq, const double *d_p1, const double *d_p2,
                   double *d_phase, double *d_outhold, double *d_out, const double *d_pan, double *d_rows, hipStream_t st) {
    const int block = 256;
    const size_t nblocks = (V + block - 1) / block;
    // per-voice block: pair rows of write-through 16-byte stores where whole pairs exist (knob osc_mix_store: 0 automatic,
    // 1 plain 8-byte stores, 2 pair rows); time parts (knob osc_mix_split: 0 automatic)
    int store = 0;
    if (d_out) {
        const bool pairs_ok = !(V & 1) && !(((uintptr_t)d_out) & 15) && V >= 2;
        const int knob = tune_get("osc_mix_store");
        store = (pairs_ok && (knob == 2 || (knob == 0 && V * N * sizeof(double) >= ((size_t)32 << 20)))) ? 2 : 1;
    }
    int split = tune_get("osc_mix_split");
    if (split == 0) {
        split = 1;
        // SMALL banks of the table oscillators, as in mxg_osc_render: fewer wavefronts than SIMDs, one chain of N dependent
        // steps each -- up to eight parts
        const bool table = waveform == MXG_OSC_SINEWAVE || waveform == MXG_OSC_COSWAVE || waveform == MXG_OSC_SINEBUF4 ||
                           waveform == MXG_OSC_SINEBUF || waveform == MXG_OSC_SAWN;
        const size_t waves = nblocks * 4;
        if (table && waves < 1024)
            while (split < 8 && (size_t)(2 * split) * waves <= 1024 && (size_t)(2 * split) * 64 <= N) split *= 2;  // (a part renders >= 64 samples)
    }
    while (split > 1 && (size_t)(split - 1) * (((N + split - 1) / split + kMixChunk - 1) / kMixChunk * kMixChunk) >= N) split--;
    PartSync psync;
    if (split > 1)
        if (int s2 = part_sync_get(st, nblocks * 4, split, &psync)) return s2;
    // the combine window: 256 samples (57 KB of LDS: two workgroups per CU) up to 131 072 voices, 128 (49 KB: three) beyond
    int win = tune_get("osc_mix_win");
    if (win == 0) win = nblocks * (size_t)split > 512 ? 128 : 256;  // (TODO after the passes sweep: by workgroups per CU)
    osc_mix_fn fn = pick_mix_wf(waveform, store, win);
    int passes = tune_get("osc_mix_passes");  // voice groups a workgroup renders one after the other (0 automatic)
    if (passes == 0) passes = 1;
    if (split > 1) passes = 1;
    if ((size_t)passes > nblocks) passes = (int)nblocks;
    const size_t grid_x = (nblocks + passes - 1) / passes;
    // the producer / consumer form (knob osc_mix_pc: 0 automatic, 1 off, 2 on): whole blocks (no time parts)
    const int pc = tune_get("osc_mix_pc");
    if (split == 1 && (pc == 2 || (pc == 0 && nblocks >= 128))) {
        KernelTimer kt("osc_mix_kernel", st);
        int pcwin = tune_get("osc_mix_pcwin");  // 0 automatic, 256, 512
        if (pcwin == 0) pcwin = 512;
        hipLaunchKernelGGL(pick_mixpc_wf(waveform, store, pcwin), dim3((unsigned)grid_x), dim3(512), 0, st, V, N, d_freq, d_p1, d_p2, d_phase,
                           d_outhold, d_out, d_pan, d_rows, (double)settings().sampleRate, passes);
        return check_hip(hipGetLastError(), "osc_mixpc_kernel launch");
    }
    KernelTimer kt("osc_mix_kernel", st);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid_x, (unsigned)split), dim3(block), 0, st, V, N, d_freq, d_p1, d_p2, d_phase,
                       d_outhold, d_out, d_pan, d_rows, (double)settings().sampleRate, psync, passes);
    return check_hip(hipGetLastError(), "osc_mix_kernel launch");
}
}  // namespace
}  // namespace mxg

extern "C" size_t mxg_osc_mix_groups(size_t V) { return (V + 255) / 256; }

extern "C" int mxg_mix_rows_sum(size_t groups, size_t count, const double *d_rows, double *d_mix, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_mix && (d_rows || groups == 0), "null device pointer");
    if (count == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("mix_partials_kernel", st);
    hipLaunchKernelGGL(mix_partials_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * kPartWaves), 0, st, groups, count, d_rows,
                       d_mix);
    return check_hip(hipGetLastError(), "mix_partials_kernel launch");
}

extern "C" int mxg_osc_render_mix_rows(int waveform, size_t V, size_t N, const double *d_freq, const double *d_p1,
                                       const double *d_p2, double *d_phase, double *d_outhold, double *d_out,
                                       const double *d_pan, double *d_rows, void *stream) {
    using namespace mxg;
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(waveform >= 0 && waveform <= 11, "unknown waveform");
    MXG_REQUIRE(d_freq && d_phase && d_outhold && d_pan && d_rows, "null device pointer");
    MXG_REQUIRE(waveform != MXG_OSC_PULSE || d_p1, "pulse needs d_p1 (duty)");
    MXG_REQUIRE(waveform != MXG_OSC_PHASORBETWEEN || (d_p1 && d_p2), "phasorBetween needs d_p1/d_p2");
    if (N == 0 || V == 0) return MXG_OK;
    return osc_mix_launch(waveform, V, N, d_freq, d_p1, d_p2, d_phase, d_outhold, d_out, d_pan, d_rows, resolve_stream(stream));
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
    const size_t nblocks = mxg_osc_mix_groups(V);
    // a bank of one workgroup (<= 256 voices): its row IS the mix -- written in place, no second kernel (4-5 us of an 18 us
    // call at 64 voices)
    if (nblocks == 1) return osc_mix_launch(waveform, V, N, d_freq, d_p1, d_p2, d_phase, d_outhold, d_out, d_pan, d_mix, st);
    double *rows = nullptr;  // per-stream scratch: [nblocks][N][2] per-workgroup sums
    if (int s = scratch_get(SCR_OSC_MIX, st, sizeof(double) * (N * nblocks * 2 + 2), (void **)&rows)) return s;
    if (V)
        if (int s = osc_mix_launch(waveform, V, N, d_freq, d_p1, d_p2, d_phase, d_outhold, d_out, d_pan, rows, st)) return s;
    return mxg_mix_rows_sum(nblocks, N * 2, rows, d_mix, stream);
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
// The column walk that makes K1's store stream fast, for a kernel without state (round 4): a lane owns two adjacent voices and every OTHER
// row (even lanes the even rows, odd lanes the odd ones), so that a wavefront's accesses are 256 contiguous bytes of draws and 512 of
// output in each of two rows, and walks down the block; the draws of the next chunk of rows are requested before this chunk's stores
// (loads and stores retire in order on one counter).  A grid-stride element-wise kernel reaches 0.62 of 8 TB/s on its 12 B per sample
// whatever the access width -- with write-through 16-byte stores 0.30 -- this form [see profiles/r04_banks.md].  ST: store flavour.
typedef int int2v __attribute__((ext_vector_type(2)));
template <int ST, int U>
__global__ void __launch_bounds__(256) osc_noise_walk_kernel(size_t V, size_t N, const int32_t *__restrict__ rnd,
                                                             double *__restrict__ outhold, double *__restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;  // (V even: pairs are live or dead together; nothing crosses lanes here anyway)
    const size_t odd = threadIdx.x & 1, vp = v & ~(size_t)1;
    const int32_t *ip = rnd + vp;
    double *op = out + vp;
    // U: rows per lane and chunk (2 U rows of the block; U 8-byte loads in flight per lane: knob rw_chunk)
    const size_t rows = (N + 1 - odd) / 2;  // rows n = 2 j + odd < N
    auto row_at = [&](size_t j) { return (j < rows ? 2 * j + odd : (rows ? 2 * (rows - 1) + odd : 0)); };  // clamped: no branch
    if (rows == 0) return;
    int2v nx[U];
#pragma unroll
    for (int i = 0; i < U; i++) nx[i] = __builtin_nontemporal_load(reinterpret_cast<const int2v *>(ip + row_at(i) * V));
    for (size_t j0 = 0; j0 < rows; j0 += U) {
        int2v cur[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            cur[i] = nx[i];
            nx[i] = __builtin_nontemporal_load(reinterpret_cast<const int2v *>(ip + row_at(j0 + U + i) * V));
        }
#pragma unroll
        for (int i = 0; i < U; i++) {
            if (j0 + i >= rows) break;
            const size_t n = 2 * (j0 + i) + odd;
            const float r0 = (float)cur[i].x / 2147483648.0f, r1 = (float)cur[i].y / 2147483648.0f;
            const double o0 = (double)(r0 * 2.0f - 1.0f), o1 = (double)(r1 * 2.0f - 1.0f);
            store2<ST>(op + n * V, o0, o1);
            if (outhold && n + 1 == N) {  // the member `output`: the last row
                outhold[vp] = o0;
                outhold[vp + 1] = o1;
            }
        }
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
    hipStream_t st = resolve_stream(stream);
    KernelTimer kt("osc_noise_kernel", st);
    // the column walk where whole voice pairs exist (knob rw_store as for the read + write bank kernels: 0 automatic = write-through stores
    // for blocks from 64 MB, 1 the element-wise kernel, 2 / 3 / 4 plain / write-through / non-temporal 16-byte stores)
    int flavour = (!(V & 1) && !(((uintptr_t)d_rand) & 7)) ? rw_store_choice(V, N, d_out, RW_READ_WRITE) : 0;
    if (flavour == 2 && tune_get("rw_store") == 0 && V < 98304) flavour = 3;  // (measured: 65 536 voices 67.5 us non-temporal / 69.7 write-through; 131 072: 145 / 143)
    if (flavour) {
        const dim3 grid((unsigned)((V + 255) / 256));
        int chunk = tune_get("rw_chunk");
        if (chunk == 0) chunk = 8;
#define MXG_NW(S)                                                                                                                     \
    if (chunk == 4) hipLaunchKernelGGL((osc_noise_walk_kernel<S, 4>), grid, dim3(256), 0, st, V, N, d_rand, d_outhold, d_out);          \
    else if (chunk == 32) hipLaunchKernelGGL((osc_noise_walk_kernel<S, 32>), grid, dim3(256), 0, st, V, N, d_rand, d_outhold, d_out);   \
    else if (chunk == 16) hipLaunchKernelGGL((osc_noise_walk_kernel<S, 16>), grid, dim3(256), 0, st, V, N, d_rand, d_outhold, d_out);   \
    else hipLaunchKernelGGL((osc_noise_walk_kernel<S, 8>), grid, dim3(256), 0, st, V, N, d_rand, d_outhold, d_out);
        if (flavour == 2) { MXG_NW(2) } else if (flavour == 3) { MXG_NW(1) } else { MXG_NW(0) }
#undef MXG_NW
        return check_hip(hipGetLastError(), "osc_noise_walk_kernel launch");
    }
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(osc_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, count, V, N, d_rand, d_outhold, d_out);
    return check_hip(hipGetLastError(), "osc_noise_kernel launch");
}
