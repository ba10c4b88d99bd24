// mxg_pace.h -- the paced store schedule of the store-bound bank kernels (K2f, K1 beyond 65 536 voices, maxiSample::play; K1's headline
// size through a trial, PaceTrial below; round 6, profiles/r06_pace.md).
//
// Where a kernel's bound is its store stream -- 65 536 voices: one wavefront on every SIMD of the chip, each storing a 512-byte piece of
// every row -- it is FASTEST when no wavefront ever meets a full store queue.  Left to run, the wavefronts reach the memory system's
// back-pressure and the stream's efficiency collapses by a fifth (K2f: 51 us); started on a schedule -- chunk k (8 samples) not before
// t0 + k P ticks of the device's constant counter (s_memrealtime: 100 MHz on MI355X, 10 ns a tick; the host takes the rate from
// hipDeviceAttributeWallClockRate), P a hair above the time the memory system needs for the chip's 8 rows --
// the same instruction stream takes 41 us.  The optimum is a knee: a tick (10 ns) below it the collapse is back, above it the time is 64 P.
//
// So P is CONTROLLED, per stream and kernel form, by eight words in device scratch (zeroed once by the host):
//   [0] P (kPaceGaveUp: the schedule was given up on this stream)   [1] launches in the window | late ones among them << 8 |
//   the descent's stage << 16 | late launches in a row << 24
//   [4:5] one 64-bit accumulator of the launch: reporters finished | reporters whose chunks were the cheap ones << 16 | the sum of their
//   latenesses << 32 (ticks behind schedule at the last chunk, each capped at 8 P)   [7] the last launch's mean lateness (diagnostics)
// One workgroup in sixteen reports: its first wavefront adds into [4:5]; the LAST reporter to finish judges the launch.  Only a
// STORE-BOUND launch counts -- one whose chunks were (7 in 8) the cheap ones, far shorter than the period: a launch of expensive chunks
// (the per-sample state machine, mode B's coefficients per sample) is behind from its first chunk, the schedule never binds and its
// lateness says nothing about the memory system.  It was LATE if it ended, on average, more than half a period behind.  Near the knee
// late launches come at a RATE that falls with P (measured: every other launch a tick below the knee, one in ten on it, one in fifty
// three ticks above), and a tick costs 0.6 us per launch where a late launch costs ~6: the period worth having is the one with 5-10 %
// late launches.  So: from the starting period (a rate every box takes) P comes down two ticks per launch until the first late one,
// then a tick per launch until the next (a dozen launches in all); then windows of 32 launches -- the fourth late launch of a window puts P up a tick at once, a window with at most one takes a tick off, anything
// between holds (the TOLERANT rule, for K1's sinebuf whose launches go late now and then at any period: eight and four).  P follows the
// box, its clocks and the other streams of the moment; a period half as long again as the starting one means the schedule does not
// describe the launch at all (a grid that is not resident at once, a device of another shape) and gives it up for good.  Timing only: the
// bits do not depend on it.
#pragma once
#include "mxg_common.h"

namespace mxg {
namespace {

constexpr int kPaceWords = 8;
constexpr unsigned kPaceWindow = 32, kPaceLatesUp = 4, kPaceLatesDown = 1;
constexpr unsigned kPaceGaveUp = 0xffffffffu;  // ctl[0]: the schedule was given up on this stream (see Pace::finish)

struct Pace {
    unsigned P, t0, k, late, cheap;
    __device__ __forceinline__ static unsigned now() { return (unsigned)__builtin_amdgcn_s_memrealtime(); }
    // ctl: the controller's words or null; arg: the starting / fixed period (0 = not paced)
    __device__ __forceinline__ void start(const unsigned *ctl, unsigned arg) {
        P = arg;
        if (ctl) {
            const unsigned p = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p == kPaceGaveUp) P = 0;                // (the controller gave the schedule up on this stream: free-running, see finish)
            else if (p) P = p < 2 * arg ? p : 2 * arg;  // (whatever the words hold, a launch waits at most twice its starting schedule)
        }
        k = late = cheap = 0;
        t0 = P ? now() : 0;
    }
    // the top of a chunk: wait for its slot (is_cheap: this chunk's arithmetic is far shorter than any period worth having)
    __device__ __forceinline__ void wait(bool is_cheap) {
        if (P) {
            const unsigned due = t0 + k * P;
            int d = (int)(now() - due);
            while (d < 0) {
                __builtin_amdgcn_s_sleep(1);
                d = (int)(now() - due);
            }
            late = (unsigned)d;
            k++;
            cheap += is_cheap ? 1u : 0u;
        }
    }
    // the end of the launch, one lane per workgroup: fold, and the last workgroup updates the controller
    // wg: this workgroup's number.  One workgroup in sixteen reports (256 workgroups' atomics on one line would queue for ~10 us).
    // tolerant: a kernel whose launches go late now and then at ANY period (K1's sinebuf) -- the windows then take eight late launches
    // for a tick up and up to four for a tick down, and the descent ends with two late launches in a row
    __device__ __forceinline__ void finish(unsigned *ctl, unsigned arg, unsigned wg, unsigned nwg, bool tolerant = false) const {
        if (!ctl || !P || (wg & 15u)) return;
        const unsigned nrep = (nwg + 15u) / 16u;
        // ONE relaxed device-scope 64-bit atomic per reporter, no fence (an agent-scope fence would write the whole L2 back, +15 us per
        // launch): tickets in bits 0-15, early waits in 16-31, the latenesses' sum above; the reporter that draws the last ticket has
        // every other one's contribution in the value that comes back
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(ctl + 4);
        const unsigned long long mine = ((unsigned long long)(late < 8 * P ? late : 8 * P) << 32) | ((unsigned long long)(k && cheap * 8 >= k * 7 ? 1u : 0u) << 16) | 1ull;
        const unsigned long long all = atomicAdd(acc, mine) + mine;
        if ((unsigned)(all & 0xffffu) != nrep) return;
        atomicExch(acc, 0ull);
        const unsigned worst = (unsigned)(all >> 32) / nrep, bound = (unsigned)((all >> 16) & 0xffffu);  // (the MEAN lateness: a single straggler is not a collapse)
        const unsigned c1 = atomicAdd(&ctl[1], 0u);
        unsigned p = P, w = c1 & 0xff, lates = (c1 >> 8) & 0xff, booted = (c1 >> 16) & 0xff, strikes = c1 >> 24;
        const unsigned up = tolerant ? 2 * kPaceLatesUp : kPaceLatesUp, down = tolerant ? 4 * kPaceLatesDown : kPaceLatesDown;
        if (bound * 8 >= nrep * 7) {  // a store-bound launch: nearly every reporter's chunks were the cheap ones
            const bool is_late = worst > p / 2;
            if (booted != 1) {
                // the descent from the starting period (a safe one: the memory system takes it on every box seen): two ticks per launch
                // on schedule until the first late one (booted 0 -> 2), then a tick per launch until the next late one (the tolerant
                // rule: until two in a row), which puts P back up a tick and hands over to the windows (booted 1)
                strikes = is_late ? strikes + 1 : 0;
                if (booted == 0) {
                    if (is_late) {
                        booted = 2;
                        strikes = 0;
                    } else if (p > arg - arg / 4 + 1) {
                        p -= 2;
                    }
                } else if (strikes >= (tolerant ? 2u : 1u)) {
                    p++;
                    booted = 1;
                } else if (!is_late && p > arg - arg / 4) {
                    p--;
                }
            } else {
                lates += is_late ? 1 : 0;
                ++w;
                if (lates >= up) {  // at once: below the knee every other launch is late
                    p++;
                    w = lates = 0;
                } else if (w >= kPaceWindow) {
                    if (lates <= down && p > arg - arg / 4) p--;
                    w = lates = 0;
                }
            }
        }
        // the safety net: a period half as long again as the starting one (itself ~10 % above every knee measured) means that the schedule
        // does not describe this launch -- a grid that is not resident at once, a device of another shape, a neighbour that never lets
        // go: the kernel would run at the pace of a schedule nobody needs.  The stream then goes back to the free-running kernel for good.
        if (p > arg + arg / 2) p = kPaceGaveUp;
        atomicExch(&ctl[0], p);
        atomicExch(&ctl[1], w | (lates << 8) | (booted << 16) | (strikes << 24));
        atomicExch(&ctl[7], worst);
    }
};

// ---- free-running or paced, and at which period?  A trial on the device -------------------------------------------------------------
// K1's sinebuf at 65 536 voices (the headline) has a free-running stream -- pair rows, write-through -- that sits ON the knee on some boxes
// (40.4-41.3 us) and collapses on others (43.5-47.6 us, box by box and sometimes run by run); paced 8-byte stores on the right period take
// 40.7 us on the first kind and 42.5 on the worst of the second -- but sinebuf is the waveform whose launches go late now and then at any
// period, and the lateness controller does not find that period quickly.  So the launches' WALL TIME decides, on the device: sixteen
// words (two slots) --
//   [0] the verdict (0 none yet, 1 free-running, >= 2 paced on that fixed period)   [1] launches so far | the block length's low 16 bits << 16
//   [2:3] this launch's accumulator (reporters: the last one to finish acts)   [4] what the free-running phase took, in ticks
//   [5] / [6] the best period's phase and the period   [8] the clock at the end of the last phase
// -- and ONE kernel that holds both loops.  Phases of 32 launches: free-running (clocks and caches settle), free-running (measured:
// the device clock from the end of the previous phase's last launch to the end of its own -- the wall time, gaps included), then the
// periods 31/32, 29/32, 27/32 of the starting one, DESCENDING towards the knee (measured at 0.83-0.9 of it), for as long as each phase
// is shorter than the one before: below the knee a phase is 10-30 % longer, and the descent stops there.  The best period wins if it
// beats the free-running phase by 2 %.  Where the free-running stream is the best (most boxes) the trial costs two phases a few
// per cent slower than it: 0.1 % of a 2000-launch run.  After 16 384 launches it is repeated; a run shorter than 96 launches never
// leaves the free-running kernel it always had.
constexpr unsigned kTrialPhase = 32, kTrialCandidates = 3, kTrialAgain = 16384 + 5 * kTrialPhase;  // (< 2^16: the count shares its word)

struct PaceTrial {
    unsigned t_begin, n, period, tag;  // period: this launch's (0 = free-running); tag: the block length's low 16 bits (wall times of
                                       // different block lengths do not compare: another length starts the trial again)
    __device__ __forceinline__ static unsigned candidate(unsigned arg, unsigned i) { return arg * (31u - 2u * i) / 32u; }  // i = 0, 1, 2: descending
    __device__ __forceinline__ void start(const unsigned *T, unsigned arg, unsigned block_len) {
        t_begin = Pace::now();
        n = 0;
        period = 0;
        tag = block_len & 0xffffu;
        if (T) {
            unsigned d = __hip_atomic_load(T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned w1 = __hip_atomic_load(T + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n = w1 & 0xffffu;
            if ((w1 >> 16) != tag) {  // (another block length, or the first launch: from the start)
                n = 0;
                d = 0;
            }
            if (d) {
                period = d >= 2 ? (d < 2 * arg ? d : 2 * arg) : 0;
            } else {
                const unsigned ph = n / kTrialPhase;  // 0, 1 free-running; 2 ... the candidates
                period = (ph >= 2 && ph < 2 + kTrialCandidates) ? candidate(arg, ph - 2) : 0;
            }
        }
    }
    // the end of the launch, one lane per workgroup
    __device__ __forceinline__ void finish(unsigned *T, unsigned arg, unsigned wg, unsigned nwg) const {
        if (!T || (wg & 15u)) return;
        const unsigned nrep = (nwg + 15u) / 16u;
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(T + 2);
        const unsigned long long all = atomicAdd(acc, 1ull) + 1ull;  // (relaxed, one line, no fence: see Pace::finish)
        if ((unsigned)(all & 0xffffu) != nrep) return;
        atomicExch(acc, 0ull);
        unsigned d = atomicAdd(&T[0], 0u), nn = n + 1;
        if ((atomicAdd(&T[1], 0u) >> 16) != tag) d = 0;  // (see start)
        if (!d) {
            if (nn % kTrialPhase == 0) {  // a phase ends with this launch
                const unsigned t_now = Pace::now(), ph = n / kTrialPhase, wall = t_now - atomicAdd(&T[8], 0u);
                atomicExch(&T[8], t_now);
                if (ph == 1) {
                    atomicExch(&T[4], wall);
                    atomicExch(&T[5], 0xffffffffu);
                    atomicExch(&T[6], 0u);
                } else if (ph >= 2) {
                    unsigned best = atomicAdd(&T[5], 0u), best_p = atomicAdd(&T[6], 0u);
                    bool done = ph + 1 == 2 + kTrialCandidates;
                    if (wall < best) {
                        best = wall;
                        best_p = candidate(arg, ph - 2);
                        atomicExch(&T[5], best);
                        atomicExch(&T[6], best_p);
                    } else {
                        done = true;  // longer than the period before it: that was the knee
                    }
                    if (done) d = ((unsigned long long)best * 100 < (unsigned long long)atomicAdd(&T[4], 0u) * 98) ? best_p : 1u;
                }
            }
        } else if (nn >= kTrialAgain) {
            d = 0;
            nn = 0;
        }
        atomicExch(&T[0], d);
        atomicExch(&T[1], nn | (tag << 16));
    }
};

}  // namespace
}  // namespace mxg
