// osctab.hip -- EXTENSION (labelled as such everywhere): maxiOsc::sinebuf over PER-VOICE wavetables.
//
// The reference has ONE shared 514-entry table (`double sineBuffer[514]`, src/maximilian.cpp:63) that sinebuf interpolates
// (C:266-274); with it the voice-bank render has no read stream to speak of (the table lives in LDS, osc.hip).  SURVEY.md 8(d) row 2
// and north_star name the variant in which every voice owns its table -- [V][514] doubles, 4112 B read per voice and block = 8.03 B per
// sample at 512-sample blocks -- as the HBM-READ roofline of the wavetable path.  Same arithmetic, same expression order, per voice
//     phase += 512./(sr/(freq*chandiv)); if (phase >= 511) phase -= 512; rem = phase - floor(phase);
//     out = (1-rem)*T_v[1+(long)phase] + rem*T_v[2+(long)phase]
// so a bank whose tables all equal sineBuffer gives mxg_osc_render(sinebuf)'s bits, i.e. the reference's (tests/test_gpu_osctab.py).
//
// A table must be on chip for the whole block to be read ONCE, and 160 KB of LDS hold 38 of them: one lane per voice would leave
// a CU with 38 busy lanes.  So a block is cut along TIME as well: 16 lanes per voice, lane (u, t) renders samples [t PL, (t+1) PL) of
// voice u (PL = 16 or 32), a workgroup of 256 lanes = 16 voices per ROUND.  The phase at the start of each part comes from a small
// first kernel (lanes = voices, the recurrence without its output -- the same additions in the same order, the same bits -- leaving 16
// marks per voice: 3 % of the table traffic, written and read once).  Rounds are double-buffered: while round r is rendered from one
// half of the LDS, the 65 792 bytes of round r + 1's tables stream into the other half with global_load_lds_dwordx4 (LDS-DMA: no
// registers, 1 KiB per wave instruction; the source of a round is ONE contiguous piece of the table array), one barrier per round.
// A persistent grid of one workgroup per CU walks contiguous ranges of voice groups.
// Output: the per-voice block is optional (out[n][v]; rows of 128 contiguous bytes per store: a convenience, not a fast path); the
// measured form is the fused maxiMix::stereo mixdown (C:503-509 + the user's sum over voices): the 16 voices of a round sit in the 16
// lanes of a DPP row, a transposing butterfly (mxg_lanefold.h) turns 16 samples x 16 voices into 16 sums, and every lane keeps the
// running sum of ITS sample over all rounds in registers; the per-workgroup rows [workgroup][N][2] are added by mix_partials_kernel.
#include "mxg_common.h"
#include "mxg_lanefold.h"
#include "mxg_osc.h"

namespace mxg {
namespace {

constexpr int kTabLen = 514;         // doubles per voice (sineBuffer[514], C:63)
constexpr int kTabParts = 16;        // lanes per voice = time parts per block
constexpr int kTabVoices = 16;       // voices per round
constexpr int kTabRound = kTabVoices * kTabLen;  // doubles per LDS buffer (65 792 B)

// pass 1: the phase of every voice at the start of each time part, and after the block
template <int PL>
__global__ void osctab_marks_kernel(size_t V, size_t N, const double *__restrict__ freq, double *__restrict__ phase_io,
                                    double *__restrict__ marks, double sr) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double ph = phase_io[v];
    const double inc = 512. / (sr / (freq[v] * kChandiv));  // C:269
    const int whole = (int)(N / PL);  // parts that lie inside the block completely
#pragma unroll 1
    for (int t = 0; t < kTabParts; t++) {
        marks[(size_t)t * V + v] = ph;
        if (t < whole) {
#pragma unroll
            for (int k = 0; k < PL; k++) {
                ph += inc;
                if (ph >= 511) ph -= 512;  // C:270
            }
        } else {
            for (size_t n = (size_t)t * PL; n < N; n++) {
                ph += inc;
                if (ph >= 511) ph -= 512;
            }
        }
    }
    phase_io[v] = ph;
}

// the tables of voice group g -> one LDS buffer (16 x 4112 B, or less for the bank's last group): 16-byte pieces, lane-linear
__device__ __forceinline__ void tables_issue(const double *__restrict__ tables, size_t g, size_t V, double *buf) {
    const size_t first = g * kTabVoices;
    const size_t nv = V - first < (size_t)kTabVoices ? V - first : (size_t)kTabVoices;
    const unsigned bytes = (unsigned)(nv * kTabLen * sizeof(double));
    const char *src = reinterpret_cast<const char *>(tables + first * kTabLen);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *dst = reinterpret_cast<char *>(buf) + wave * 1024;
#pragma unroll
    for (int i = 0; i <= 16; i++) {  // 17 x 4 KiB >= 65 792 B
        const unsigned off = (unsigned)(i * 256 + (int)threadIdx.x) * 16u;
        if (off < bytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                             (__attribute__((address_space(3))) void *)(dst + i * 4096), 16, 0, 0);
    }
}

// One lane = (voice u of the round, time part t).  The per-voice values of round g + 1 (mark, frequency, pan) are requested while
// round g is rendered; inside a round the table reads of a whole 16-sample chunk are in flight at once and the second chunk's drain
// under the first chunk's fold (osc_pipe_*, mxg_osc.h: at one wavefront per SIMD nothing else hides an LDS or a memory latency).
template <int CH, bool STORE, bool MIX>
__global__ __launch_bounds__(256) void osctab_kernel(size_t V, size_t N, const double *__restrict__ freq,
                                                     const double *__restrict__ tables, const double *__restrict__ marks,
                                                     double *__restrict__ hold_io, double *__restrict__ out,
                                                     const double *__restrict__ pan, double *__restrict__ rows, double sr) {
    __shared__ __attribute__((aligned(16))) double s_tab[2 * kTabRound];
    const int lane = threadIdx.x & 63;
    const int u = threadIdx.x & 15, t = threadIdx.x >> 4;
    constexpr int PL = CH * kMixChunk;
    const size_t groups = (V + kTabVoices - 1) / kTabVoices;
    const size_t per = (groups + gridDim.x - 1) / gridDim.x;
    const size_t g0 = (size_t)blockIdx.x * per;
    const size_t g1 = g0 + per < groups ? g0 + per : groups;
    double acc[CH][2];
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c][0] = acc[c][1] = 0.0;
    // the sample N - 1 (the member `output` after the block) belongs to part t_last, chunk c_last, position i_last
    const int t_last = (int)((N - 1) / PL), c_last = (int)(((N - 1) % PL) / kMixChunk), i_last = (int)((N - 1) % kMixChunk);
    auto voice_of = [&](size_t g) {  // a surplus lane shadows the bank's last voice (a table that IS in the buffer), gain 0
        const size_t vraw = g * kTabVoices + u;
        return vraw < V ? vraw : V - 1;
    };
    double n_mark = 0.0, n_freq = 1.0, n_pan = 0.0;
    if (g0 < g1) {
        tables_issue(tables, g0, V, s_tab);
        const size_t v = voice_of(g0);
        n_mark = marks[(size_t)t * V + v];
        n_freq = freq[v];
        if constexpr (MIX) n_pan = pan[v];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (size_t g = g0; g < g1; g++) {
        const int b = (int)((g - g0) & 1);
        double ph = n_mark;
        const double inc = 512. / (sr / (n_freq * kChandiv));  // C:269
        double x = n_pan;
        if (g + 1 < g1) {
            tables_issue(tables, g + 1, V, s_tab + (b ^ 1) * kTabRound);
            const size_t vn = voice_of(g + 1);
            n_mark = marks[(size_t)t * V + vn];
            n_freq = freq[vn];
            if constexpr (MIX) n_pan = pan[vn];
        }
        const size_t first = g * kTabVoices;
        const bool live = first + u < V;
        const size_t v = voice_of(g);
        const double *T = s_tab + b * kTabRound + (v - first) * kTabLen - 1;  // T[i + 1] == table[i]: the layout osc_pipe_* index
        double gl = 0.0, gr = 0.0;
        if constexpr (MIX) {
            if (x > 1) x = 1;  // C:504
            if (x < 0) x = 0;  // C:505
            gl = live ? sqrt(1.0 - x) : 0.0;  // two[0] = input*sqrt(1.0-x)   C:506
            gr = live ? sqrt(x) : 0.0;        // two[1] = input*sqrt(x)       C:507
        }
        OscPre q;
        q.inc = inc; q.k = 0.0; q.p1 = 0.0; q.p2 = 0.0;
        double hd = 0.0;
        OscPipe<kMixChunk> P[CH];
        osc_pipe_phase<MXG_OSC_SINEBUF, kMixChunk>(ph, q, P[0]);
        __builtin_amdgcn_sched_barrier(0);
        osc_pipe_fetch<MXG_OSC_SINEBUF, kMixChunk>(P[0], T);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CH == 2) {
            osc_pipe_phase<MXG_OSC_SINEBUF, kMixChunk>(ph, q, P[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            double r[kMixChunk];
            osc_pipe_finish<MXG_OSC_SINEBUF, kMixChunk>(P[c], r, hd);
            __builtin_amdgcn_sched_barrier(0);
            if (CH == 2 && c == 0) {
                osc_pipe_fetch<MXG_OSC_SINEBUF, kMixChunk>(P[1], T);
                __builtin_amdgcn_sched_barrier(0);
            }
            const size_t nb = (size_t)t * PL + (size_t)c * kMixChunk;
            if constexpr (STORE) {
                if (live) {
#pragma unroll
                    for (int i = 0; i < kMixChunk; i++)
                        if (nb + i < N) out[(nb + i) * V + v] = r[i];
                }
            }
            if (t == t_last && c == c_last && live) {  // (one row of lanes per round)
                double h = r[0];
#pragma unroll
                for (int i = 1; i < kMixChunk; i++) h = i == i_last ? r[i] : h;
                hold_io[v] = h;
            }
            if constexpr (MIX) {  // (samples at or beyond N are summed too and never written out)
                double L[kMixChunk], R[kMixChunk];
#pragma unroll
                for (int i = 0; i < kMixChunk; i++) {
                    L[i] = r[i] * gl;
                    R[i] = r[i] * gr;
                }
                acc[c][0] += fold_chunk<double>(L, lane);
                acc[c][1] += fold_chunk<double>(R, lane);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next round have landed ...
        __syncthreads();                                   // ... and everybody's; and everybody is done with this round's buffer
    }
    if constexpr (MIX) {
        int idx[kMixChunk];
#pragma unroll
        for (int i = 0; i < kMixChunk; i++) idx[i] = i;
        const int slot = fold_chunk<int>(idx, lane);  // which sample of a chunk this lane's sums belong to
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const size_t n = (size_t)t * PL + (size_t)c * kMixChunk + slot;
            if (slot >= 0 && n < N) {
                double2v pr = {acc[c][0], acc[c][1]};
                *reinterpret_cast<double2v *>(rows + ((size_t)blockIdx.x * N + n) * 2) = pr;
            }
        }
    }
}

typedef void (*osctab_fn)(size_t, size_t, const double *, const double *, const double *, double *, double *, const double *, double *,
                          double);
osctab_fn pick_tab(int ch, bool store, bool mix) {
    if (ch == 1) return store ? (mix ? osctab_kernel<1, true, true> : osctab_kernel<1, true, false>) : osctab_kernel<1, false, true>;
    return store ? (mix ? osctab_kernel<2, true, true> : osctab_kernel<2, true, false>) : osctab_kernel<2, false, true>;
}

}  // namespace
}  // namespace mxg

using namespace mxg;

extern "C" size_t mxg_osc_tables_groups(size_t V) {
    if (ensure_init_only()) return 0;
    const size_t groups = (V + kTabVoices - 1) / kTabVoices;
    const size_t cus = (size_t)device_cus();
    return groups < cus ? groups : cus;
}

extern "C" int mxg_osc_render_tables(size_t V, size_t N, const double *d_freq, const double *d_tables, double *d_phase,
                                     double *d_outhold, double *d_out, const double *d_pan, double *d_rows, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_freq && d_tables && d_phase && d_outhold, "null device pointer");
    MXG_REQUIRE(d_out || d_pan, "nothing to produce: give d_out (the per-voice block), d_pan + d_rows (the mixdown), or both");
    MXG_REQUIRE(!d_pan || d_rows, "the mixdown needs d_rows");
    MXG_REQUIRE(N <= (size_t)kTabParts * 2 * kMixChunk, "blocks of at most 512 samples (16 lanes per voice x 32 samples)");
    MXG_REQUIRE(!(((uintptr_t)d_tables) & 15), "d_tables must be 16-byte aligned");
    if (V == 0 || N == 0) return MXG_OK;
    hipStream_t st = resolve_stream(stream);
    const int ch = N <= (size_t)kTabParts * kMixChunk ? 1 : 2;
    double *marks = nullptr;  // [16][V] per-stream scratch
    if (int s = scratch_get(SCR_OSCTAB_MARKS, st, sizeof(double) * kTabParts * V, (void **)&marks)) return s;
    {
        KernelTimer kt("osctab_marks_kernel", st);
        if (ch == 1)
            hipLaunchKernelGGL(osctab_marks_kernel<16>, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, N, d_freq, d_phase, marks,
                               (double)settings().sampleRate);
        else
            hipLaunchKernelGGL(osctab_marks_kernel<32>, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, N, d_freq, d_phase, marks,
                               (double)settings().sampleRate);
    }
    const size_t grid = mxg_osc_tables_groups(V);
    KernelTimer kt("osctab_kernel", st);
    hipLaunchKernelGGL(pick_tab(ch, d_out != nullptr, d_pan != nullptr), dim3((unsigned)grid), dim3(256), 0, st, V, N, d_freq, d_tables,
                       marks, d_outhold, d_out, d_pan, d_rows, (double)settings().sampleRate);
    return check_hip(hipGetLastError(), "osctab_kernel launch");
}
