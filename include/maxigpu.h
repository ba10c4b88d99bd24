/* include/maxigpu.h -- the C-ABI of libmaxigpu.so: the MI355X (gfx950) voice-bank DSP path
 * behind the Maximilian class API.
 *
 * The reference (micknoise/Maximilian, cited as C = src/maximilian.cpp, H = src/maximilian.h,
 * L/ = src/libs/) has no FFI: its "plugin API" is the C++ class surface a user's
 * `void play(double*)` calls once per sample (cpp/commandline/player.cpp:25-44 drives it).
 * Each entry point below is what a binding for that surface would bind: it renders N samples
 * of a BANK of V independent instances of one reference class in a single launch.  The host
 * facade (include/maximilian_bank.hpp) keeps the reference's class/method names on top.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross this boundary.
 *   - every `d_*` pointer is DEVICE memory (from mxg_malloc, hipMalloc, or a torch tensor's
 *     data_ptr()); `h_*` is host memory.  Pointers are borrowed for the call only.
 *   - bank signals are sample-major / voice-minor:  out[n*V + v],  n < N, v < V.
 *   - state is SoA: one array of length V per reference member, updated in place so
 *     consecutive calls continue the stream exactly like consecutive play() calls.
 *   - `stream` is a hipStream_t passed as void* (NULL = the library's own default stream; to
 *     target HIP's null stream pass hipStreamLegacy, i.e. (void*)1).
 *     Calls are asynchronous on that stream; mxg_sync()/mxg_stream_sync() wait.
 *   - returns 0 on success, a negative mxg_status otherwise; never throws, never exit()s
 *     (the reference exit(1)s on a bad FFT size, L/fft.cpp:129-132; here that is
 *     MXG_ERR_INVALID).  mxg_last_error() gives a thread-local message.
 *   - there is NO CPU fallback: without a HIP device every compute call fails with
 *     MXG_ERR_NO_DEVICE.
 */
#ifndef MAXIGPU_H
#define MAXIGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MXG_OK = 0,
    MXG_ERR_INVALID = -1,   /* bad argument (null pointer, unknown waveform, odd size ...) */
    MXG_ERR_NO_DEVICE = -2, /* no HIP device / runtime failure at init */
    MXG_ERR_HIP = -3,       /* a HIP call failed; see mxg_last_error() */
    MXG_ERR_NOMEM = -4
} mxg_status;

/* maxiOsc waveforms, numbered as in oracle/maxi_oracle.c.  Reference: C:228-373. */
typedef enum {
    MXG_OSC_SINEWAVE = 0,       /* C:228-235 */
    MXG_OSC_COSWAVE = 1,        /* C:276-283 */
    MXG_OSC_PHASOR = 2,         /* C:285-291 */
    MXG_OSC_SAW = 3,            /* C:333-340 */
    MXG_OSC_TRIANGLE = 4,       /* C:362-373 */
    MXG_OSC_SQUARE = 5,         /* C:293-300 */
    MXG_OSC_PULSE = 6,          /* C:302-311  p1 = duty */
    MXG_OSC_IMPULSE = 7,        /* C:312-319 */
    MXG_OSC_SINEBUF = 8,        /* C:266-274 */
    MXG_OSC_SINEBUF4 = 9,       /* C:237-264 */
    MXG_OSC_SAWN = 10,          /* C:342-359 */
    MXG_OSC_PHASORBETWEEN = 11  /* C:321-330  p1 = startphase, p2 = endphase */
} mxg_osc_waveform;

/* maxiFilter kinds.  Reference: C:442-500. */
typedef enum {
    MXG_FLT_LORES = 0,   /* C:455-468 */
    MXG_FLT_HIRES = 1,   /* C:471-484 */
    MXG_FLT_BANDPASS = 2,/* C:487-500 */
    MXG_FLT_LOPASS = 3,  /* C:442-446 */
    MXG_FLT_HIPASS = 4   /* C:449-453 */
} mxg_filter_kind;

/* ---- library / device ---------------------------------------------------------------- */
/* Select the HIP device (-1 = current), create the default stream.  Idempotent. */
int mxg_init(int device);
const char *mxg_last_error(void);
const char *mxg_version(void);
/* maxiSettings::setup (H:138-143): global sampleRate / channels / bufferSize (size_t). */
int mxg_settings(size_t sampleRate, size_t channels, size_t bufferSize);
size_t mxg_sample_rate(void);

/* ---- device memory / streams (thin pass-throughs, so a host needs no HIP headers) ----- */
void *mxg_malloc(size_t bytes);
int mxg_free(void *d_ptr);
int mxg_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int mxg_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int mxg_memset(void *d_dst, int value, size_t bytes, void *stream);
/* Asynchronous forms: the copy is only enqueued on `stream`; the host buffer must stay valid (and for a real overlap be
 * pinned: mxg_host_alloc) until the stream has passed it -- mxg_event_record + mxg_event_sync / mxg_event_query.  Together
 * with the render entry points (all of which only enqueue) this is the asynchronous double-buffering of SURVEY 8(b):
 * render block k+1 and its download on a stream while the audio thread still serves block k.  A bank's state is the
 * caller's own device SoA arrays, so a "state get/set" is a copy of those arrays (mxg_memcpy_d2h / _h2d). */
int mxg_memcpy_h2d_async(void *d_dst, const void *h_src, size_t bytes, void *stream);
int mxg_memcpy_d2h_async(void *h_dst, const void *d_src, size_t bytes, void *stream);
int mxg_memcpy_d2d_async(void *d_dst, const void *d_src, size_t bytes, void *stream);  /* device to device, in stream order */
void *mxg_host_alloc(size_t bytes); /* pinned host memory */
int mxg_host_free(void *h_ptr);
void *mxg_stream_create(void);
int mxg_stream_destroy(void *stream);
int mxg_stream_sync(void *stream);
int mxg_sync(void);
/* HIP events on `stream`, for timing a launch from the host side of the boundary. */
void *mxg_event_create(void);
int mxg_event_destroy(void *event);
int mxg_event_record(void *event, void *stream);
int mxg_event_elapsed_ms(void *start, void *stop, float *h_ms);
int mxg_event_sync(void *event);
int mxg_event_query(void *event); /* 1 = complete, 0 = not yet, < 0 error */
int mxg_stream_wait_event(void *stream, void *event);

/* The headless host: cpp/commandline/player.cpp:25-44 `routing()` restated without RtAudio -- calls the user's
 * `void play(double *output)` (src/maximilian.cpp:205-207) once per frame and copies `channels` doubles per frame into
 * the interleaved buffer.  h_lastValues [channels] persists between calls like the callback's userData. */
int mxg_host_render(void (*play)(double *), size_t channels, size_t nFrames, double *h_interleaved, double *h_lastValues);

/* ---- per-kernel timing (measurement only) --------------------------------------------------------- */
/* mxg_prof_enable(1): every entry point brackets its kernels with HIP events on the launch stream (one label per
 * kernel, e.g. "osc_kernel", "fft1024_kernel"); mxg_prof_read(i, ...) waits for label i's events and returns the
 * summed kernel time and the number of launches since the last mxg_prof_reset().  Off by default (no events, no
 * overhead).  This is how bench.py measures the average launch duration of the dominant kernel inside its timed
 * region; the rocprofv3 kernel-trace averages in profiles/ agree with it. */
int mxg_prof_enable(int on);  /* returns the previous setting */
int mxg_prof_reset(void);
int mxg_prof_count(void);
int mxg_prof_read(int index, const char **h_label, double *h_total_ms, size_t *h_launches);
/* Average interval of `pairs` EMPTY event pairs on `stream` (two records back to back, nothing between): the marker
 * overhead every figure of mxg_prof_read carries, for a caller that wants to subtract it. */
int mxg_prof_overhead_ms(void *stream, int pairs, double *h_ms);

/* ---- tuning knobs (performance only; results are identical for every setting) ---------- */
/* key: "osc_vpl" (voices per lane: 0 automatic, 1|2), "osc_block" (64..1024), "osc_nt" (non-temporal stores: 0 never, 1 always, 2 by the size of the output block),
 * "voice_block", "voice_nt" (as osc_nt), "mix_rows" (sample rows per workgroup of the mixdown, 1|2), "fft_generic",
 * "mfcc_tiled" (LDS-staged spectra 0|1), "grain_chunked" (time-sharded granular render 0|1), "grain_lanes_k",
 * "grain_unit" (coalesced unit-increment render 0|1), "grain_line" (tile render for arbitrary increments 0|1), "grain_fast_sched" (event-driven schedulers 0|1), "grain_streamed" (unit-path maxiTimeStretch call as ONE launch whose scheduler lanes
 * and tile renders run side by side, 1 default; 0 = time slices on the library's auxiliary streams), "grain_slices" (those time slices, 1..16), "osc_mix_win" (K1m: samples per workgroup combine window, 0 automatic, 128 or 256), "osc_mix_pcwin" (the same for the producer / consumer form: 0 automatic, 256 or 512), "osc_mix_pc" (K1m as producer / consumer wavefront pairs: 0 automatic, 1 off, 2 on),
 * "rw_store" (the read + write bank kernels' 16-byte pair-row streams: 0 automatic, 1 off, 2 / 3 / 4 on with plain / write-through /
 * non-temporal stores), "fft_exact" (1 default; 0 = TOLERANCE MODE of mxg_fft_mfcc_batch: the 512-point transform as true radix-8 butterflies with correctly
 * rounded twiddles and fused multiply-adds, hardware square root -- about a quarter fewer instructions; magnitudes within 6e-7 x the
 * frame's peak of the TRUE transform and -- because the reference's fp32 twiddle recurrences drift by ~1e-4 -- within 4e-4 of the reference's,
 * mfcc within 5e-4 of the reference's; bit-exactness is given up, accuracy is gained),
 * "time_parallel" (the other exception to "identical results": 1 lets banks of at most 4096 linear filters with block-constant
 * coefficients -- maxiBiquad, maxiSVF, maxiDCBlocker through mxg_filter2_render, lores / hires through mxg_filter_render -- and blocks
 * of 64 * {1..32} samples be cut along time and joined by a wavefront scan: a 6-voice x 512-sample block in a few microseconds
 * instead of 23-28, with reordered arithmetic: |error| <= 1e-10 x the block's peak (measured <= 5e-12); default 0 = the bit-exact kernels),
 * "osc_plan" (K1, banks beyond ~350 MB per block: rendered as passes of 98 304 voices + a remainder launch; 0 automatic, 1 never, 2 / 3 always),
 * "osc_passes" / "osc_mix_passes" (K1 / K1m: voice groups a wavefront renders one after the other, the grid covering 1 / passes of the bank: 0 automatic, 1..64),
 * "osc_split" (time parts per voice group in K1: 0 automatic, 1..8), "osc_mix_split" (the same for the fused render + mixdown K1m,
 * 0 automatic, 1..4), "osc_mix_store" (K1m's per-voice block: 0 automatic, 1 plain 8-byte stores, 2 pair rows of 16-byte stores),
 * "smp_split" (time parts of a block-constant playAtSpeed / playOnceAtSpeed / playUntilAtSpeed launch: 0 automatic, 1..8),
 * "smp_pipe" (that launch issues the window loads of chunk k+1 before the stores of chunk k, 0|1),
 * "ifft_stream" (mxg_ifft_batch: inverse transform and hop buffer in one kernel: 0 never, 1 where hop >= fftSize / 2, 2 wherever it fits),
 * "mfcc_mfma_fullk" (the MFMA mel contraction runs over all numBins bins instead of the ones that carry weight, 0|1).
 * "fused_layout" (mxg_fft_mfcc_batch: 0 automatic, 1 = two frames in flight per wavefront and two 4-wave workgroups per CU, 2 = one
 * frame in flight and one 12-wave workgroup per CU; same bits either way),
 * "fused_waves16" (mxg_fft_mfcc_batch: the 16-waves-per-CU form of the fused kernel when the request allows it, 0|1),
 * "fused_mel" (mxg_fft_mfcc_batch, the stage after the magnitudes: 0 automatic = 3, or 2 when d_melraw / d_melbands are requested; 1 = sparse mel walk, logs and DCT on the vector ALU in
 * the reference's summation orders; 2 = the same walk -- band sums bit-exact -- with the DCT's 42-term sums on the matrix pipe
 * (v_mfma_f64_4x4x4_4b_f64: fused multiply-adds, within 1e-13 x the largest band log of form 1); 3 = the mel contraction on the matrix
 * pipe as well, banded per quad of filters: band sums within 4e-15 x the frame's largest band (measured 2.2e-16), mfcc within 1e-13 of form 1
 * (measured 1.4e-15) -- the north star's
 * "MFMA for the mel-filterbank x frame contraction" inside the one-kernel path).
 * "osc_store" (K1's store stream: 0 automatic by waveform and bank size; one voice per lane: 1 plain 8-byte stores, 2 non-temporal,
 * 3 / 4 / 5 two samples of a lane pair exchanged into one 16-byte store per lane, plain / write-through (sc1) / non-temporal; two
 * voices per lane: 1 plain, 2 non-temporal, 3 write-through 16-byte stores), "voice_store" / "voice_xcd" (the same for the fused voice kernel; one voice
 * per lane only), "voice_mix_store" (the store stream of mxg_voice_render_mix*: 0 automatic = "voice_store"'s rule, 1 ... 5 its flavours),
 * "voice_diet" (the fused voice, mode A without mixdown: 0 automatic = the short instruction stream of the fast paths, except around 65 536 voices when the launch cannot be paced; 1 = the round-5 stream; 2 = the short one; same bits),
 * "voice_pace" (the fused voice's paced store schedule, a chunk of 8 samples every P ticks of 10 ns: 0 automatic = a per-stream controller
 * at the store-bound bank sizes whose whole grid is resident at once (45 056 ... 262 144 voices; mode B to 131 072, the mixdown form to 65 536); 1 never; >= 2 a fixed P; timing only, same bits), "osc_pace" (the same for mxg_osc_render: 0 automatic = the table-free waveforms at 90 112 ... 327 680 voices; 1 never; >= 2 a fixed P),
 * "tab_sides" (mxg_osc_render_tables*: 0 automatic = 1 workgroups of 256 lanes, one round of 8 voices at a time, two per CU; 2 = workgroups of 512 lanes, two rounds side by side),
 * "smp_pace" (the paced schedule for mxg_sample_render: 0 automatic = play() at 45 056 ... 229 375 voices; 1 never; >= 2 a fixed P), "smp_ring" (the *AtSpeed players' window rows as rings: 0 automatic = samples beyond 1 GiB, 1 off, 2 on), "grain_spin_limit" (see mxg_granular_retries), "osc_xcd" (workgroups renumbered so that each of the eight XCDs renders
 * one contiguous eighth of the bank: 0 automatic, 1 off, 2 on), "grain_sync" (mxg_granular_render reads its error word back before it returns, 0|1;
 * default 0: deferred, see mxg_last_async_error), "part_spin_limit" (polls a time-split kernel's writer part makes before it gives up
 * and reports through mxg_last_async_error), "part_fault" (test-only fault injection: that writer waits for a signal that never comes).
 * "rw_chunk" (samples per chunk of the pair-row maxiFilter kernel / rows per chunk of the noise column walk: 0 automatic = 8, 4 / 8 / 16 / 32;
 * measured: 8 is the optimum, fewer loads in flight starve the read stream, longer chunks burst the stores).
 * The tests flip every one of them and demand identical bits.  Returns the previous value or MXG_ERR_INVALID. */
int mxg_tune(const char *key, int value);

/* Asynchronous device errors.  A kernel that detects, while it runs, a failure no argument check could see -- a time-split
 * launch whose writer part never got its sibling parts' signals (its state is then NOT stored), a granular render that ran out
 * of rand() draws or met a live grain its plan could not have made -- stores a code in a word of pinned host memory.  No call
 * synchronises for it: the NEXT call of any entry point (and every synchronising call -- mxg_stream_sync, mxg_sync,
 * mxg_memcpy_d2h -- after its wait) returns the failure as its status with the message in mxg_last_error(), once, and clears
 * it.  mxg_last_async_error() is that check by itself: MXG_OK, or the pending failure.  (Render entry points therefore never
 * block the host, and their launch sequences can be captured into a hipGraph; a captured launch reports the same way.)
 * After a time-part time-out the caller must synchronise that stream and treat the state of every bank rendered on it since the failed
 * launch as invalid (launches already enqueued behind it may have run from stale part counters): re-upload or re-create those banks.
 * "rw_store" (mxg_tune) is documented with the render entry points that honour it (maxiFilter, maxiEnv, maxiDelayline, maxiSample,
 * maxiEnvGen, filter2: 0 automatic, 1 8-byte streams, 2 / 3 / 4 16-byte pair rows with plain / write-through / non-temporal stores). */
int mxg_last_async_error(void);
/* Streamed granular launches (mxg_granular_render*, one-launch form) whose tile renders gave up waiting for the scheduler workgroups of
 * their own launch -- a pre-empted or partitioned device -- and were therefore rendered AGAIN by the kernel that follows every such
 * launch (the renderer-alone form over the completed lists: the same bits, no silence, no error).  A statistic: 0 on a healthy device.
 * Counted when the call's last kernel has run.  Knob "grain_spin_limit" (polls without progress before a render gives up; 0 = 2^22). */
int mxg_granular_retries(void);

/* ---- maxiOsc bank -------------------------------------------------------------------- */
/* Renders out[n][v] = bank[v].<waveform>(freq) for n < N, exactly as N consecutive per-sample
 * calls would (H:169-215).  d_freq is [V] (fps=0, block-constant) or [N][V] (fps=1, audio-rate
 * modulation); fps=2: d_freq AND d_p1 are [N][V] (a pulse width / start phase that changes per
 * sample as well).  d_p1/d_p2: per-voice extra arguments (see mxg_osc_waveform), may be NULL when
 * unused.  d_phase / d_outhold: the members `phase` and `output` (H:173,176), in/out.
 * maxiOsc::noise (C:214-220) draws from the process-wide rand(): see mxg_osc_noise. */
int mxg_osc_render(int waveform, size_t V, size_t N, const double *d_freq, int fps,
                   const double *d_p1, const double *d_p2, double *d_phase, double *d_outhold,
                   double *d_out, void *stream);
/* The same with a row pitch: sample n of voice v goes to d_out[n * out_pitch_bytes / 8 + v] (out_pitch_bytes a multiple of 8, >= V * 8;
 * a multiple of 16 keeps the 16-byte store streams).  mxg_osc_render is this call with out_pitch_bytes = V * 8.  Why a caller
 * would pad: when V * 8 is a multiple of 2 MB (262 144 voices and its multiples) the same column of every row lands on the same HBM
 * channel and the store stream runs at 0.68-0.71 of the peak instead of 0.75-0.80; a pitch of V * 8 + 256 ... 4096 bytes removes
 * that (profiles/r05_osc_pitch.md).  src/maximilian.cpp:266-274 per voice, nothing else changes. */
int mxg_osc_render_pitch(int waveform, size_t V, size_t N, const double *d_freq, int fps,
                         const double *d_p1, const double *d_p2, double *d_phase, double *d_outhold,
                         double *d_out, size_t out_pitch_bytes, void *stream);

/* maxiOsc::noise (C:214-220): `float r = rand()/(float)RAND_MAX; output = r*2-1`.  rand() is one
 * serial process-wide stream (not a per-object state), so which draw a voice sees is decided by
 * the order of the caller's per-sample loop.  The caller therefore supplies the draws, exactly as
 * the granular entry point does for its rand() uses: d_rand[n][v] is the value rand() returned for
 * voice v at sample n (for a voice-inner loop: draw number n*V+v); the kernel applies the
 * reference's float arithmetic (bit-exact).  count = N*V values; d_outhold ([V], may be NULL)
 * receives the member `output` (H:176) after the last sample. */
int mxg_osc_noise(size_t V, size_t N, const int32_t *d_rand, double *d_outhold, double *d_out,
                  void *stream);

/* Render + fused stereo mixdown: as mxg_osc_render (block-constant frequencies) and, in the same
 * pass, d_mix[n][0..1] = sum_v (out[n][v]*sqrt(1-pan_v), out[n][v]*sqrt(pan_v)) (maxiMix::stereo,
 * C:503-509, plus the user-side sum over voices, 15.polysynth/main.cpp:67).  The per-voice block is
 * never re-read: each workgroup sums its 256 voices through an LDS transpose and a small second
 * kernel adds the per-workgroup rows (fixed order: deterministic; tolerance on the mix as for
 * mxg_mix_stereo).  d_out may be NULL: then only the mix is produced (what a play() callback needs)
 * and nothing but [N][2] leaves the chip. */
int mxg_osc_render_mix(int waveform, size_t V, size_t N, const double *d_freq, const double *d_p1,
                       const double *d_p2, double *d_phase, double *d_outhold, double *d_out,
                       const double *d_pan, double *d_mix, void *stream);
/* The same render WITHOUT the second kernel: d_rows[g][n][0..1], g < mxg_osc_mix_groups(V) = ceil(V / 256), holds the
 * mix of voices [256 g, 256 g + 256); the mix is their sum in ascending g.  For a caller that adds the rows elsewhere:
 * a grouped mix queue (mxg_mixq_create_grouped) does it once per batch on its own stream, so that the render stream
 * carries ONE kernel per block; mxg_mix_rows_sum(groups, count = N * 2, d_rows, d_mix) is that sum as a call. */
size_t mxg_osc_mix_groups(size_t V);
int mxg_osc_render_mix_rows(int waveform, size_t V, size_t N, const double *d_freq, const double *d_p1,
                            const double *d_p2, double *d_phase, double *d_outhold, double *d_out,
                            const double *d_pan, double *d_rows, void *stream);
int mxg_mix_rows_sum(size_t groups, size_t count, const double *d_rows, double *d_mix, void *stream);

/* EXTENSION (no reference counterpart; SURVEY.md 8(d) row 2, north_star "HBM-read roofline on the wavetable path"):
 * maxiOsc::sinebuf (C:266-274) with a table PER VOICE.  d_tables[V][514] (16-byte aligned) plays the part of the reference's one
 * shared `double sineBuffer[514]` (C:63) for voice v: out = (1-rem)*T_v[1+(long)phase] + rem*T_v[2+(long)phase], same phase
 * recurrence, same expression order -- a bank whose tables all hold sineBuffer gives mxg_osc_render(MXG_OSC_SINEBUF)'s bits.
 * 4112 B of table are read per voice and block (8.03 B per sample at N = 512), each exactly once.  N <= 512.
 * Outputs, either or both: d_out [N][V] (the per-voice block) and the fused maxiMix::stereo mixdown as partial rows
 * d_rows[g][n][0..1], g < mxg_osc_tables_groups(V) (d_pan [V] given): the mix is the sum of the rows in ascending g
 * (mxg_mix_rows_sum, or a grouped mix queue).  d_phase / d_outhold as for mxg_osc_render. */
size_t mxg_osc_tables_groups(size_t V);
int mxg_osc_render_tables(size_t V, size_t N, const double *d_freq, const double *d_tables, double *d_phase,
                          double *d_outhold, double *d_out, const double *d_pan, double *d_rows, void *stream);
/* The same render with two additions (round 6), each optional and neither changing a bit of the block, the rows or the carried phase:
 *   d_mix [N][2] (needs d_pan + d_rows): the sum of the rows, formed INSIDE the render kernel by the workgroups that finish last --
 *     mxg_mix_rows_sum's additions in its order, no second launch;
 *   flags & MXG_TABLES_AHEAD: pipelined blocks.  The phase recurrence of a block (512 dependent steps per voice) has to run before
 *     the block's tables are rendered; with this flag the render of block k also walks block k + 1's, beside its table traffic, so
 *     from the second call on a call is ONE kernel.  The caller's promise: the next call on this stream has the same V and N, the same
 *     d_freq / d_pan (pointers AND contents) and d_phase untouched in between.  A call that breaks the pattern (other arguments, no
 *     flag) is still correct as long as d_phase was not written: d_phase always holds the phase after the last rendered block.
 *     Banks of more than 512 voices per workgroup (131 072 on a 256-CU device) ignore the flag. */
#define MXG_TABLES_AHEAD 1
int mxg_osc_render_tables_ex(size_t V, size_t N, const double *d_freq, const double *d_tables, double *d_phase, double *d_outhold,
                             double *d_out, const double *d_pan, double *d_rows, double *d_mix, int flags, void *stream);

/* ---- maxiFilter bank ------------------------------------------------------------------ */
/* d_st = [5][V]: x, y, outputs[0], outputs[1], outputs[2] (H:289-302), in/out.
 * lores/hires/bandpass with block-constant cutoff/resonance (cps=rps=0): the coefficients that
 * need cos/pow/sqrt (C:459-461: c, r; C:492-495: inputs[0..2]) must be supplied precomputed in
 * d_coef = [3][V] -- computed on the host with the host libm by mxg_filter_coeffs_host(),
 * which makes the recurrence bit-exact.  With cps or rps = 1 (per-sample modulation, e.g.
 * 14.monosynth/main.cpp:53) the coefficients are evaluated on the device (cos/sqrt) under the
 * tolerance stated in DESIGN.md; d_coef is ignored. d_res/d_coef may be NULL for
 * lopass/hipass. */
int mxg_filter_render(int kind, size_t V, size_t N, const double *d_in, const double *d_cutoff,
                      int cps, const double *d_res, int rps, const double *d_coef, double *d_st,
                      double *d_out, void *stream);
/* lores / hires / bandpass with PER-SAMPLE coefficients computed by the caller: d_coef_ps = [N][3][V], rows as mxg_filter_coeffs_host
 * writes them (c, r, unused | inputs[0..2]) for the cutoff / resonance of that sample.  The bit-exact form of a modulated cutoff
 * (the reference evaluates cos / pow / sqrt with the host libm on every call, src/maximilian.cpp:456-461): mxg_filter_render's own
 * per-sample mode evaluates them on the device, under a tolerance.  include/maximilian.h uses this when a cutoff follows another
 * object's output. */
int mxg_filter_render_coefs(int kind, size_t V, size_t N, const double *d_in, const double *d_coef_ps, double *d_st, double *d_out,
                            void *stream);
/* Host-side coefficient evaluation on this machine's libm: kind lores/hires -> h_coef[0]=c,
 * h_coef[1]=r (C:456-461); bandpass -> h_coef[0..2]=inputs[0..2] (C:489-495).  h_coef is [3][V]. */
int mxg_filter_coeffs_host(int kind, size_t V, const double *h_cutoff, const double *h_res,
                           double *h_coef);

/* ---- maxiEnvGen bank (H:2268-2547) ------------------------------------------------------------------ */
/* maxiEnvGen::setup(levels, times, curves, ...) (H:2366-2399) on the host: times in ms, or -46692
 * (maxiEnvGen::HOLD) for the one allowed hold stage; h_stages [nlevels-1][6] = startlevel, endlevel,
 * gradient, curve, length (samples), hold.  Returns the number of stages (<= 32) or MXG_ERR_INVALID
 * where setup() returns false.  setupAR/ASR/ADSR (H:2480-2504) are particular level/time/curve lists. */
int mxg_envgen_stages_host(size_t nlevels, const double *h_levels, const double *h_times,
                           const double *h_curves, double *h_stages);
/* d_out[n][v] = bank[v].play(trigger) (H:2277-2354) for one shared envelope shape d_stages [nstages][6]
 * (device copy of the table above), loop / retrigger flags, and a trigger signal d_trig that is [N][V]
 * (tpv = 1) or one shared [N] gate (tpv = 0).  State, in/out: d_dst = [5][V] envval,
 * stages[phase].currentlevel, previousValue of trigDetector / holdDetector / retriggerDetector;
 * d_ist = [7][V] int64: phase, state (0 WAITING, 1 TRIGGERED, 2 HOLDING), nxcHappened,
 * stages[phase].counter, firstTrigger of the three detectors.  A freshly set-up object is all zeros
 * except previousValue = 1.0 and firstTrigger = 1 (H:593-594).  State machine bit-exact; the value is exact
 * for curve == 1 and within the device pow's accuracy otherwise. */
int mxg_envgen_render(size_t V, size_t N, const double *d_trig, int tpv, const double *d_stages, int nstages,
                      int loop, int retrigger, double *d_dst, int64_t *d_ist, double *d_out, void *stream);

/* ---- maxiDCBlocker / maxiSVF / maxiBiquad banks (H:1255-1486) ------------------------------------ */
/* kind 0 maxiDCBlocker::play(input, R) H:1261-1266: d_coef = [1][V] R; d_st = [3][V] xm1, ym1, (unused).
 * kind 1 maxiSVF::play(w, lpmix, bpmix, hpmix, notchmix) H:1303-1317: d_coef = [9][V] g1, g2, g3, g4, k
 *        (rows 0-4 from mxg_svf_coeffs_host = setCutoff/setResonance -> setParams H:1320-1332) then the
 *        four mix arguments; d_st = [3][V] v0z, v1, v2.
 * kind 2 maxiBiquad::play(input) H:1360-1367: d_coef = [5][V] a0, a1, a2, b1, b2 (mxg_biquad_coeffs_host
 *        = set(type, cutoff, Q, peakGain) H:1376-1478); d_st = [3][V] v[0], v[1], v[2].
 * Coefficients come from the host libm (tan, pow, sqrt), so the recurrences are bit-exact. */
int mxg_filter2_render(int kind, size_t V, size_t N, const double *d_in, const double *d_coef,
                       double *d_st, double *d_out, void *stream);
int mxg_svf_coeffs_host(size_t V, const double *h_cutoff, const double *h_res, double *h_coef);
/* h_type[v]: 0 LOWPASS 1 HIGHPASS 2 BANDPASS 3 NOTCH 4 PEAK 5 LOWSHELF 6 HIGHSHELF (H:1349-1358) */
int mxg_biquad_coeffs_host(size_t V, const int32_t *h_type, const double *h_cutoff, const double *h_Q,
                           const double *h_peakGain, double *h_coef);

/* ---- maxiEnv bank --------------------------------------------------------------------- */
/* mode 0: adsr(input, trigger) (C:1415-1466)   mode 1: ar(input, attack, release, holdtime,
 * trigger) (C:1319-1358).  d_in NULL = constant 1.0 input.  d_trig: int32 [N] (tpv=0, one
 * trigger signal for the whole bank) or [N][V] (tpv=1).  d_par = [4][V] attack, decay,
 * sustain, release as stored by the setters; d_holdtime int64 [V]; d_dst = [2][V] amplitude,
 * output; d_ist = int64 [6][V] holdcount, attackphase, decayphase, sustainphase, holdphase,
 * releasephase.  All state in/out; zero-initialise to match static-storage maxiEnv objects. */
int mxg_env_render(int mode, size_t V, size_t N, const double *d_in, const int32_t *d_trig,
                   int tpv, const double *d_par, const int64_t *d_holdtime, double *d_dst,
                   int64_t *d_ist, double *d_out, void *stream);
/* maxiEnv setters on the host libm (C:1469-1494): which 0 setAttack 1 setDecay 2 setRelease
 * 3 setAttackMS. */
double mxg_env_coeff_host(int which, double ms);
/* maxiConvert::mtof (H:941, C:1498-1500): the reference's 129-entry table of six-decimal literals, bit for bit (host side;
 * midinote outside [0, 128], which the reference indexes out of bounds, returns 0). */
double mxg_mtof_host(int midinote);

/* ---- fused subtractive voice (saw -> lores -> adsr in registers, one store per sample) ---- */
/* mode 0: out = adsr(lores(saw(freq), cutoff, res), trig); coefficients hoisted: d_coef =
 *         [3][V] from mxg_filter_coeffs_host(MXG_FLT_LORES, ...) (bit-exact).
 * mode 1: e = adsr(1., trig); out = lores(saw(freq), e*cutoff, res) * e  (the call order of
 *         14.monosynth/main.cpp:50-55); cutoff is modulated per sample, coefficients on device,
 *         stated tolerance.  d_cutoff/d_res [V] are used, d_coef ignored.
 * d_ost = [2][V] osc phase/output; d_fst = [5][V] filter state; env arrays as mxg_env_render. */
int mxg_voice_render(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff,
                     const double *d_res, const double *d_coef, const int32_t *d_trig, int tpv,
                     const double *d_par, const int64_t *d_holdtime, double *d_ost, double *d_fst,
                     double *d_dst, int64_t *d_ist, double *d_out, void *stream);
/* The same render with the maxiMix::stereo mixdown of the bank fused into it (C:503-509 applied voice after voice, the sum over
 * voices of the user's loop: 15.polysynth/main.cpp:54-70) -- the block is never read back.  d_out may be NULL (mix only).  The per-voice
 * block and every state array get mxg_voice_render's bits; the sum over voices is a fixed tree (tolerance on the mix, as
 * mxg_osc_render_mix).  _rows: d_rows[mxg_osc_mix_groups(V)][N][2], one row per 256 voices -- what a grouped mix queue's slot takes
 * (mxg_mixq_create_grouped) or mxg_mix_rows_sum adds; _mix: d_mix[N][2] (rows in library scratch + the row sum on `stream`).
 * Knob "voice_mix_store": the store stream of this form (0 automatic, 1 ... 5 as "voice_store"). */
int mxg_voice_render_mix_rows(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff, const double *d_res,
                              const double *d_coef, const int32_t *d_trig, int tpv, const double *d_par, const int64_t *d_holdtime,
                              double *d_ost, double *d_fst, double *d_dst, int64_t *d_ist, double *d_out, const double *d_pan,
                              double *d_rows, void *stream);
int mxg_voice_render_mix(int mode, size_t V, size_t N, const double *d_freq, const double *d_cutoff, const double *d_res,
                         const double *d_coef, const int32_t *d_trig, int tpv, const double *d_par, const int64_t *d_holdtime,
                         double *d_ost, double *d_fst, double *d_dst, int64_t *d_ist, double *d_out, const double *d_pan, double *d_mix,
                         void *stream);

/* ---- maxiMix::stereo + mixdown over voices (C:503-509; user-side sum e.g. 15.polysynth:67) --- */
/* d_mix[n][0..1] = sum_v ( in[n][v]*sqrt(1-pan_v), in[n][v]*sqrt(pan_v) ).  The per-voice
 * products are exact; the sum over voices is a fixed-shape tree (deterministic, but not the
 * reference's sequential order => stated tolerance on the mix). d_mix is [N][2]. */
int mxg_mix_stereo(size_t V, size_t N, const double *d_in, const double *d_pan, double *d_mix,
                   void *stream);

/* maxiMix::stereo (channels 2, C:503-509), quad (4, C:512-522) or ambisonic (8, C:525-541) over a
 * bank: d_x/d_y/d_z are the per-voice [V] pan arguments (d_y for channels >= 4, d_z for 8).
 * d_bus, if not NULL, receives the per-voice bus signals bus[n][c][v] = what the reference leaves
 * in two/four/eight[c] for voice v at sample n (bit-exact, including ambisonic's quirks: z is
 * never clamped, `z>1` / `z<0` overwrite y, and eight[0..3] = input*(sqrt(..)*1.0 - z)).
 * d_mix [N][channels] = the sum over voices (fixed-shape tree, tolerance as mxg_mix_stereo). */
int mxg_mix_bus(int channels, size_t V, size_t N, const double *d_in, const double *d_x,
                const double *d_y, const double *d_z, double *d_bus, double *d_mix, void *stream);

/* ---- maxiDelayline bank ---------------------------------------------------------------- */
/* mode 0: dl(input, size, feedback) (C:420-429)   mode 1: dlFromPosition(input, size, feedback,
 * position) (C:431-439).  d_size int32 [V], d_feedback [V], d_position int32 [V] (mode 1).
 * d_mem = [cap][V] slot-major ring (the reference's per-object memory[88200*8], H:273, cut to
 * `cap` slots; every size must be <= cap), zero-initialise like the ctor (C:415-417);
 * d_phase int32 [V] in/out.  d_in/d_out are [N][V]. */
int mxg_delay_render(int mode, size_t V, size_t N, const double *d_in, const int32_t *d_size,
                     const double *d_feedback, const int32_t *d_position, double *d_mem, size_t cap,
                     int32_t *d_phase, double *d_out, void *stream);

/* ---- maxiSample play family -------------------------------------------------------------- */
typedef enum {
    MXG_SMP_PLAY = 0,                    /* play()                       C:740-747   */
    MXG_SMP_PLAYONCE = 1,                /* playOnce()                   C:982-991   */
    MXG_SMP_PLAYLOOP = 2,                /* playLoop(start,end)          C:960-967   */
    MXG_SMP_PLAYUNTIL = 3,               /* playUntil(end)               C:969-978   */
    MXG_SMP_PLAYATSPEED = 4,             /* playAtSpeed(a)               C:1060-1075 */
    MXG_SMP_PLAYONCEATSPEED = 5,         /* playOnceAtSpeed(a)           C:994-1003  */
    MXG_SMP_PLAYUNTILATSPEED = 6,        /* playUntilAtSpeed(end,a)      C:1047-1058 */
    MXG_SMP_PLAY4 = 7,                   /* play4(a=frequency,start,end) C:884-956   */
    MXG_SMP_PLAYATSPEEDBETWEENPOINTS = 8,/* playAtSpeedBetweenPoints(a=frequency,start,end) C:823-880 */
    /* trigger-driven players, mxg_sample_render_trig only */
    MXG_SMP_PLAYONZX = 9,                /* playOnZX(trig)                                  C:1006-1011 */
    MXG_SMP_PLAYONZXATSPEED = 10,        /* playOnZXAtSpeed(trig, a)                        C:1013-1018 */
    MXG_SMP_PLAYONZXATSPEEDFROMOFFSET = 11, /* playOnZXAtSpeedFromOffset(trig, a, p0=offset) C:1020-1026 */
    MXG_SMP_PLAYONZXATSPEEDBETWEENPOINTS = 12, /* ...BetweenPoints(trig, a, p0=offset, p1=length) C:1028-1035 */
    MXG_SMP_LOOPSETPOSONZX = 13,         /* loopSetPosOnZX(trig, p0=pos)                    C:1037-1042 */
    MXG_SMP_PLAYWITHPHASOR = 14          /* playWithPhasor(trig=pha)                        C:753-816   */
} mxg_sample_mode;
/* Upload a mono sample (what maxiSample::setSample holds, H:670-678) into a device buffer that
 * is valid on [-4, len+5] with 0.0 guards: the reference reads amplitudes[len], [len+1]
 * (C:1063-1064) and [-1] (C:898) out of bounds; zero is the parity convention.  The
 * wider margin backs the players' index clamp: where the reference itself indexes outside its
 * vector (undefined: a play4 step longer than its loop, playLoop with end > 1, a head uploaded far
 * outside the sample) the device reads a guard zero, never memory outside this allocation.
 * Returns the device pointer to element 0 (NULL on failure); release with mxg_sample_free. */
double *mxg_sample_upload(const double *h_samples, size_t len);
int mxg_sample_free(double *d_samples);
/* V play heads over one shared sample.  d_a: speed or frequency, [V] (aps=0) or [N][V] (aps=1);
 * d_start/d_end [V] (fractions of the length for playLoop/playUntil*, sample indices for
 * play4/playAtSpeedBetweenPoints, exactly as in the reference); d_position [V] in/out is the
 * member `position` (setSample leaves it at len-1, H:677; trigger() sets 0, C:597-600).
 * mySampleRate is the int member (44100 after setSample): the step divides by the INTEGER
 * quotient sampleRate/mySampleRate (C:1070). */
int mxg_sample_render(int mode, size_t V, size_t N, const double *d_samples, size_t len,
                      int mySampleRate, const double *d_a, int aps, const double *d_start,
                      const double *d_end, double *d_position, double *d_out, void *stream);

/* maxiSample::playAtSpeedBetweenPointsFromPos(frequency, start, end, pos) (C:826-880) with the CALLER's position signal
 * d_pos [N][V]: the member `position` takes no part (it is the by-value parameter of C:823-825), so the call is a pure
 * function of its arguments.  d_freq [V] (fps = 0) or [N][V]; d_start / d_end [V] in samples.  Bit-exact. */
int mxg_sample_render_frompos(size_t V, size_t N, const double *d_samples, size_t len, const double *d_freq, int fps,
                              const double *d_start, const double *d_end, const double *d_pos, double *d_out, void *stream);

/* Trigger-driven players (modes 9-14).  d_trig is the per-sample [N][V] first argument of the
 * reference call: the trigger signal of playOnZX* / loopSetPosOnZX, or the phasor of
 * playWithPhasor.  d_a = speed ([V], or [N][V] when aps), d_p0/d_p1 = per-voice offset/length (or
 * pos), as listed in mxg_sample_mode.  State, in/out: d_position [V]; d_tprev [V] / d_tfirst [V] =
 * maxiSample::zxTrig's previousValue / firstTrigger (H:593-594: a fresh object holds 1.0 / 1), or
 * for mode 14 phasorPrev / phasorFirst (H:731-732: 0.0 / 1; d_position is not used and may be NULL).
 * All of it is compares, + - * / round and integer indexing => bit-exact. */
int mxg_sample_render_trig(int mode, size_t V, size_t N, const double *d_samples, size_t len,
                           int mySampleRate, const double *d_trig, const double *d_a, int aps,
                           const double *d_p0, const double *d_p1, double *d_position,
                           double *d_tprev, int32_t *d_tfirst, double *d_out, void *stream);

/* int32 <-> int64 copies of a device array (hosts that carry every integer member of a slot as int64 -- include/maximilian.h --
 * feed the int32 flag arrays above through these). */
int mxg_i64_from_i32(int64_t *d_dst, const int32_t *d_src, size_t n, void *stream);
int mxg_i32_from_i64(int32_t *d_dst, const int64_t *d_src, size_t n, void *stream);

/* maxiSample::load(fileName, channel) / read() (C:605-692): parse a 16-bit PCM RIFF/WAVE file the way the
 * reference walks it, upload the payload as int16 and de-interleave + normalise on the device
 * (amplitudes[i] = short/32767.0, C:679, bit-exact).  Returns the device sample buffer (guarded like
 * mxg_sample_upload, free with mxg_sample_free) or NULL (cannot open / malformed; the reference returns
 * false / reads garbage).  *h_len = amplitudes.size() = dataSize/2 -- for a multi-channel file that is the
 * FULL interleaved length: the reference keeps every (2*channels)-th short at the front and leaves the rest
 * (C:667-674, see wav.hip).  After read() the member `position` equals the size (C:681): start the bank's
 * play heads there.  h_hdr (int32[8], may be NULL) receives ChunkSize, SubChunk1Size, Format, Channels,
 * SampleRate, ByteRate, BlockAlign, BitsPerSample (set the bank's mySampleRate from [4]). */
double *mxg_sample_load_wav(const char *path, int channel, size_t *h_len, int32_t *h_hdr);
/* maxiSample::save(filename) (C:698-725): shorts = static_cast<short>(round(a*32767.0)) computed on the
 * device, written behind the 44-byte header built from h_hdr (same 8 fields).  Synchronous. */
int mxg_sample_save_wav(const char *path, const double *d_samples, size_t len, const int32_t *h_hdr,
                        void *stream);

/* ---- maxiSampler banks (L/maxiSynths.h:137-187, maxiSynths.cpp:262-300) ---------------------------------- */
/* V = NS * voices slots (any voices in 1 .. 32, as maxiSampler::setNumVoices accepts: maxiSynths.cpp:284-289), all over one sample
 * buffer (mxg_sample_upload / mxg_sample_load_wav).  Renders N calls of maxiSampler::play() for every
 * sampler: per slot envOut = adsr(gain, trigger); if (envOut > 0) { outputs = play4(freq, 0, len)*envOut;
 * output += outputs/voices; if (trigger == 1 && !sustain) trigger = 0; }.  d_mix [N][NS] = play();
 * d_outputs (optional) [N][V] = the member outputs[i] after each call.  d_freq [V] from
 * mxg_sampler_freq_host (pitchRatios[(int)pitch + 67] * ((1./len)*sampleRate), maxiSynths.cpp:297).
 * Slot state, in/out: d_position [V], d_trigger [V] (envelopes[i].trigger), d_outhold [V] (outputs[i]),
 * d_dst [2][V] / d_ist [6][V] as mxg_env_render; d_gain [V] = envOutGain, d_par [4][V] + d_holdtime [V] the
 * envelope parameters.  The control calls (trigger(), midiNoteOn/Off, setPitch: maxiSynths.cpp:351-391,
 * 484-491) edit that state between renders on the host.  Bit-exact, including the in-order sum. */
int mxg_sampler_freq_host(size_t V, const double *h_pitch, size_t len, double *h_freq);
int mxg_sampler_render(size_t V, size_t N, int voices, int sustain, const double *d_samples, size_t len,
                       const double *d_freq, const double *d_gain, const double *d_par,
                       const int64_t *d_holdtime, double *d_position, int32_t *d_trigger, double *d_outhold,
                       double *d_dst, int64_t *d_ist, double *d_mix, double *d_outputs, void *stream);

/* ---- maxiFFT batch ---------------------------------------------------------------------- */
/* A plan is what maxiFFT::setup(fftSize, hopSize, windowSize) prepares (L/maxiFFT.cpp:45-60): the
 * Hann window (fft::genWindow type 3, L/fft.cpp:409-413) and the fp32 twiddle sequences the
 * reference generates by recurrence (L/fft.cpp:161-182, :245-272), replayed on the host.
 * fftSize: power of two in [8, 8192] (the reference exit(1)s otherwise, L/fft.cpp:129-132 ->
 * here NULL + MXG_ERR_INVALID); windowSize < fftSize is raised to fftSize as in the reference;
 * windowSize > fftSize overruns the reference's buffers and is rejected. */
typedef struct mxg_fft_plan mxg_fft_plan;
mxg_fft_plan *mxg_fft_plan_create(int fftSize, int hopSize, int windowSize);
int mxg_fft_plan_destroy(mxg_fft_plan *plan);
int mxg_fft_plan_bins(const mxg_fft_plan *plan); /* getNumBins() = fftSize/2 */
/* Transform nframes frames; frame k = d_signal[k*frame_stride .. k*frame_stride + fftSize).
 * With frame_stride = hopSize over a signal prefixed by (windowSize-hopSize) zeros this is exactly
 * the sequence of frames maxiFFT::process() analyses (L/maxiFFT.cpp:65-91).  Outputs are
 * [nframes][bins] fp32, any of them may be NULL: real/imag = getReal()/getImag() (bin 0 packs
 * DC and Nyquist, L/fft.cpp:274-275), mags/phases = getMagnitudes()/getPhases() (cartToPol,
 * L/fft.cpp:507-515).  real, imag and mags are bit-exact; phases use the device atan2f. */
int mxg_fft_batch(const mxg_fft_plan *plan, const float *d_signal, size_t frame_stride,
                  size_t nframes, float *d_real, float *d_imag, float *d_mags, float *d_phases,
                  void *stream);

/* maxiFFT::magsToDB (fft::convToDB, L/fft.cpp:526-534), spectralFlatness and spectralCentroid
 * (L/maxiFFT.cpp:113-132) of nframes frames of magnitudes d_mags [nframes][bins] (e.g. the d_mags
 * output of mxg_fft_batch).  Outputs, any may be NULL: d_db [nframes][bins], d_flatness [nframes],
 * d_centroid [nframes] (uses maxiSettings::sampleRate).  The per-frame sums run over the bins in the
 * reference's order in float; centroid is bit-exact, dB and flatness go through the device
 * log10f/logf/expf (tolerances in DESIGN.md). */
int mxg_fft_features(const mxg_fft_plan *plan, const float *d_mags, size_t nframes, float *d_db,
                     float *d_flatness, float *d_centroid, void *stream);

/* ---- maxiIFFT batch (L/maxiFFT.cpp:140-192, SPECTRUM mode) ---------------------------------------- */
/* maxiIFFT::setup(fftSize, hopSize, windowSize): Hann window over (windowSize ? windowSize : fftSize),
 * zero beyond (note: not maxiFFT's max(windowSize, fftSize)).  fftSize: power of two in [8, 8192]. */
typedef struct mxg_ifft_plan mxg_ifft_plan;
mxg_ifft_plan *mxg_ifft_plan_create(int fftSize, int hopSize, int windowSize);
int mxg_ifft_plan_destroy(mxg_ifft_plan *plan);
/* nframes spectra d_mags/d_phases [nframes][bins] -> d_signal [nframes*hopSize]: the samples
 * maxiIFFT::process(mags, phases) returns when it is called hopSize times per spectrum (the spectrum is
 * consumed at pos == 0, L/maxiFFT.cpp:156).  Per frame: polToCart (L/fft.cpp:590-604), a full fftSize-point
 * complex inverse FFT with the reference's fp32 recurrence twiddles, /fftSize, x window (:606-611), then
 * the overlap-add hop buffer (:176-183) evaluated in the reference's order of additions.  d_buffer
 * ([fftSize], in/out, may be NULL = a fresh object) is the member `buffer`, so consecutive calls continue
 * one stream.  d_ifft_out (optional [nframes][fftSize]) receives each frame's windowed inverse transform
 * (`ifftOut`).  Bit-exact given the same cartesian inputs; the float cos/sin of polToCart are the
 * device's => tolerance.  COMPLEX mode of the reference copies its inputs into the wrong
 * arrays (L/fft.cpp:613-619: out_real/out_img, which calcIFFT then overwrites) and is not provided. */
int mxg_ifft_batch(const mxg_ifft_plan *plan, const float *d_mags, const float *d_phases, size_t nframes,
                   float *d_buffer, float *d_signal, float *d_ifft_out, void *stream);

/* maxiIFFT::process(real, imag, COMPLEX) (L/maxiFFT.cpp:155-192 with fftModes COMPLEX).  The reference's
 * fft::inverseFFTComplex (L/fft.cpp:613-619) copies its inputs into out_real/out_img -- the arrays calcIFFT then overwrites
 * with the inverse transform of in_real/in_img, which COMPLEX mode never writes.  On a fresh object those are the zeros
 * of fft::setup, so every frame's inverse transform is exactly 0 and the call returns the carried hop buffer running
 * out: as_reference = 1 reproduces that, bit for bit.  as_reference = 0 is what the function was written to do: real/imag
 * [nframes][bins] go to the transform's inputs (negative frequencies zero, like polToCart), everything after -- the
 * full-size inverse FFT with replayed twiddles, /fftSize, window, overlap-add -- exactly as mxg_ifft_batch; bit-exact
 * against the reference's own calcIFFT fed that way (tests/test_gpu_convolve.py). */
int mxg_ifft_batch_complex(const mxg_ifft_plan *plan, const float *d_real, const float *d_imag, size_t nframes,
                           int as_reference, float *d_buffer, float *d_signal, float *d_ifft_out, void *stream);

/* ---- maxiConvolve (L/maxiConvolve.cpp:13-107): partitioned convolution over a frequency delay line ------------------ */
/* maxiConvolve::setup(impulseFile, fftsize, hopsize) with the impulse already in memory: h_amplitudes [len] = the
 * loaded maxiSample's amplitudes, position0 = its play head (load()/read() leave it at len, C:681).  Replicated: the
 * impulse is played with play() (C:740-747; the first read is amplitudes[len], one past the end = 0 here) len times
 * into maxiFFT::setup(fftsize, fftsize, hopsize) -- hop = fftsize, full Hann window: non-overlapping frames (:35-40);
 * then getNumBins() - len % getNumBins() zeros (:41-45, which may or may not complete a last frame); the frames'
 * real / imaginary spectra are divided by the largest positive real / imaginary value (:47-52, float division).
 * The analysis runs on the device (mxg_fft_batch, bit-exact); NULL on failure. */
typedef struct mxg_convolve mxg_convolve;
mxg_convolve *mxg_convolve_create(const double *h_amplitudes, size_t len, double position0, int fftsize, int hopsize);
int mxg_convolve_destroy(mxg_convolve *c);
int mxg_convolve_frames(const mxg_convolve *c);  /* impulseReal.size() */
/* the normalised impulse spectra [frames][bins]; either pointer may be NULL */
int mxg_convolve_impulse(const mxg_convolve *c, float *h_real, float *h_imag);
/* play(w) (:76-107) for nblocks*fftsize consecutive samples d_in -> d_out (float, device).  The delay line, the current
 * sums and maxiIFFT's buffer are the object's state and carry from call to call.  Per fftsize samples: the input frame's
 * spectrum enters the delay line; sumReal/sumImag = sum over k of impulse[k] (x) FDL[k] accumulated in float in k order
 * (bin 0: real*real and imag*imag only, :86-87); block b of the output is the maxiIFFT (COMPLEX mode,
 * setup(fftsize, fftsize, hopsize)) of the sums formed at the end of block b-1.
 * mode 0 = as the reference computes (the COMPLEX-mode defect above: silence, while the state still advances);
 * mode 1 = as intended (the sums reach the inverse transform).  Both bit-exact against the reference (tests/test_gpu_convolve.py). */
int mxg_convolve_play(mxg_convolve *c, const float *d_in, size_t nblocks, float *d_out, int mode, void *stream);
int mxg_convolve_reset(mxg_convolve *c);
/* The two halves of one block of play(), for a host that is called once per sample (include/maxiConvolve.h): the output of the
 * NEXT fftsize samples depends only on the sums the previous input frame left, so it can be fetched before those samples'
 * inputs exist; mxg_convolve_input then takes the fftsize inputs (delay line, sums).  output + input == play(nblocks = 1). */
int mxg_convolve_output(mxg_convolve *c, float *d_out, int mode, void *stream);
int mxg_convolve_input(mxg_convolve *c, const float *d_in, void *stream);

/* ---- maxiMFCC batch --------------------------------------------------------------------- */
/* maxiMFCC::setup(numBins, numFilters, numCoeffs, minFreq, maxFreq) (L/maxiMFCC.h:56-75): builds
 * the mel filterbank and DCT tables on the host libm.  Works without a device (tables only). */
typedef struct mxg_mfcc_plan mxg_mfcc_plan;
mxg_mfcc_plan *mxg_mfcc_plan_create(unsigned numBins, unsigned numFilters, unsigned numCoeffs,
                                    double minFreq, double maxFreq);
int mxg_mfcc_plan_destroy(mxg_mfcc_plan *plan);
/* Copy out the host tables in the reference's layouts: melFilters[filter + bin*numFilters],
 * dct[i + j*numCoeffs].  Either may be NULL.  Returns the number of leading bins that carry a
 * non-zero weight. */
int mxg_mfcc_plan_tables(const mxg_mfcc_plan *plan, double *h_melFilters, double *h_dct);
/* The same tables cut for the matrix pipe (knob "fused_mel" of mxg_fft_mfcc_batch): filters in quads, two quads per pair,
 * each quad contracting over a band of 16 * nb[pair] bins from base[pair][quad] on.  h_nb [6], h_base [6][2],
 * h_W [(batches + 1) * 128] = [batch][half][lane32 = 8 k + 4 quad + filter-of-quad][2] weights (melFilters of
 * L/maxiMFCC.h:118-182, zero outside a filter's support), h_D [4][6][32] = dct (L/maxiMFCC.h:183-203) as
 * [coefficient quad][pair][8 filter-of-quad + 4 quad + coefficient-of-quad].  Any pointer may be NULL.  Returns the number
 * of batches, 0 when the bank has no such tables (more than 48 filters or 16 coefficients, a filter reading bin 0 or beyond
 * bin 255).  Works without a device. */
int mxg_mfcc_plan_matrix_tables(const mxg_mfcc_plan *plan, int *h_nb, int *h_base, double *h_W, size_t capW, double *h_D);
/* mfcc() over nframes spectra d_mags[f*mag_stride + bin] (fp32, as getMagnitudes() yields) ->
 * d_mfcc [nframes][numCoeffs] (L/maxiMFCC.h:77-81).  Optional: d_melraw = the band sums before
 * the log (bit-exact with method 0), d_melbands = melBands after log-square, [nframes][numFilters].
 * method 0: exact sparse evaluation in the reference's summation order; method 1: dense fp64 MFMA
 * contraction (v_mfma_f64_16x16x4_f64), reordered/fused sums => tolerance. */
int mxg_mfcc_batch(const mxg_mfcc_plan *plan, const float *d_mags, size_t mag_stride, size_t nframes,
                   double *d_melraw, double *d_melbands, double *d_mfcc, int method, void *stream);

/* ---- maxiFFT + maxiMFCC in one pass (the loop of cpp/commandline/tests/mfcctest/mfcctest.cpp:21-32 over a batch) ---- */
/* For every frame k = d_signal[k*frame_stride .. +1024): mags = maxiFFT(1024).process(...) magnitudes (L/fft.cpp:499-511),
 * then maxiMFCC::mfcc(mags) (L/maxiMFCC.h:77-81) -> d_mfcc [nframes][numCoeffs], in ONE kernel: the magnitudes stay in
 * LDS, so a frame costs its 4096 B of input and its numCoeffs*8 B of output in HBM traffic.  fft_plan must be a
 * 1024-point plan and mfcc_plan one for 512 bins with at most 64 filters (otherwise MXG_ERR_INVALID: use mxg_fft_batch +
 * mxg_mfcc_batch).  Optional outputs (NULL = not produced): d_mags [nframes][512] (bit-exact, as mxg_fft_batch),
 * d_melraw / d_melbands [nframes][numFilters] (as mxg_mfcc_batch method 0: band sums bit-exact, log-square / mfcc
 * within the device log's tolerance). */
int mxg_fft_mfcc_batch(const mxg_fft_plan *fft_plan, const mxg_mfcc_plan *mfcc_plan, const float *d_signal,
                       size_t frame_stride, size_t nframes, float *d_mags, double *d_melraw, double *d_melbands,
                       double *d_mfcc, void *stream);

/* ---- maxiGrains: maxiTimeStretch / maxiStretch banks --------------------------------------- */
/* Window kinds = the functors of L/maxiGrains.h:18-90: 0 hann 1 hamming 2 cosine 3 rect 4 triangle
 * 5 triangleNZ 6 blackmanHarris 7 blackmanNutall 8 gaussian(0.3).  A plan holds the window table
 * maxiGrainWindowCache::getWindow (:112-120) would build for grains of grainLength seconds of a
 * sample at mySampleRate (host libm); the length must be < sampleRate/2 like the cache (:98). */
typedef struct mxg_grain_plan mxg_grain_plan;
mxg_grain_plan *mxg_grain_plan_create(int window_kind, double grainLength, int mySampleRate);
int mxg_grain_plan_destroy(mxg_grain_plan *plan);
int mxg_grain_plan_window(const mxg_grain_plan *plan, double *h_window); /* returns sampleDur */
/* S independent streams over one shared sample (mxg_sample_upload), T samples each, d_out[n*S + s].
 * mode 0 = maxiTimeStretch<F>::play(speed = a[s], grainLength, overlaps, posMod[s]) (:341-355);
 * mode 1 = maxiStretch<F>::play(pitchstretch = a[s], timestretch = b[s], grainLength, overlaps,
 * posMod[s]) with the default loop (whole sample) (:512-530);
 * mode 2 = maxiTimeStretch<F>::playAtPosition(pos = a[n*S + s], grainLength, overlaps) (:359-367):
 * d_a is the per-sample [T][S] position signal, the member `position` is not touched;
 * mode 3 = maxiPitchShift<F>::play(speed = a[s], grainLength, overlaps, posMod[s]) (:412-430): the
 * `looper` slot of d_st holds the member `cycles` (a long), grains are born with
 * speed - (cycleMod/cycleLength)*0.1 and therefore arbitrary increments; no rand() is drawn.
 * d_posmod may be NULL (0).
 * d_rnd: int32 [S][R], the values `rand() % 10` returns at each spawn, consumed in order per
 * stream (NULL = 0): the reference draws them from the process-wide rand() stream, which a bank
 * cannot reproduce.  State, in/out: d_st = [4][S] position, looper, randomOffset, rand cursor
 * (setPosition(p) == position = clamp(p*len, 0, len-1), :335-338); d_gst = [4][8][S]: the live
 * grains in creation order: pos, inc, sampleIdx, sampleDur (0 = empty slot).  A live grain must be
 * one this plan made (same sampleDur: a bank has one window table per plan, so let grains finish -- or
 * clear d_gst -- before switching to a plan of another grain length).  The call only enqueues on `stream`; what the device finds while
 * it renders -- more than 8 live grains in a stream, an exhausted d_rnd, such a foreign / corrupt grain, a grain born with a step its
 * reads could not survive: MXG_ERR_INVALID -- is reported by the next synchronising call or by mxg_last_async_error (knob "grain_sync" 1:
 * by the call itself, which then waits).  A tile-rendered call is ONE launch whose first workgroups are the per-stream schedulers and whose
 * other workgroups render tiles as the schedulers' lists reach them (knob "grain_streamed"). */
int mxg_granular_render(const mxg_grain_plan *plan, int mode, size_t S, size_t T, const double *d_samples,
                        size_t len, int overlaps, const double *d_a, const double *d_b,
                        const double *d_posmod, const int32_t *d_rnd, size_t R, double *d_st, double *d_gst,
                        double *d_out, void *stream);

/* mxg_granular_render plus the maxiMix::stereo mixdown of the S streams (C:503-509 and the user-side sum,
 * e.g. ofApp.cpp:166-173 of the reference's granular example): d_pan [S], d_mix [T][2] -- BASELINE configs[4]'s "stereo
 * mixdown".  On the unit-increment path (maxiTimeStretch at 44.1 kHz) the render kernel mixes each 64-stream x 64-sample
 * tile while it still sits in LDS, so the [T][S] block is not read back; every other path runs mxg_mix_stereo after the
 * render.  Per-stream outputs bit-exact as mxg_granular_render; the sum over streams is tree-ordered (mix tolerance). */
int mxg_granular_render_mix(const mxg_grain_plan *plan, int mode, size_t S, size_t T, const double *d_samples,
                            size_t len, int overlaps, const double *d_a, const double *d_b,
                            const double *d_posmod, const int32_t *d_rnd, size_t R, double *d_st, double *d_gst,
                            double *d_out, const double *d_pan, double *d_mix, void *stream);

/* ---- multi-GPU mixdown: RCCL over xGMI ------------------------------------------------ */
/* The reference is single-device; there is nothing to cite but the mix itself (maxiMix, C:503-541, and the user-side
 * sum over voices, 15.polysynth/main.cpp:67).  Banks shard over the GPUs of a node by contiguous voice / stream /
 * frame ranges, one process per GPU, private state, no data-path collective -- except that the per-rank
 * [samples][channels] fp64 mix blocks are summed onto one GPU with ncclReduce(ncclDouble, ncclSum, root).
 * Sum order across ranks is RCCL's, not the reference's sequential voice order: the mix carries the tolerance of
 * mxg_mix_stereo, per-voice signals stay bit-exact.
 * mxg_comm_unique_id: rank 0 obtains the 128-byte ncclUniqueId; the host distributes it (torch.distributed, MPI,
 * a file ...); every rank then calls mxg_comm_create(id, nranks, rank) on the device it selected with mxg_init.
 * librccl is resolved at this point (dlopen by SONAME), not when libmaxigpu.so loads. */
#define MXG_COMM_ID_BYTES 128
typedef struct mxg_comm mxg_comm;
int mxg_comm_unique_id(void *h_id);
mxg_comm *mxg_comm_create(const void *h_id, int nranks, int rank);
int mxg_comm_destroy(mxg_comm *comm);
int mxg_comm_rank(const mxg_comm *comm);
int mxg_comm_size(const mxg_comm *comm);
/* d_recv (on `root`; on every rank when all != 0) = sum over ranks of d_send[count], in `stream` order (in place
 * allowed).  comm NULL: a device copy (a one-rank communicator goes through RCCL like any other). */
int mxg_comm_reduce(mxg_comm *comm, const double *d_send, double *d_recv, size_t count, int root, int all,
                    void *stream);
/* maxiMix bus over this rank's V voices (as mxg_mix_bus without d_bus) into d_mix [N][channels], then the sum over
 * ranks into the root's d_mix -- the whole exchange step for a host that reduces block by block. */
int mxg_mix_reduce(mxg_comm *comm, int channels, size_t V, size_t N, const double *d_in, const double *d_x,
                   const double *d_y, const double *d_z, double *d_mix, int root, void *stream);
/* Batched, overlapped form: an 8 KiB [512][2] block is a latency-bound message, so the local mixes of `depth_blocks`
 * consecutive blocks are staged and reduced with ONE ncclReduce (depth 16 = 128 KiB) on the queue's own stream while
 * the caller's stream renders the next batch into the second staging buffer (device-side event ordering only; no call
 * here blocks the host).  Per block:  p = mxg_mixq_slot(q, stream);  <enqueue the local mix of the block into p, e.g.
 * mxg_osc_render_mix / mxg_mix_stereo with d_mix = p>;  mxg_mixq_push(q, stream).  mxg_mixq_flush reduces a partial
 * batch and makes `stream` wait for every outstanding reduce.  mxg_mixq_result: device pointer to the most recently
 * submitted batch's sum [blocks][block_doubles] (meaningful on the root once its reduce has completed, e.g. after
 * flush + stream sync; the reduce of the batch after next overwrites it ON THE QUEUE'S STREAM -- a consumer that reads it with
 * work enqueued on a stream calls mxg_mixq_release(q, that stream) after enqueueing its reads, and the overwriting reduce is
 * ordered behind them; a consumer that synchronises and copies before it pushes further blocks needs nothing).  mxg_mixq_set_sink: the root additionally copies every summed
 * block into a pinned host ring [ring_blocks][block_doubles] (the audio callback's side of the boundary). */
typedef struct mxg_mixq mxg_mixq;
mxg_mixq *mxg_mixq_create(mxg_comm *comm, size_t block_doubles, int depth_blocks, int root);
/* Grouped slots: mxg_mixq_slot hands out [groups][block_doubles] -- the rows of mxg_osc_render_mix_rows -- and the queue adds the
 * rows of a whole batch with one kernel on ITS stream in front of the reduce (groups = 1: the plain queue). */
mxg_mixq *mxg_mixq_create_grouped(mxg_comm *comm, size_t block_doubles, int depth_blocks, int root, size_t groups);
int mxg_mixq_destroy(mxg_mixq *q);
int mxg_mixq_set_sink(mxg_mixq *q, double *h_pinned, size_t ring_blocks);
double *mxg_mixq_slot(mxg_mixq *q, void *stream);
int mxg_mixq_push(mxg_mixq *q, void *stream);
int mxg_mixq_flush(mxg_mixq *q, void *stream);
const double *mxg_mixq_result(const mxg_mixq *q, size_t *h_blocks, size_t *h_batches);
int mxg_mixq_release(mxg_mixq *q, void *stream);

/* (The bandwidth-probe kernels bench.py and tools/ measure ceilings with are NOT part of this ABI: include/maxicalib.h,
 * libmaxicalib.so -- measurement code, built beside the product library.) */

#ifdef __cplusplus
}
#endif
#endif /* MAXIGPU_H */
