"""GPU (-m gpu): the DROP-IN boundary.  host/dropin_{01,14,15} are the reference's own example patches --
cpp/commandline/main.cpp (BASELINE config 1), maximilian_examples/14.monosynth/main.cpp and 15.polysynth/main.cpp --
compiled VERBATIM (host/Makefile, where /root/reference exists) against include/maximilian.h, whose maxiOsc / maxiEnv /
maxiFilter run on the GPU through the C-ABI, and driven by mxg_host_render (the loop of cpp/commandline/player.cpp:25-44).
Expected samples: tests/golden/dropin.npz, dumped from the SAME source files linked with the unmodified reference
library (oracle/example_host.cpp).  14 and 15 must match bit for bit (pulse / sinebuf / phasor oscillators, the lores
filter with host-libm coefficients, the adsr state machine, and the user code's own host arithmetic in between);
01 (sinewave) within 1 ULP.  Ten more of the examples (2.TwoTones ... 13.Advanced-Filters) follow at the end of the file."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal, ulp_diff

pytestmark = pytest.mark.gpu


def run_dropin(ex, frames, tmp_path, cwd=None, stdout=subprocess.DEVNULL):
    exe = os.path.join(ROOT, "host", "dropin_" + ex)
    if not os.path.exists(exe):
        pytest.fail("host/dropin_%s is not built (python -c 'import __graft_entry__ as g; g.build()' where /root/reference exists)" % ex)
    out = str(tmp_path / ("dropin_%s.f64" % ex))
    r = subprocess.run([exe, str(frames), out], stdout=stdout, stderr=subprocess.PIPE, text=True, timeout=900, cwd=cwd)
    assert r.returncode == 0, r.stderr
    return np.fromfile(out, np.float64).reshape(frames, 2), (r.stderr if stdout is subprocess.DEVNULL else r.stdout)


def test_config1_sinewave_patch(golden, tmp_path):
    """cpp/commandline/main.cpp verbatim: one maxiOsc::sinewave(440) into both channels, 44 100 frames.  Constant
    arguments: after the block length has doubled up to 512 every block is rendered asynchronously ahead of the audio
    thread (the statistics line reports them)."""
    exp = golden("dropin.npz")["ex01"]
    got, log = run_dropin("01", exp.shape[0], tmp_path)
    assert ulp_diff(got, exp).max() <= 1
    assert np.array_equal(got[:, 0].view(np.uint64), got[:, 1].view(np.uint64))
    m = re.search(r"osc (\d+) \(async blocks (\d+)\)", log)
    launches, hits = int(m.group(1)), int(m.group(2))
    assert hits >= 44100 // 512 - 3 and launches <= 120, log     # ~86 blocks of 512 + the 10 doublings


def test_monosynth_patch_bit_exact(golden, tmp_path):
    exp = golden("dropin.npz")["ex14"]
    got, log = run_dropin("14", exp.shape[0], tmp_path)
    assert_bits_equal(got, exp, "14.monosynth through the drop-in header")
    assert np.abs(exp).max() > 0.5      # the tick at frame 88 200 is inside the window


def test_polysynth_patch_bit_exact(golden, tmp_path):
    exp = golden("dropin.npz")["ex15"]
    got, log = run_dropin("15", exp.shape[0], tmp_path)
    assert_bits_equal(got, exp, "15.polysynth through the drop-in header")
    assert np.abs(exp).max() > 0.05


def _launches(log):
    m = re.search(r"launches: osc (\d+) .*?env (\d+) .*?filter (\d+)", log)
    return int(m.group(1)) + int(m.group(2)) + int(m.group(3))


@pytest.mark.parametrize("ex,frames,max_launches_per_1000", [("14", 20000, 150), ("15", 12000, 400), ("05", 6000, 150), ("10", 6000, 250)])
def test_signal_arguments_are_predicted_per_sample(tmp_path, ex, frames, max_launches_per_1000):
    """Derived arguments (include/maximilian.h): a cutoff that follows an envelope, a frequency that follows an LFO, a filter input
    that is the sum of two oscillators are predicted per sample from the producers' cached blocks, so these patches render in blocks
    instead of one launch per call -- and with the prediction switched off (MXG_PS_DERIVE=0) the output is the same bits, at one
    launch per call and object.  (That those bits are the reference's is what the tests around this one check.)"""
    got, log = run_dropin(ex, frames, tmp_path)
    n_on = _launches(log)
    env = dict(os.environ, MXG_PS_DERIVE="0")
    exe = os.path.join(ROOT, "host", "dropin_" + ex)
    out2 = str(tmp_path / "off.f64")
    r = subprocess.run([exe, str(frames), out2], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr
    off = np.fromfile(out2, np.float64).reshape(frames, 2)
    n_off = _launches(r.stderr)
    assert np.array_equal(got.view(np.uint64), off.view(np.uint64)), "derived-argument prediction changed the samples"
    print("%s: %d launches with derived arguments, %d without, %d frames" % (ex, n_on, n_off, frames))
    assert n_on * 1000 <= max_launches_per_1000 * frames, log
    assert n_off >= 5 * n_on, "the patch was expected to need one launch per call without the prediction"


# ten more of the reference's own example patches, compiled verbatim (host/Makefile dropin_<tag>): what each one exercises and
# how closely it can match.  Patches built only from the wavetable / ramp oscillators, the envelope and the filter are
# bit-identical; a sinewave is within 1 ULP of glibc's, so sums / products of sinewaves carry a few ULP, and where a sinewave
# drives another oscillator's FREQUENCY (FM) the carrier's phase inherits that last-bit difference and integrates it.
FFT_EXAMPLE_TOL = 1e-6   # measured 8.9e-8 on a peak of 0.635 (float32 output; cosf / sinf of arguments up to a few hundred)
EXAMPLES = {
    "02": ("2.TwoTones: sinewave(440) + sinewave(441)", 4.5e-16),
    "03": ("3.AM1: sinewave(440) * sinewave(1)", 3.5e-16),
    "04": ("4.AM2: sinewave(440) * sinewave(phasorBetween(0.01, 0, 440))", 3.5e-16),
    "05": ("5.FM1: sinewave(440 + sinewave(1) * 100)", 2e-14),     # measured 8.9e-16 over 6000 frames
    "06": ("6.FM2: sinewave(sinewave(sinewave(0.1) * 30) * 440)", 4e-14),  # measured 1.8e-15
    "08b": ("8.Counting2: sawn(int(phasorBetween(1, 1, 9)) * 100)", 0.0),
    "08c": ("8.Counting3: square / sinewave switched by the counter", 1.2e-16),
    "08d": ("8.Counting4: counter rate from a sawn LFO, array lookup, square / sawn", 0.0),
    "10": ("10.Filters: adsr gated by a counter, sawn through lores with the envelope on its cutoff", 0.0),
    "11": ("11.Mixing: noise() (the process's rand() stream) panned by maxiMix::stereo with a sinewave autopanner", 4.5e-16),
    "13": ("13.Advanced-Filters: sawn through the patch's own float biquad", 0.0),
    "16": ("16.Replicant: seven oscillators, adsr / ar with explicit coefficients, two lores filters, a delay line, mtof", 0.0),
}


@pytest.mark.parametrize("ex", list(EXAMPLES))
def test_more_reference_examples_verbatim(golden, tmp_path, ex):
    what, tol = EXAMPLES[ex]
    exp = golden("dropin_examples.npz")["ex" + ex]
    got, log = run_dropin(ex, exp.shape[0], tmp_path)
    if ex not in ("11", "13"):
        assert np.array_equal(got[:, 0].view(np.uint64), got[:, 1].view(np.uint64))   # output[1] = output[0]
    err = np.nanmax(np.abs(got[:, 0] - exp))
    print("%s: max |difference| %.3e (allowed %.1e)" % (what, err, tol))
    if tol == 0.0:
        assert_bits_equal(got[:, 0], exp, what)
    else:
        assert err <= tol, what
    assert np.nanmax(np.abs(exp)) > 0.05    # (8.Counting2's sawn leaves its table on some samples: NaN in the reference too)


def test_sample_player_patch_bit_exact(golden, tmp_path):
    """12.SamplePlayer verbatim: maxiSample::load of "../../../beat2.wav" (relative to the working directory, as the patch
    writes it; the file put there is tests/golden/wav/mono.wav), getSummary() printed by setup(), playAtSpeed(0.68) -- the head
    starts on `size` (C:681), wraps on the first call and once more inside the 6000 frames."""
    import shutil
    exp = golden("dropin_examples.npz")["ex12"]
    cwd = tmp_path / "a" / "b" / "c"
    cwd.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "wav", "mono.wav"), str(tmp_path / "beat2.wav"))
    got, printed = run_dropin("12", exp.shape[0], tmp_path, cwd=str(cwd), stdout=subprocess.PIPE)
    # playAtSpeed interpolates amplitudes[1 + i] and [2 + i] for every head i < len (C:1061-1064): on the last two positions
    # before a wrap the reference reads past its vector (the golden stream holds whatever followed it on the heap, a denormal
    # here); the device reads the zero guards (mxg_smp.h).  Those samples -- found by replaying the head -- are compared to
    # nothing; every other one bit for bit.
    L, pos, beyond = 3001, 3001.0, []
    for n in range(exp.shape[0]):
        i = int(pos)
        if L - 2 <= i < L:
            beyond.append(n)
        pos = pos + (0.68 * 1.0) / (44100 // 44100)
        if int(pos) >= L:
            pos -= L
    assert 1 <= len(beyond) <= 4, beyond
    keep = np.ones(exp.shape[0], bool)
    keep[beyond] = False
    assert_bits_equal(got[keep, 0], exp[keep], "12.SamplePlayer through the drop-in header")
    assert np.all(np.abs(got[beyond, 0]) <= 1.0)
    assert np.array_equal(got[:, 0].view(np.uint64), got[:, 1].view(np.uint64))
    assert np.abs(exp).max() > 0.05
    want = " Format: 1\n Channels: 1\n SampleRate: 44100\n ByteRate: 88200\n BlockAlign: 2\n BitsPerSample: 16"
    assert want in printed, printed    # maxiSample::getSummary (C:727-733)


def test_fft_example_patch(golden, tmp_path):
    """20.FFT_example verbatim (it includes "libs/maxim.h" next to "maximilian.h"): maxiSample::play() -> maxiFFT(1024, 512,
    1024) -> the bins moved by a looping maxiEnvGen::play(1) (512 calls per frame) -> maxiIFFT::process(mags, phases) every
    sample.  The patch hands the MAGNITUDES to the inverse transform as phases too, so polToCart takes cos / sin of values up to
    a few hundred: the device's cosf / sinf against glibc's (the one documented tolerance of the inverse path);
    everything else -- forward transform, envelope, overlap-add -- is exact."""
    import shutil
    exp = golden("dropin_examples.npz")["ex20"]
    cwd = tmp_path / "a" / "b" / "c"
    cwd.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "wav", "mono.wav"), str(tmp_path / "beat2.wav"))
    got, printed = run_dropin("20", exp.shape[0], tmp_path, cwd=str(cwd), stdout=subprocess.PIPE)
    assert np.array_equal(got[:, 0].view(np.uint64), got[:, 1].view(np.uint64))
    assert np.abs(exp).max() > 0.05
    err = np.abs(got[:, 0] - exp).max()
    print("20.FFT_example: max |difference| %.3e, peak %.3f" % (err, np.abs(exp).max()))
    assert_bits_equal(got[:512, 0], exp[:512], "before the first spectrum: the empty overlap-add buffer")
    assert err <= FFT_EXAMPLE_TOL
    assert printed.count("SC: ") == exp.shape[0] // 512    # spectralCentroid printed once per frame


def test_newer_filter_and_envelope_classes_bit_exact(golden, tmp_path):
    """tests/patches/filters2_patch.cpp -- maxiSVF (with a cutoff change every 1500 samples), three maxiBiquad types, maxiDCBlocker and
    a maxiEnvGen ADSR (trigger, attack, decay, HOLD while the gate is up, release) gated by maxiOsc::square -- compiled against the
    reference (golden) and against the drop-in header: coefficients from the host libm, recurrences on the device, every sample of
    both channels bit-identical."""
    g = golden("dropin_examples.npz")
    exp, env = g["exp1"], g["exp1_env"]
    got, _ = run_dropin("p1", exp.shape[0], tmp_path)
    assert_bits_equal(got[:, 1], env, "maxiEnvGen ADSR")
    assert_bits_equal(got[:, 0], exp, "SVF -> biquads -> DC blocker")
    assert env.max() == 1.0 and np.all(env[7000:11000] == env[8000]) and 0.39 < env[8000] < 0.41   # the peak, the HOLD plateau
    assert env[14000] < 1e-3 and np.all(np.diff(env[11100:13500]) < 0)                            # the release, run to its end
    assert np.abs(exp).max() > 0.3


# ---- the reference's OWN test programs of this path (cpp/commandline/tests/{ffttest,mfcctest,svftest}), compiled verbatim -------
def test_reference_ffttest_verbatim(golden, tmp_path):
    """ffttest.cpp: sawn(maxiMap::linexp(phasor(0.2), 0, 1, 100, 5000)) -> maxiFFT(1024, 256) -> magnitudes moved up ten bins ->
    maxiIFFT(1024, 256)::process(mags, phases, maxiIFFT::fftModes::SPECTRUM) every sample.  The forward transform is exact;
    the phases (atan2f) and the inverse's polToCart (cosf / sinf) are the device's: the tolerance of the inverse path."""
    exp = golden("dropin_examples.npz")["extfft"]
    got, _ = run_dropin("tfft", exp.shape[0], tmp_path)
    assert np.array_equal(got[:, 0].view(np.uint64), got[:, 1].view(np.uint64))
    err = np.abs(got[:, 0] - exp).max()
    print("ffttest: max |difference| %.3e, peak %.3f" % (err, np.abs(exp).max()))
    assert np.abs(exp).max() > 0.3 and err <= FFT_EXAMPLE_TOL


def test_reference_mfcctest_verbatim(golden, tmp_path):
    """mfcctest.cpp: the same source -> maxiFFT(1024, 256, 1024)::process(w, maxiFFT::WITH_POLAR_CONVERSION) -> maxiMFCC(512 bins,
    256 filters, 13 coefficients)::mfcc(mags), mfccs[1] printed once per frame.  The signal is bit-identical; the printed
    coefficients (six significant digits, as `cout <<` writes them) are the reference's."""
    g = golden("dropin_examples.npz")
    exp, printed = g["extmfcc"], g["extmfcc_printed"]
    got, out = run_dropin("tmfcc", exp.shape[0], tmp_path, stdout=subprocess.PIPE)
    assert_bits_equal(got[:, 0], exp, "mfcctest's signal")
    vals = np.array([float(x) for x in out.replace("Setup", "").split()])
    assert vals.shape == printed.shape == (exp.shape[0] // 256,)
    assert np.all(np.abs(vals - printed) <= 1.5e-6 * np.abs(printed) + 1e-9), (vals, printed)
    assert np.abs(printed).max() > 0.5


def test_reference_svftest_verbatim(golden, tmp_path):
    """svftest.cpp: maxiSVF with setCutoff(40 + |sinewave(0.5)| * 500) and setResonance(phasor(0.2) * 1.2) on EVERY sample, saw(100)
    through its low-pass output.  The cutoff carries the sinewave's <= 1 ULP, tan() and the recursive filter pass it on: a
    relative tolerance like the modulated lores (tests/test_gpu_voice.py MOD_FILTER_RTOL = 1e-11), stated against the peak."""
    MOD_FILTER_RTOL = 1e-11
    exp = golden("dropin_examples.npz")["extsvf"]
    got, _ = run_dropin("tsvf", exp.shape[0], tmp_path)
    assert np.array_equal(got[:, 0].view(np.uint64), got[:, 1].view(np.uint64))
    err = np.abs(got[:, 0] - exp).max() / np.abs(exp).max()
    print("svftest: max relative difference %.3e (allowed %.1e)" % (err, MOD_FILTER_RTOL))
    assert np.abs(exp).max() > 0.05 and err <= MOD_FILTER_RTOL


# ---- round 3: the rest of the drop-in surface (include/maxiGrains.h, maxiConvolve.h, maxiSynths.h, maxiSample's trigger-driven
# players and buffer editors), each through a patch compiled both ways -------------------------------------------------------
def _run_patch(tag, frames, tmp_path):
    import shutil
    shutil.copy(os.path.join(ROOT, "tests", "golden", "wav", "mono.wav"), str(tmp_path / "mono.wav"))
    return run_dropin(tag, frames, tmp_path, cwd=str(tmp_path))


def test_granular_patch_bit_exact(golden, tmp_path):
    """tests/patches/granular_patch.cpp: maxiTimeStretch<hann>, maxiPitchShift<hamming> and maxiStretch<triangle> called once per
    sample with stepping arguments, next to a maxiOsc::noise() that shares the process-wide rand() stream with the grain schedulers'
    jitter (`randomOffset = rand() % 10`, L/maxiGrains.h:352): every sample of both channels identical to the reference's, i.e. the
    grain arithmetic, the schedulers, the rewinds at argument changes AND the order of the rand() draws."""
    g = golden("dropin_r3.npz")
    got, log = _run_patch("p2", g["exp2_l"].shape[0], tmp_path)
    assert_bits_equal(got[:, 0], g["exp2_l"], "granular patch, left")
    assert_bits_equal(got[:, 1], g["exp2_r"], "granular patch, right")
    assert np.abs(g["exp2_l"]).max() > 0.5


def test_sample_trigger_players_and_editors_patch_bit_exact(golden, tmp_path):
    """tests/patches/sampler_zx_patch.cpp: playOnZX / ...AtSpeed / ...FromOffset / ...BetweenPoints, loopSetPosOnZX, playWithPhasor
    under a phasor, playAtSpeedBetweenPointsFromPos, operator=, normalise, reset and loopRecord into a sample that is playing."""
    g = golden("dropin_r3.npz")
    got, _ = _run_patch("p3", g["exp3_l"].shape[0], tmp_path)
    assert_bits_equal(got[:, 0], g["exp3_l"], "trigger-driven players")
    assert_bits_equal(got[:, 1], g["exp3_r"], "normalise / reset / loopRecord + play")


def test_convolve_and_sampler_patch_bit_exact(golden, tmp_path):
    """tests/patches/convolve_sampler_patch.cpp: maxiConvolve::setup(file, 256, 64) + play(w) per sample (the reference's output:
    its COMPLEX-mode inverse never sees the sums, so silence -- reproduced) and an eight-slot maxiSampler driven by midiNoteOn /
    trigger / midiNoteOff / setPitch between play() calls."""
    g = golden("dropin_r3.npz")
    got, _ = _run_patch("p4", g["exp4_l"].shape[0], tmp_path)
    assert_bits_equal(got[:, 0], g["exp4_l"], "maxiConvolve")
    assert_bits_equal(got[:, 1], g["exp4_r"], "maxiSampler")
    assert np.abs(g["exp4_r"]).max() > 0.05


def test_public_members_and_value_semantics_patch_bit_exact(golden, tmp_path):
    """tests/patches/public_members_patch.cpp (round 4): every public member of maxiOsc / maxiFilter / maxiSample / maxiEnv, the classes in
    std::vector, copy construction and copy assignment in mid-flight, `maxiEnv::amplitude / holdcount / *phase` and `maxiSample::amplitudes`
    read and written from user code, `maxiFilter::cutoff` as lores() leaves it -- the same source compiled against the reference gives
    the golden stream."""
    g = golden("dropin_r4.npz")
    got, _ = _run_patch("p5", g["exp5_l"].shape[0], tmp_path)
    # channel 0 carries sinewave / coswave (within 1 ULP of glibc each) through filters: a tolerance; channel 1 (envelopes, samples) exact
    err = np.abs(got[:, 0] - g["exp5_l"]).max()
    print("public members patch, left: max |difference| %.3e on a peak of %.3f" % (err, np.abs(g["exp5_l"]).max()))
    assert err <= 1e-12 * max(1.0, np.abs(g["exp5_l"]).max())
    assert_bits_equal(got[:, 1], g["exp5_r"], "envelopes and samples: members, copies")
    assert np.abs(g["exp5_r"]).max() > 0.5


def test_refused_call_silences_that_object_only(golden, tmp_path):
    """an argument the C-ABI refuses (MXG_ERR_INVALID: maxiTimeStretch::play with ten overlapping grains -- the renderer
    holds eight -- then overlaps = 0) is NOT a device failure: one printed line per call site, that call returns silence, and every
    other unit generator plays on -- the left channel is still cpp/commandline/main.cpp's sinewave(440) stream, to the ULP."""
    frames = 12000
    got, log = run_dropin("p6", frames, tmp_path)
    exp = golden("dropin.npz")["ex01"][:frames, 0]
    assert ulp_diff(got[:, 0], exp).max() <= 1
    # (the first eight grains are fine -- one spawn per 220.5 samples -- and play as in the reference; the ninth spawn is refused, and from
    # there on, and through the overlaps = 0 calls from frame 6000, the object is silent)
    assert got[:1500, 1].any() and not got[2100:, 1].any(), "the refused granular calls return silence"
    assert "the device path is off" not in log, log
    assert 1 <= log.count("this call returns silence") <= 4, log
