"""GPU parity (-m gpu): maxiFFT / maxiMFCC batches through the C-ABI vs oracle + golden."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

# Stated tolerances (DESIGN.md "Numerics"):
PHASE_ATOL = 2e-6      # rad: device atan2f vs glibc atan2f (both fp32), |phase| <= pi
MFCC_RTOL = 1e-12      # device log() vs glibc log() through the 42-term DCT (relative to max |band|)
MFMA_RTOL = 1e-11      # the legacy dense fp64 MFMA GEMM over stored spectra (mxg_mfcc_batch method 1): 256 / 512-term sums in the matrix pipe's order
FUSED_MM_RTOL = 1e-13  # the fused kernel's matrix-pipe forms (fused_mel 2 / 3) against its vector form on the same device: measured 1.4e-15

CASES = [(1024, 1024, 1024), (1024, 256, 0), (512, 128, 512), (2048, 1024, 2048), (64, 64, 64)]


def sig_slice(sig, fs, hop):
    return sig[:fs * 4] if fs > 1024 else sig[:sig.size // (2 if hop < 1024 else 1)]


def f32bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def check_fft(f, exp_real, exp_imag, exp_mags, exp_phases, what):
    assert np.array_equal(f32bits(f.getReal().numpy()), f32bits(exp_real)), what + " real"
    assert np.array_equal(f32bits(f.getImag().numpy()), f32bits(exp_imag)), what + " imag"
    assert np.array_equal(f32bits(f.getMagnitudes().numpy()), f32bits(exp_mags)), what + " mags"
    ph = f.getPhases().numpy()
    d = np.abs(ph - exp_phases)
    d = np.minimum(d, 2 * np.pi - d)  # +pi and -pi are the same angle
    assert d.max() <= PHASE_ATOL, (what, float(d.max()))


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("fs,hop,win", CASES)
def test_fft_golden(mx, golden, fs, hop, win, generic):
    g = golden("spectral.npz")
    prev = mx.lib().mxg_tune(b"fft_generic", generic)
    try:
        f = mx.maxiFFT()
        f.setup(fs, hop, win)
        n = f.process_signal(sig_slice(g["signal"], fs, hop), want_complex=True)
        tag = "%d_%d" % (fs, hop)
        assert n == g["mags_" + tag].shape[0]
        check_fft(f, g["real_" + tag], g["imag_" + tag], g["mags_" + tag], g["phases_" + tag], tag)
    finally:
        mx.lib().mxg_tune(b"fft_generic", prev)


def test_fft_large_vs_oracle_and_unaligned(mx, port):
    rng = np.random.default_rng(41)
    nfr = 777
    sig = rng.uniform(-1, 1, 1024 * nfr + 3).astype(np.float32)
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    assert f.process_signal(sig[:1024 * nfr], want_complex=True) == nfr
    e = port.fft_stream(sig[:1024 * nfr], 1024, 1024, 1024)
    check_fft(f, e["real"], e["imag"], e["mags"], e["phases"], "large")
    # odd frame stride / unaligned base: frames start every 1025 samples from sample 3
    d = mx.DeviceBuffer.from_numpy(sig)
    n2 = 700
    f.process_frames(d.ptr + 4 * 3, 1025, n2, want_complex=True)
    frames = np.stack([sig[3 + 1025 * k: 3 + 1025 * k + 1024] for k in range(n2)])
    e = port.fft_stream(frames.reshape(-1), 1024, 1024, 1024)
    check_fft(f, e["real"], e["imag"], e["mags"], e["phases"], "unaligned")


@pytest.mark.parametrize("fs", [4096, 8192, 8, 16])
def test_fft_generic_extreme_sizes_vs_oracle(mx, port, fs):
    """The generic kernel at the ends of its range (8..8192): one LDS frame per wave, workgroup shape chosen
    from the size so the launch stays within 64 KB of LDS."""
    rng = np.random.default_rng(fs)
    nfr = 37
    sig = rng.uniform(-1, 1, fs * nfr).astype(np.float32)
    f = mx.maxiFFT()
    f.setup(fs, fs, fs)
    assert f.process_signal(sig, want_complex=True) == nfr
    e = port.fft_stream(sig, fs, fs, fs)
    check_fft(f, e["real"], e["imag"], e["mags"], e["phases"], "size%d" % fs)


def test_fft_invalid_sizes(mx):
    L = mx.lib()
    for bad in (0, 6, 1000, 16384):
        assert not L.mxg_fft_plan_create(bad, 4, bad)
    assert not L.mxg_fft_plan_create(1024, 512, 2048)   # window > fft overruns the reference's buffers
    assert not L.mxg_fft_plan_create(1024, 0, 1024)
    assert b"mxg_fft_plan_create" in L.mxg_last_error()


def test_fft_linearity_and_parseval_at_scale(mx):
    """Size-independent properties on a large batch (65 536 frames = 256 MB of input)."""
    import ctypes
    nfr = 65536
    rng = np.random.default_rng(43)
    base = rng.uniform(-1, 1, 4096 + 1024).astype(np.float32)
    # frame k = base[k % 4096 : +1024]: build by tiling so the expected spectra repeat with period 4096
    sig = np.concatenate([base[:4096]] * (nfr * 1024 // 4096 // 1)).astype(np.float32)[:nfr * 1024]
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    d = mx.DeviceBuffer.from_numpy(sig)
    f.process_frames(d, 1024, nfr)
    m = f.getMagnitudes().numpy()
    # identical frames (period 4 frames) give bit-identical spectra wherever they sit in the batch
    assert np.array_equal(f32bits(m[:4]), f32bits(m[4:8]))
    assert np.array_equal(f32bits(m[:4]), f32bits(m[nfr - 4:]))
    assert np.array_equal(f32bits(m.reshape(-1, 4, 512)[::997]), f32bits(np.broadcast_to(m[:4], (len(m.reshape(-1, 4, 512)[::997]), 4, 512))))
    # Parseval on the windowed frame (bins 1..511 of the packed real transform), loose fp32 bound
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1023)
    xw = (sig[:1024].astype(np.float64) * win.astype(np.float32))
    X = np.fft.rfft(xw)
    # the fp32 recurrence twiddles of the reference are only ~1e-4 accurate: loose bound by design
    assert np.allclose(m[0][1:], np.abs(X[1:512]), rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("nf,nc", [(42, 13), (256, 13), (40, 20)])
def test_mfcc_golden(mx, golden, nf, nc):
    g = golden("spectral.npz")
    mags = g["mags_1024_1024"]
    m = mx.maxiMFCC()
    m.setup(512, nf, nc, 20.0, 20000.0)
    out = m.mfcc(mx.DeviceBuffer.from_numpy(mags), want_bands=True).numpy()
    emel, emf = g["melbands_%d_%d" % (nf, nc)], g["mfcc_%d_%d" % (nf, nc)]
    bands = m.melBands.numpy()
    scale = np.abs(emel).max()
    assert np.abs(bands - emel).max() <= MFCC_RTOL * scale
    assert np.abs(out - emf).max() <= MFCC_RTOL * scale
    # zero bands stay exactly zero (the `> 0.000001 ? log : 0` branch)
    assert np.array_equal(bands == 0.0, emel == 0.0)


@pytest.fixture(params=[0, 1, 2], ids=["waves8", "waves16", "waves12"])
def fused_form(mx, request):
    """The forms of the fused kernel: 8 waves per CU (8 frames x 8 slots per wavefront, two frames in flight; knob fused_layout 1),
    16 waves per CU (4 frames x 16 slots; only when no magnitudes are requested) and 12 waves per CU (one frame in flight,
    fused_layout 2; only without the full magnitude rows)."""
    L = mx.lib()
    prev16 = L.mxg_tune(b"fused_waves16", 1 if request.param == 1 else 0)
    prevl = L.mxg_tune(b"fused_layout", 2 if request.param == 2 else 1)
    prevm = L.mxg_tune(b"fused_mel", 1)  # the vector form: what these layouts are layouts OF (mfcc bits = mxg_mfcc_batch's DCT order);
    yield request.param                  # the matrix-pipe forms have their own tests below
    L.mxg_tune(b"fused_waves16", prev16)
    L.mxg_tune(b"fused_layout", prevl)
    L.mxg_tune(b"fused_mel", prevm)


@pytest.mark.parametrize("nf,nc,nfr,off", [(42, 13, 1003, 0), (40, 20, 77, 0), (64, 13, 8, 0), (42, 13, 250, 3), (13, 5, 1, 0),
                                           (42, 13, 16 * 4 * 256 + 5, 0)])
def test_fused_fft_mfcc_vs_two_kernels_and_oracle(mx, port, fused_form, nf, nc, nfr, off):
    """mxg_fft_mfcc_batch (one kernel, magnitudes kept in LDS): magnitudes and raw band sums bit-identical to the
    separate kernels (and the magnitudes to the oracle's), mfcc within the log tolerance -- for ragged frame counts
    (not a multiple of the 8-frame group), an unaligned / odd-stride signal, 40x20 and 64-filter banks, with and
    without the optional outputs."""
    rng = np.random.default_rng(nf * 100 + nfr)
    stride = 1024 if off == 0 else 1025
    sig = (rng.uniform(-1, 1, stride * nfr + 8) * np.repeat(10.0 ** rng.uniform(-4, 0, nfr + 1), stride)[:stride * nfr + 8]
           ).astype(np.float32)
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC()
    m.setup(512, nf, nc, 20.0, 20000.0)
    base = d.ptr + 4 * off
    out = m.mfcc_of_frames(f, base, nfr, frame_stride=stride, want_mags=True, want_bands=True).numpy()
    mags, raw, bands = m.mags.numpy(), m.melraw.numpy(), m.melBands.numpy()
    f.process_frames(base, stride, nfr)
    mags2 = f.getMagnitudes()
    assert np.array_equal(f32bits(mags), f32bits(mags2.numpy())), "fused magnitudes != mxg_fft_batch magnitudes"
    out2 = m.mfcc(mags2, want_bands=True).numpy()
    assert_bits_equal(raw, m.melraw.numpy(), "raw band sums: fused vs mxg_mfcc_batch")
    assert_bits_equal(bands, m.melBands.numpy(), "melBands: same device log on the same sums")
    assert_bits_equal(out, out2, "mfcc: same DCT order on the same bands")
    frames = np.stack([sig[off + stride * k: off + stride * k + 1024] for k in range(nfr)])
    e = port.fft_stream(frames.reshape(-1), 1024, 1024, 1024, want=("mags",))["mags"]
    assert np.array_equal(f32bits(mags), f32bits(e)), "fused magnitudes != oracle"
    emel, emf = port.mfcc(e, nf, nc, 20.0, 20000.0)
    assert np.abs(out - emf).max() <= MFCC_RTOL * max(np.abs(emel).max(), 1e-300)
    # mfcc-only launch (no optional outputs: the half-spectrum variant of the kernel): identical coefficients
    out3 = m.mfcc_of_frames(f, base, nfr, frame_stride=stride).numpy()
    assert_bits_equal(out3, out, "mfcc-only variant")
    # bands without magnitudes (the 16-wave form when enabled): raw sums and bands still the same bits
    out4 = m.mfcc_of_frames(f, base, nfr, frame_stride=stride, want_bands=True).numpy()
    assert_bits_equal(out4, out, "mfcc, bands-only variant")
    assert_bits_equal(m.melraw.numpy(), raw, "raw band sums, bands-only variant")
    assert_bits_equal(m.melBands.numpy(), bands, "melBands, bands-only variant")


def test_fused_non_finite_frames(mx, port, fused_form):
    """Frames holding Inf / NaN samples.  The reference multiplies every sample by a twiddle of (1, 0) in the first stage, so one such
    sample becomes NaN in every bin (0 * Inf), the band sums are NaN, `mb > 0.000001` is false and the frame's mfcc are all zero.  The
    half-spectrum kernel skips those products (round3_s1) and restores exactly this behind its square-root range test: magnitudes
    NaN where the oracle's are, mfcc the oracle's bits, and the finite frames around them untouched."""
    rng = np.random.default_rng(99)
    nfr = 40
    sig = rng.uniform(-1, 1, 1024 * nfr).astype(np.float32)
    sig[3 * 1024 + 100] = np.nan
    sig[8 * 1024 + 0] = np.inf           # times window[0] = 0: NaN
    sig[9 * 1024 + 513] = -np.inf
    sig[17 * 1024 + 1023] = np.inf
    sig[33 * 1024 + 7] = np.nan
    sig[33 * 1024 + 8] = np.inf
    bad = [3, 8, 9, 17, 33]
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
    full = m.mfcc_of_frames(f, d.ptr, nfr, want_mags=True, want_bands=True).numpy()   # all 512 bins: the generic first round
    mags, raw_full = m.mags.numpy(), m.melraw.numpy()
    half = m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True).numpy()                   # the half-spectrum kernel
    raw_half = m.melraw.numpy()
    with np.errstate(all="ignore"):
        e = port.fft_stream(sig, 1024, 1024, 1024, want=("mags",))["mags"]
        emel, emf = port.mfcc(e, 42, 13, 20.0, 20000.0)
    assert np.isnan(e[bad]).all() and np.isfinite(np.delete(e, bad, axis=0)).all(), "the oracle's picture of these frames"
    assert np.array_equal(np.isnan(mags), np.isnan(e))
    good = np.setdiff1d(np.arange(nfr), bad)
    assert np.array_equal(f32bits(mags[good]), f32bits(e[good]))
    assert np.array_equal(np.isnan(raw_half), np.isnan(raw_full)) and np.isnan(raw_half[bad]).sum(axis=1).min() >= 40  # (an empty filter stays 0)
    assert_bits_equal(raw_half[good], raw_full[good], "band sums of the finite frames")
    assert_bits_equal(half, full, "mfcc: half-spectrum kernel vs full kernel")
    assert np.array_equal(half[bad], np.zeros((len(bad), 13))), "NaN band sums take the `: 0` branch, as in the reference"
    assert np.array_equal(emf[bad], np.zeros((len(bad), 13)))
    assert np.abs(half[good] - emf[good]).max() <= MFCC_RTOL * max(np.abs(emel[good]).max(), 1e-300)


MM_BAND_RTOL = 4e-15   # matrix-pipe mel contraction (fused_mel 3): fused multiply-adds in the matrix pipe's order, x the frame's largest band (measured 2.2e-16)
MM_DCT_RTOL = 1e-13    # matrix-pipe DCT (fused_mel 2 and 3) against the sequential DCT on the SAME logs, x the largest |band log|


@pytest.mark.parametrize("nf,nc,nfr,off", [(42, 13, 1003, 0), (42, 13, 250, 3), (13, 5, 1, 0), (48, 16, 77, 0), (26, 13, 100, 0),
                                           (40, 20, 77, 0), (64, 13, 9, 0), (42, 13, 8 * 8 * 256 + 13, 0)])
@pytest.mark.parametrize("mel", [2, 3])
def test_fused_matrix_pipe_forms(mx, port, mel, nf, nc, nfr, off):
    """Knob fused_mel: 2 = the sparse walk (band sums bit-exact) with the logs in matrix layout and the DCT on
    v_mfma_f64_4x4x4; 3 = the banded mel contraction on the matrix pipe as well.  Against the vector form (fused_mel 1) on the
    same device and against the oracle; banks the quad tables do not cover (more than 48 filters / 16 coefficients) fall back to
    the vector form and must give its bits."""
    rng = np.random.default_rng(nf * 100 + nfr + mel)
    stride = 1024 if off == 0 else 1025
    sig = (rng.uniform(-1, 1, stride * nfr + 8) * np.repeat(10.0 ** rng.uniform(-4, 0, nfr + 1), stride)[:stride * nfr + 8]
           ).astype(np.float32)
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC(); m.setup(512, nf, nc, 20.0, 20000.0)
    base = d.ptr + 4 * off
    L = mx.lib()
    prev = L.mxg_tune(b"fused_mel", 1)
    try:
        ref = m.mfcc_of_frames(f, base, nfr, frame_stride=stride, want_bands=True).numpy()
        raw1, bands1 = m.melraw.numpy(), m.melBands.numpy()
        L.mxg_tune(b"fused_mel", mel)
        out = m.mfcc_of_frames(f, base, nfr, frame_stride=stride, want_bands=True).numpy()
        raw, bands = m.melraw.numpy(), m.melBands.numpy()
        out_only = m.mfcc_of_frames(f, base, nfr, frame_stride=stride).numpy()
    finally:
        L.mxg_tune(b"fused_mel", prev)
    assert_bits_equal(out_only, out, "mfcc-only launch vs launch with the band outputs")
    covered = nf <= 48 and nc <= 16
    top = max(np.abs(bands1).max(), 1e-300)
    if not covered:
        assert_bits_equal(out, ref, "bank outside the quad tables: the vector form's bits")
        assert_bits_equal(raw, raw1, "band sums")
        return
    if mel == 2:
        assert_bits_equal(raw, raw1, "fused_mel 2: the walk's band sums")
        assert_bits_equal(bands, bands1, "fused_mel 2: the same log on the same sums")
        assert np.abs(out - ref).max() <= MM_DCT_RTOL * top
    else:
        rowmax = np.maximum(np.abs(raw1).max(axis=1, keepdims=True), 1e-300)
        assert (np.abs(raw - raw1) / rowmax).max() <= MM_BAND_RTOL
        assert np.array_equal(raw == 0.0, raw1 == 0.0), "empty filters stay exactly zero"
        assert np.abs(out - ref).max() <= FUSED_MM_RTOL * max(1.0, top)
    frames = np.stack([sig[off + stride * k: off + stride * k + 1024] for k in range(nfr)])
    e = port.fft_stream(frames.reshape(-1), 1024, 1024, 1024, want=("mags",))["mags"]
    emel, emf = port.mfcc(e, nf, nc, 20.0, 20000.0)
    assert np.abs(out - emf).max() <= MFCC_RTOL * max(np.abs(emel).max(), 1.0)  # (both forms: the device log's distance from glibc's)


def test_fused_matrix_pipe_band_at_the_row_end(mx, port):
    """a bank whose last quad band is moved down to END at the last float of the 260-float magnitude row (setup(512, 46, 13,
    20, 21900): pair 5 reads bins 180 .. 259) multiplies floats 257 .. 259 -- which no post-pass stores -- with zero weights; they are
    zeroed at kernel start now, so whatever the LDS held before (here: a launch of another kernel family in between) cannot turn a
    band sum into NaN.  Matrix form against the vector form and the oracle."""
    rng = np.random.default_rng(46)
    nfr = 300
    sig = rng.uniform(-1, 1, 1024 * nfr).astype(np.float32)
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC(); m.setup(512, 46, 13, 20.0, 21900.0)
    L = mx.lib()
    prev = L.mxg_tune(b"fused_mel", 1)
    try:
        ref = m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True).numpy()
        raw1 = m.melraw.numpy()
        for rep in range(3):
            # dirty the LDS of every CU with NaN bit patterns: the voice bank's mixdown form fills 110 KB per CU with its samples, and
            # lores with the cutoff clamped to sr has r = 0 / 0 (src/maximilian.cpp:461) -- every sample NaN, in the reference too
            junk = mx.maxiVoiceBank(65536)
            junk.render_mix(0, np.full(65536, 440.0), 1e9, 1.0, np.ones(64, np.int32), np.full(65536, 0.5), 64, store=False)
            L.mxg_tune(b"fused_mel", 3)
            out = m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True).numpy()
            raw = m.melraw.numpy()
            assert np.isfinite(raw).all() and np.isfinite(out).all(), "rep %d" % rep
            rowmax = np.maximum(np.abs(raw1).max(axis=1, keepdims=True), 1e-300)
            assert (np.abs(raw - raw1) / rowmax).max() <= MM_BAND_RTOL
            assert np.abs(out - ref).max() <= FUSED_MM_RTOL * max(1.0, np.abs(m.melBands.numpy()).max())
    finally:
        L.mxg_tune(b"fused_mel", prev)
    e = port.fft_stream(sig, 1024, 1024, 1024, want=("mags",))["mags"]
    emel, emf = port.mfcc(e, 46, 13, 20.0, 21900.0)
    assert np.abs(out - emf).max() <= MFCC_RTOL * max(np.abs(emel).max(), 1.0)


def test_fused_automatic_form(mx, port):
    """fused_mel 0 (the default): a launch that asks for the band sums gets the sparse walk's (bit for bit the separate kernels' and the
    reference's sequential sums); a launch that asks for the coefficients only takes the matrix pipe for the mel contraction too -- its
    mfcc within the matrix form's stated distance of the first launch's, and both within the device log's tolerance of the oracle."""
    rng = np.random.default_rng(2024)
    nfr = 777
    sig = (rng.uniform(-1, 1, 1024 * nfr) * np.repeat(10.0 ** rng.uniform(-4, 0, nfr), 1024)).astype(np.float32)
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
    L = mx.lib()
    prev = L.mxg_tune(b"fused_mel", 0)
    try:
        with_bands = m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True).numpy()
        raw = m.melraw.numpy()
        only = m.mfcc_of_frames(f, d.ptr, nfr).numpy()
        L.mxg_tune(b"fused_mel", 1)
        m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True)
        raw1 = m.melraw.numpy()
    finally:
        L.mxg_tune(b"fused_mel", prev)
    assert_bits_equal(raw, raw1, "band sums of the automatic form when they are requested")
    e = port.fft_stream(sig, 1024, 1024, 1024, want=("mags",))["mags"]
    emel, emf = port.mfcc(e, 42, 13, 20.0, 20000.0)
    top = max(np.abs(emel).max(), 1.0)
    assert np.abs(only - with_bands).max() <= FUSED_MM_RTOL * top
    assert np.abs(with_bands - emf).max() <= MFCC_RTOL * top
    assert np.abs(only - emf).max() <= MFCC_RTOL * top


@pytest.mark.parametrize("mel", [2, 3])
def test_fused_matrix_pipe_non_finite_and_silent_frames(mx, port, mel):
    """NaN / Inf frames (every band NaN -> the reference's `: 0` branch -> mfcc 0) must not leak into the other frames of their
    8-frame group (a matrix block multiplies the frames of one half column by column), and silent frames stay exactly zero."""
    rng = np.random.default_rng(7 + mel)
    nfr = 40
    sig = rng.uniform(-1, 1, 1024 * nfr).astype(np.float32)
    sig[3 * 1024 + 100] = np.nan
    sig[9 * 1024 + 513] = -np.inf
    sig[17 * 1024 + 1023] = np.inf
    sig[20 * 1024: 22 * 1024] = 0.0
    bad, silent = [3, 9, 17], [20, 21]
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
    L = mx.lib()
    prev = L.mxg_tune(b"fused_mel", 1)
    try:
        ref = m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True).numpy()
        raw1 = m.melraw.numpy()
        L.mxg_tune(b"fused_mel", mel)
        out = m.mfcc_of_frames(f, d.ptr, nfr, want_bands=True).numpy()
        raw = m.melraw.numpy()
    finally:
        L.mxg_tune(b"fused_mel", prev)
    # (filter 0 has no support: the walk never visits it and leaves 0, the matrix product multiplies its zero weights with the NaN
    # magnitudes like the reference's dense loop does, L/maxiMFCC.cpp:52-60 -- NaN; both take the `: 0` branch of :63)
    assert np.array_equal(np.isnan(raw[:, 1:]), np.isnan(raw1[:, 1:])) and np.isfinite(np.delete(raw, bad, axis=0)).all()
    assert np.array_equal(out[bad], np.zeros((len(bad), 13))) and np.array_equal(out[silent], np.zeros((2, 13)))
    good = np.setdiff1d(np.arange(nfr), bad)
    assert np.isfinite(out[good]).all()
    assert np.abs(out[good] - ref[good]).max() <= (MM_DCT_RTOL if mel == 2 else FUSED_MM_RTOL) * 30.0


def test_survey_mfcc_anchor_on_device(mx, port):
    """SURVEY 8(c)'s anchor (third frame of sawn(220), fft 1024/512/1024, mfcc 512/42/13/20/20000) through the device
    path: streaming maxiFFT + maxiMFCC, within the log tolerance of the recorded glibc values."""
    sig, _, _ = port.osc(10, np.array([220.0]), 1024 * 3)
    f = mx.maxiFFT()
    f.setup(1024, 512, 1024)
    assert f.process_signal(sig[:, 0].astype(np.float32)) >= 3
    m = mx.maxiMFCC()
    m.setup(512, 42, 13, 20.0, 20000.0)
    out = m.mfcc(f.getMagnitudes()).numpy()
    anchor = {0: 0.40082902638055068, 1: -0.31312622651102895, 12: 0.32965843633589265}
    for i, v in anchor.items():
        assert abs(out[2, i] - v) <= 1e-13


def test_fused_fft_mfcc_tiny_and_silent_frames(mx, port, fused_form):
    """Magnitudes around the sqrt's small-input branch (power < 2^-96) and all-zero frames: bit-exact magnitudes,
    zero bands stay exactly zero."""
    rng = np.random.default_rng(5)
    nfr = 24
    sig = np.zeros((nfr, 1024), np.float32)
    for k in range(1, nfr):
        sig[k] = (rng.uniform(-1, 1, 1024) * 10.0 ** (-2.5 * k)).astype(np.float32)   # down to denormal spectra
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC()
    m.setup(512, 42, 13, 20.0, 20000.0)
    out = m.mfcc_of_frames(f, mx.DeviceBuffer.from_numpy(sig), nfr, want_mags=True, want_bands=True).numpy()
    e = port.fft_stream(sig.reshape(-1), 1024, 1024, 1024, want=("mags",))["mags"]
    assert np.array_equal(f32bits(m.mags.numpy()), f32bits(e))
    emel, emf = port.mfcc(e, 42, 13, 20.0, 20000.0)
    assert np.array_equal(m.melBands.numpy() == 0.0, emel == 0.0)
    assert np.abs(out - emf).max() <= MFCC_RTOL * max(np.abs(emel).max(), 1.0)
    assert (out[0] == 0).all()
    out2 = m.mfcc_of_frames(f, mx.DeviceBuffer.from_numpy(sig), nfr, want_bands=True).numpy()   # no magnitudes out
    assert_bits_equal(out2, out, "mfcc without magnitudes")


def test_fused_fft_mfcc_rejects_what_it_cannot_do(mx):
    L = mx.lib()
    f = mx.maxiFFT(); f.setup(512, 512, 512)
    m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
    b = mx.DeviceBuffer(4096, np.float32)
    o = mx.DeviceBuffer((4, 13))
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, b.ptr, 512, 4, None, None, None, o.ptr, None) == -1   # not a 1024-point plan
    f.setup(1024, 1024, 1024)
    m.setup(512, 256, 13, 20.0, 20000.0)   # mfcctest's bank: 256 filters -> two-kernel path
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, b.ptr, 1024, 4, None, None, None, o.ptr, None) == -1
    m.setup(512, 42, 13, 20.0, 20000.0)
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, b.ptr, 1024, 0, None, None, None, o.ptr, None) == 0     # empty batch
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, None, 1024, 4, None, None, None, o.ptr, None) == -1


def test_mfcc_melraw_bit_exact_and_mfma(mx, port):
    rng = np.random.default_rng(47)
    nfr = 1000   # not a multiple of 64 or 16: ragged tail
    sig = rng.uniform(-1, 1, 1024 * nfr).astype(np.float32)
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    f.process_signal(sig)
    mags = f.getMagnitudes()
    hm = mags.numpy()
    m = mx.maxiMFCC()
    m.setup(512, 42, 13, 20.0, 20000.0)
    W, D, used = m.tables()
    Wm = W.reshape(512, 42)
    # dense sequential sums in the reference's bin order (numpy cumulative => same order)
    raw_exp = np.zeros((nfr, 42))
    for b in range(512):
        raw_exp += Wm[b][None, :] * hm[:, b:b + 1].astype(np.float64)
    out = m.mfcc(mags, want_bands=True).numpy()
    assert_bits_equal(m.melraw.numpy(), raw_exp, "melraw (sparse in-order == dense in-order)")
    emel, emf = port.mfcc(hm, 42, 13, 20.0, 20000.0)
    scale = np.abs(emel).max()
    assert np.abs(out - emf).max() <= MFCC_RTOL * scale
    # method 2 = the filter-major tile kernel (fallback path): same bits
    out3 = m.mfcc(mags, method=2, want_bands=True).numpy()
    assert_bits_equal(m.melraw.numpy(), raw_exp, "melraw (tile kernel)")
    assert_bits_equal(out3, out, "mfcc stream kernel == tile kernel")
    # MFMA method: tolerance on the band sums as well
    out2 = m.mfcc(mags, method=1, want_bands=True).numpy()
    raw2 = m.melraw.numpy()
    assert np.abs(raw2 - raw_exp).max() <= MFMA_RTOL * np.abs(raw_exp).max()
    assert np.abs(out2 - emf).max() <= MFMA_RTOL * scale * 10


@pytest.mark.parametrize("nf,nc,nfr,fullk,unaligned", [(42, 13, 1000, 0, False), (42, 13, 257, 1, False), (256, 13, 700, 0, False),
                                                        (64, 20, 64, 0, False), (16, 5, 1, 1, False), (42, 13, 300, 0, True)])
def test_mfma_gemm_vs_exact(mx, port, nf, nc, nfr, fullk, unaligned):
    """Method 1 (dense fp64 MFMA contraction, tiled GEMM kernel): band sums and mfcc against the exact sparse path --
    mfcctest's 512/256/13 bank (filter groups of 64 + the log/DCT kernel), K trimmed to the weighted bins or all 512,
    ragged frame counts, and the single-wave kernel for spectra that are not 16-byte aligned."""
    rng = np.random.default_rng(nf + nfr)
    mags = (np.abs(rng.standard_normal((nfr + 1, 512))) * 10.0 ** rng.uniform(-3, 1.5, (nfr + 1, 1))).astype(np.float32)
    d = mx.DeviceBuffer.from_numpy(mags.reshape(-1))
    base = d.ptr + (4 * 3 if unaligned else 0)      # 12-byte offset: rows of 512 floats starting at element 3
    hm = mags.reshape(-1)[3 if unaligned else 0:][:nfr * 512].reshape(nfr, 512)
    m = mx.maxiMFCC()
    m.setup(512, nf, nc, 20.0, 20000.0)
    prev = mx.lib().mxg_tune(b"mfcc_mfma_fullk", fullk)
    try:
        exact = m.mfcc(base, nframes=nfr, want_bands=True).numpy()
        rawx = m.melraw.numpy()
        dense = m.mfcc(base, nframes=nfr, method=1, want_bands=True).numpy()
        rawd, bandsd = m.melraw.numpy(), m.melBands.numpy()
    finally:
        mx.lib().mxg_tune(b"mfcc_mfma_fullk", prev)
    rel = (np.abs(rawd - rawx).max(axis=1) / np.maximum(np.abs(rawx).max(axis=1), 1e-300)).max()
    assert rel <= 1e-13, rel
    emel, emf = port.mfcc(hm, nf, nc, 20.0, 20000.0)
    scale = max(np.abs(emel).max(), 1.0)
    assert np.abs(bandsd - emel).max() <= MFMA_RTOL * scale
    assert np.abs(dense - emf).max() <= MFMA_RTOL * scale
    assert np.abs(dense - exact).max() <= MFMA_RTOL * scale
    # without the optional outputs (the wide bank then uses library scratch for the raw sums): same coefficients
    dense2 = m.mfcc(base, nframes=nfr, method=1).numpy()
    assert_bits_equal(dense2, dense, "mfma method with / without optional outputs")


@pytest.mark.parametrize("fftSize", [1024, 64, 8])
def test_fft_features(mx, port, fftSize):
    """magsToDB / spectralFlatness / spectralCentroid (L/fft.cpp:526-534, L/maxiFFT.cpp:113-132).
    centroid: float + - * / only, sequential over bins -> bit-exact.  dB: one device log10f vs
    glibc's, times 20 -> <= 4 ULP of the float result (measured: 3).  flatness: a float sum of `bins` device logf, then expf -> the
    1-ULP-per-term differences random-walk; bound 4e-6 relative (measured ~1e-6)."""
    rng = np.random.default_rng(fftSize)
    bins, n = fftSize // 2, 333
    m = np.abs(rng.normal(0, 30, (n, bins))).astype(np.float32)
    m[5] = 0                      # all-zero frame: flatness 0 / centroid 0 branches
    m[6, ::3] = 0                 # zeros skipped by the log sum
    m[7, : bins // 2] = 5e-7      # below the dB floor
    m[8] = -m[8]                  # negative magnitudes: fabs in the centroid, NaN in the flatness
    f = mx.maxiFFT()
    f.setup(fftSize, fftSize // 2, fftSize)
    dm = mx.DeviceBuffer.from_numpy(m)
    eflat, ecen = port.fft_features(m, fftSize)
    edb = port.fft_to_db(m)
    assert_bits_equal(f.spectralCentroid(dm).numpy(), ecen, "centroid")
    db = f.magsToDB(dm).numpy()
    assert np.array_equal(db == 0, edb == 0)
    assert np.abs(db.view(np.int32).astype(np.int64) - edb.view(np.int32).astype(np.int64)).max() <= 4  # ULPs (all >= 0)
    flat = f.spectralFlatness(dm).numpy()
    assert np.array_equal(np.isnan(flat), np.isnan(eflat)) and np.isnan(eflat[8])
    ok = ~np.isnan(eflat)
    np.testing.assert_allclose(flat[ok], eflat[ok], rtol=4e-6, atol=0)
    assert flat[5] == 0 and ecen[5] == 0


IFFT_CASES = [(1024, 512, 0), (1024, 256, 1024), (64, 16, 48), (8, 8, 0), (2048, 512, 0), (4096, 1024, 0)]


@pytest.mark.parametrize("fs,hop,win", IFFT_CASES)
def test_ifft_zero_phase_bit_exact(mx, port, fs, hop, win):
    """maxiIFFT with phases == 0: cos = 1, sin = 0 exactly on every libm, so the whole path (full-size
    complex inverse FFT with replayed fp32 twiddles, /n, window, overlap-add order, buffer carry across two
    calls) must be bit-identical to the reference."""
    rng = np.random.default_rng(fs + hop)
    nf = 11
    m = np.abs(rng.normal(0, 3, (nf, fs // 2))).astype(np.float32)
    ph = np.zeros_like(m)
    f = mx.maxiIFFT()
    f.setup(fs, hop, win)
    o1 = f.process_frames(m[:4], ph[:4], keep_ifft=True).numpy()
    io1 = f.ifftOut.numpy()
    o2 = f.process_frames(m[4:], ph[4:]).numpy()
    e1, eio1, buf = port.ifft_stream(m[:4], ph[:4], fs, hop, win)
    e2, _, buf2 = port.ifft_stream(m[4:], ph[4:], fs, hop, win, buffer=buf)
    assert np.array_equal(io1.view(np.uint32), eio1.view(np.uint32)), "ifftOut"
    assert np.array_equal(o1.view(np.uint32), e1.view(np.uint32)), "signal, first call"
    assert np.array_equal(o2.view(np.uint32), e2.view(np.uint32)), "signal, carried buffer"
    assert np.array_equal(f.buffer.numpy().view(np.uint32), buf2.view(np.uint32)), "buffer state"


@pytest.mark.parametrize("stream", [2, 1, 0])
@pytest.mark.parametrize("fs,hop,win,nf", [(1024, 256, 0, 700), (1024, 128, 1024, 300), (1024, 1024, 0, 150), (256, 96, 200, 500),
                                           (2048, 512, 0, 130), (8192, 2048, 0, 9)])
def test_ifft_many_frames_both_forms_bit_exact(mx, port, fs, hop, win, nf, stream):
    """Zero phases again (everything exact), but enough frames that the streaming kernel cuts them into chunks whose hop buffer is
    rebuilt from the frames before them -- against the reference's one sequential stream, in two calls with the buffer carried;
    and the two-kernel form (knob ifft_stream 0; 8192 points always take it; 1 = the default mix) on the same data."""
    rng = np.random.default_rng(fs + hop + nf)
    m = np.abs(rng.normal(0, 3, (nf, fs // 2))).astype(np.float32)
    ph = np.zeros_like(m)
    cut = nf // 3
    prev = mx.lib().mxg_tune(b"ifft_stream", stream)
    try:
        f = mx.maxiIFFT()
        f.setup(fs, hop, win)
        o1 = f.process_frames(m[:cut], ph[:cut]).numpy()
        o2 = f.process_frames(m[cut:], ph[cut:], keep_ifft=True).numpy()
        io2 = f.ifftOut.numpy()
    finally:
        mx.lib().mxg_tune(b"ifft_stream", prev)
    e1, _, buf = port.ifft_stream(m[:cut], ph[:cut], fs, hop, win)
    e2, eio2, buf2 = port.ifft_stream(m[cut:], ph[cut:], fs, hop, win, buffer=buf)
    assert np.array_equal(o1.view(np.uint32), e1.view(np.uint32)), "signal, first call"
    assert np.array_equal(o2.view(np.uint32), e2.view(np.uint32)), "signal, carried buffer"
    assert np.array_equal(io2.view(np.uint32), eio2.view(np.uint32)), "ifftOut"
    assert np.array_equal(f.buffer.numpy().view(np.uint32), buf2.view(np.uint32)), "buffer state"


@pytest.mark.parametrize("fs,hop,win", IFFT_CASES[:4])
def test_ifft_random_phase(mx, port, fs, hop, win):
    """Random phases: polToCart uses the float cos/sin of the device (L/fft.cpp:597-598 resolves to cosf/sinf),
    which may differ from glibc's by an ULP; the transform is linear, so the output moves by at most
    ~2^-23 * sum|mag| / n per ULP.  Bound: 4e-7 * (sum of magnitudes of a frame) / n * overlap."""
    rng = np.random.default_rng(fs * 3 + hop)
    nf = 9
    m = np.abs(rng.normal(0, 3, (nf, fs // 2))).astype(np.float32)
    ph = rng.uniform(-np.pi, np.pi, (nf, fs // 2)).astype(np.float32)
    f = mx.maxiIFFT()
    f.setup(fs, hop, win)
    o = f.process_frames(m, ph).numpy()
    e, _, buf = port.ifft_stream(m, ph, fs, hop, win)
    tol = 4e-7 * m.sum(axis=1).max() / fs * (fs // hop)
    assert np.abs(o - e).max() <= tol
    assert np.abs(f.buffer.numpy() - buf).max() <= tol
    assert np.abs(e).max() > 100 * tol


def test_fft_ifft_round_trip(mx):
    """ffttest.cpp's chain maxiFFT(1024, hop) -> maxiIFFT(1024, hop) as a size-independent property: with Hann
    analysis x Hann synthesis at hop = N/4 the overlap-added w^2 is constant, and the inverse (positive
    frequencies only, real part) returns half of it, so the output is the delayed input times gain/2 -- up to
    the reference's packing of DC and Nyquist into bin 0 (L/fft.cpp:274-275), which the inverse treats as an
    ordinary bin: a few % of a white signal.  Checked: correlation and RMS error, not samples."""
    fs, hop = 1024, 256
    rng = np.random.default_rng(5)
    sig = rng.uniform(-0.5, 0.5, hop * 64).astype(np.float32)
    fwd = mx.maxiFFT()
    fwd.setup(fs, hop, fs)
    n = fwd.process_signal(sig)
    inv = mx.maxiIFFT()
    inv.setup(fs, hop, fs)
    y = inv.process_frames(fwd.getMagnitudes(), fwd.getPhases()).numpy()
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(fs) / (fs - 1))).astype(np.float64)
    gain = sum((w[j * hop:(j + 1) * hop] ** 2) for j in range(fs // hop))       # overlap-added w^2 per hop phase
    delay = fs - hop
    k0 = fs // hop                                                              # skip the warm-up frames
    got = y[k0 * hop:(n - 1) * hop].astype(np.float64)
    ref = sig[k0 * hop - delay:(n - 1) * hop - delay].astype(np.float64)
    exp = ref * np.tile(gain, got.size // hop) * 0.5
    assert np.corrcoef(got, exp)[0, 1] > 0.998
    assert np.sqrt(np.mean((got - exp) ** 2)) < 0.05 * np.sqrt(np.mean(exp ** 2))


@pytest.mark.parametrize("fs,hop,win", [(1024, 512, 0), (1024, 256, 1024), (64, 16, 48)])
def test_ifft_golden(mx, golden, fs, hop, win):
    """Against the reference's own values (tests/golden/ifft.npz): a third of the phases are exactly 0, the
    rest random, so the bound is the random-phase one; the zero-phase-only case is bit-exact above."""
    g = golden("ifft.npz")
    tag = "%d_%d_%d" % (fs, hop, win)
    m, ph = g["mags_" + tag], g["phases_" + tag]
    f = mx.maxiIFFT()
    f.setup(fs, hop, win)
    o = np.concatenate([f.process_frames(m[:3], ph[:3]).numpy(), f.process_frames(m[3:], ph[3:]).numpy()])
    tol = 4e-7 * m.sum(axis=1).max() / fs * (fs // hop)
    assert np.abs(o - g["signal_" + tag]).max() <= tol
    assert np.abs(f.buffer.numpy() - g["buffer_" + tag]).max() <= tol


@pytest.mark.parametrize("nf,nc,nfr", [(42, 13, 1000), (256, 13, 77), (40, 20, 513), (42, 13, 1)])
def test_mfcc_tiled_equals_row_loads(mx, nf, nc, nfr):
    """K7a-t (LDS-staged tiles, default) and K7a (per-lane row loads) do the same additions in the same
    order: melraw, melbands and mfcc must be bit-identical, for ragged frame counts and every (S, NC)."""
    rng = np.random.default_rng(nf + nfr)
    mags = mx.DeviceBuffer.from_numpy(np.abs(rng.normal(0, 20, (nfr, 512))).astype(np.float32))
    m = mx.maxiMFCC()
    m.setup(512, nf, nc, 20.0, 20000.0)
    L = mx.lib()
    res = []
    for tiled in (1, 0):
        prev = L.mxg_tune(b"mfcc_tiled", tiled)
        try:
            out = m.mfcc(mags, want_bands=True).numpy()
            res.append((out, m.melraw.numpy(), m.melBands.numpy()))
        finally:
            L.mxg_tune(b"mfcc_tiled", prev)
    for a, b, what in zip(res[0], res[1], ("mfcc", "melraw", "melbands")):
        assert_bits_equal(a, b, what)
    assert np.isfinite(res[0][0]).all() and np.abs(res[0][0]).max() > 0


# ---- tolerance mode of the fused kernel (knob fft_exact = 0): true radix-8 butterflies, correctly rounded twiddles, FMAs, hardware
# sqrt and log2.  Reordered arithmetic => stated tolerances instead of bits (north_star: "a stated fp tolerance for the FFT path").
# What the tolerance is made of: the REFERENCE's transform is itself only accurate to ~1e-4 of a frame's peak -- its twiddles come
# from fp32 recurrences that drift (L/fft.cpp:161-182, :245-272); the exact kernel reproduces that drift bit for bit, the tolerance
# kernel uses correctly rounded twiddles and lands ~1000x closer to the true DFT.  So two bounds are asserted:
TOL_MAG_TRUE = 6e-7   # |mag - true DFT magnitude (float64 numpy)| <= this x the frame's peak: a few fp32 ulps of the peak
TOL_MAG_REF = 4e-4    # |mag - reference| <= this x the frame's peak: the reference's own distance from the true transform
TOL_MFCC_REF = 5e-4   # |mfcc - reference mfcc|: that magnitude difference through log and the 42-term DCT / 13


@pytest.mark.parametrize("nf,nc,nfr,off", [(42, 13, 1000, 0), (42, 13, 37, 1), (40, 20, 513, 0), (64, 13, 9, 0)])
def test_fused_tolerance_mode_within_stated_tolerance(mx, port, nf, nc, nfr, off):
    rng = np.random.default_rng(nf * 7 + nfr)
    stride = 1024 if off == 0 else 1025
    n = np.arange(stride * nfr + 8)
    sig = (0.4 * np.sin(2 * np.pi * 220 * n / 44100) + 0.3 * np.sin(2 * np.pi * 1337.5 * n / 44100) +
           0.1 * rng.uniform(-1, 1, n.size)).astype(np.float32)
    sig[: stride * 3] *= 1e-3                                   # a few quiet frames
    d = mx.DeviceBuffer.from_numpy(sig)
    f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC(); m.setup(512, nf, nc, 20.0, 20000.0)
    base = d.ptr + 4 * off
    exact = m.mfcc_of_frames(f, base, nfr, frame_stride=stride, want_mags=True).numpy()
    mags_exact = m.mags.numpy()
    L = mx.lib()
    prev = L.mxg_tune(b"fft_exact", 0)
    try:
        out = m.mfcc_of_frames(f, base, nfr, frame_stride=stride, want_mags=True).numpy()
        mags = m.mags.numpy()
        out_nomags = m.mfcc_of_frames(f, base, nfr, frame_stride=stride).numpy()     # the half-spectrum variant of the kernel
        prevl = L.mxg_tune(b"fused_layout", 2)                                        # ... and its 12-wave layout: the same arithmetic
        try:
            out_nomags12 = m.mfcc_of_frames(f, base, nfr, frame_stride=stride).numpy()
        finally:
            L.mxg_tune(b"fused_layout", prevl)
        assert np.array_equal(out_nomags12.view(np.uint64), out_nomags.view(np.uint64)), "tolerance mode: layouts differ"
    finally:
        L.mxg_tune(b"fft_exact", prev)
    frames = np.stack([sig[off + stride * k: off + stride * k + 1024] for k in range(nfr)])
    e = port.fft_stream(frames.reshape(-1), 1024, 1024, 1024, want=("mags",))["mags"]
    assert np.array_equal(f32bits(mags_exact), f32bits(e)), "the exact kernel must not be disturbed"
    # the true transform: float64 FFT of the frame times the plan's float32 Hann window (L/fft.cpp:409-413); bins 1..511 of the
    # reference's RealFFT are the DFT's (bin 0 packs DC and Nyquist, :274-275, and is left out of this comparison)
    win = (0.50 - 0.50 * np.cos(2 * np.pi * np.arange(1024) / 1023.0)).astype(np.float32).astype(np.float64)
    true = np.abs(np.fft.rfft(frames.astype(np.float64) * win, axis=1))[:, :512]
    peak = np.maximum(true[:, 1:].max(axis=1, keepdims=True), 1e-30)
    scale_ok = np.abs(e[:, 1:] - true[:, 1:]).max() / peak.max()
    assert scale_ok < 1e-2, "the float64 model of the reference's magnitudes is wrong (scale)"
    d_true = float((np.abs(mags[:, 1:].astype(np.float64) - true[:, 1:]) / peak).max())
    d_ref = float((np.abs(mags.astype(np.float64) - e) / np.maximum(e.max(axis=1, keepdims=True), 1e-30)).max())
    ref_true = float((np.abs(e[:, 1:].astype(np.float64) - true[:, 1:]) / peak).max())
    emel, emf = port.mfcc(e, nf, nc, 20.0, 20000.0)
    err = float(np.abs(out - emf).max())
    print("tolerance mode, %d filters, %d frames, x frame peak: |tol - true DFT| %.2e (stated %.0e), |reference - true DFT| %.2e, "
          "|tol - reference| %.2e (stated %.0e); mfcc |tol - reference| %.2e (stated %.0e), exact kernel %.2e"
          % (nf, nfr, d_true, TOL_MAG_TRUE, ref_true, d_ref, TOL_MAG_REF, err, TOL_MFCC_REF, np.abs(exact - emf).max()))
    assert d_true <= TOL_MAG_TRUE
    assert d_ref <= TOL_MAG_REF
    assert err <= TOL_MFCC_REF
    assert np.abs(out_nomags - emf).max() <= TOL_MFCC_REF
    assert d_true < 0.1 * ref_true, "the tolerance kernel is supposed to be much closer to the true transform than the reference is"
