"""GPU (-m gpu): degenerate and hostile shapes through every bank entry point -- empty banks/blocks, one voice x
one sample, voice counts that are not multiples of the wavefront, extreme / non-finite parameters.  Results are
compared with the oracle where it is defined; otherwise the call must simply not fault and leave state intact."""
import numpy as np
import pytest

from conftest import assert_bits_equal, ulp_diff

pytestmark = pytest.mark.gpu


def test_empty_shapes_are_noops(mx):
    L = mx.lib()
    b = mx.DeviceBuffer(8)
    i = mx.DeviceBuffer(8, np.int64)
    assert L.mxg_osc_render(8, 0, 16, b.ptr, 0, None, None, b.ptr, b.ptr, b.ptr, None) == 0
    assert L.mxg_osc_render(8, 4, 0, b.ptr, 0, None, None, b.ptr, b.ptr, b.ptr, None) == 0
    assert L.mxg_filter_render(3, 0, 5, b.ptr, b.ptr, 0, None, 0, None, b.ptr, b.ptr, None) == 0
    assert L.mxg_env_render(0, 0, 5, b.ptr, b.ptr, 0, b.ptr, i.ptr, b.ptr, i.ptr, b.ptr, None) == 0
    assert L.mxg_delay_render(0, 4, 0, b.ptr, b.ptr, b.ptr, None, b.ptr, 2, b.ptr, b.ptr, None) == 0
    assert L.mxg_mix_stereo(4, 0, b.ptr, b.ptr, b.ptr, None) == 0
    assert L.mxg_filter2_render(1, 0, 3, b.ptr, b.ptr, b.ptr, b.ptr, None) == 0
    assert L.mxg_osc_noise(0, 3, b.ptr, None, b.ptr, None) == 0
    L.mxg_sync()
    assert (b.numpy() == 0).all()


@pytest.mark.parametrize("V,N", [(1, 1), (1, 513), (63, 1), (65, 9), (257, 130)])
def test_small_and_ragged_banks_vs_oracle(mx, port, V, N):
    rng = np.random.default_rng(V * 1000 + N)
    freq = rng.uniform(0.1, 20000, V)
    for wf in (0, 3, 8, 9, 10):
        bank = mx.maxiOscBank(V)
        o = bank.render(wf, freq, N).numpy()
        e, ph, _ = port.osc(wf, freq, N)
        if wf == 0:
            assert ulp_diff(o, e).max() <= 1
        else:
            assert_bits_equal(o, e, "waveform %d" % wf)
            assert_bits_equal(bank.phase.numpy(), ph)
    x = rng.uniform(-1, 1, (N, V))
    dx = mx.DeviceBuffer.from_numpy(x)
    fb = mx.maxiFilterBank(V)
    cut, res = rng.uniform(20, 15000, V), rng.uniform(1, 10, V)
    assert_bits_equal(fb.render("lores", dx, cut, res).numpy(), port.filter(0, x, cut, res)[0], "lores")
    m = mx.maxiMixBank(V).stereo(dx, rng.uniform(0, 1, V)).numpy()
    assert m.shape == (N, 2) and np.isfinite(m).all()
    d = mx.maxiDelaylineBank(V, 40)
    size = rng.integers(1, 41, V).astype(np.int32)
    assert_bits_equal(d.dl(dx, size, 0.5).numpy(), port.delay(0, x, size, np.full(V, 0.5), 40)[0], "dl")


def test_hostile_parameters_do_not_fault(mx, port):
    """Zero, negative, huge and non-finite frequencies / cutoffs: same bits as the reference (incl. NaN/Inf)."""
    V, N = 64, 40
    freq = np.array([0.0, -440.0, 1e300, 1e-300, np.inf, -np.inf, np.nan, 22050.0, 44100.0, 88200.0, 5e-324, -0.0]
                    + [440.0] * (V - 12))
    for wf in (2, 3, 4, 5, 7, 8, 9):            # phasor, saw, triangle, square, impulse, sinebuf, sinebuf4
        if wf in (8, 9):   # wavetables: outside [0, sr/2] the reference indexes beyond sineBuffer (undefined)
            f = np.where(np.isfinite(freq) & (freq >= 0) & (freq <= 22050.0), freq, 440.0)
        else:
            f = freq
        o = mx.maxiOscBank(V).render(wf, f, N).numpy()
        e, _, _ = port.osc(wf, f, N)
        assert_bits_equal(o, e, "waveform %d" % wf)
    x = np.random.default_rng(2).uniform(-1, 1, (N, V))
    x[3, :4] = [np.nan, np.inf, -np.inf, 1e308]
    dx = mx.DeviceBuffer.from_numpy(x)
    cut = np.array([0.0, -5.0, 1e9, np.nan] + [1000.0] * (V - 4))
    res = np.array([0.0, 0.5, 1e9, 2.0] + [2.0] * (V - 4))
    for kind, name in enumerate(["lores", "hires", "bandpass", "lopass", "hipass"]):
        c = np.clip(np.nan_to_num(cut, nan=0.5), 0, 1) if kind >= 3 else cut
        o = mx.maxiFilterBank(V).render(name, dx, c, res).numpy()
        assert_bits_equal(o, port.filter(kind, x, c, res)[0], name)


@pytest.mark.parametrize("rw", [0, 3, 4])
def test_blocks_that_are_only_8_byte_aligned(mx, port, rw):
    """The 16-byte pair-row streams need 16-byte aligned blocks; a block that starts 8 bytes into an allocation (a caller's view into a
    larger buffer) must fall back to the 8-byte streams, automatically or with the pair rows requested by the knob, and give the same
    bits: K1 (sinebuf, large enough for the pair-row rule), maxiFilter, maxiEnv, maxiDelayline, maxiSample, maxiEnvGen, noise."""
    L = mx.lib()
    chk = mx._lib.check
    V, N = 1026, 48
    rng = np.random.default_rng(rw)
    big_in = mx.DeviceBuffer.from_numpy(np.concatenate([[0.0], rng.uniform(-1, 1, N * V)]))   # the block starts at element 1
    big_out = mx.DeviceBuffer(N * V + 1)
    x = big_in.numpy()[1:].reshape(N, V)
    pin, pout = big_in.ptr + 8, big_out.ptr + 8

    keep = []

    def D(a):   # (device copies of the parameters must outlive the asynchronous launches that read them)
        keep.append(mx.DeviceBuffer.from_numpy(a))
        return keep[-1].ptr

    def result():
        return big_out.numpy()[1:].reshape(N, V).copy()

    prev = L.mxg_tune(b"rw_store", rw)
    prev_osc = L.mxg_tune(b"osc_store", 4 if rw else 0)   # (K1: pair rows requested -- must be refused for this block)
    try:
        # K1
        freq = rng.uniform(20, 9000, V)
        bank = mx.maxiOscBank(V)
        chk(L.mxg_osc_render(8, V, N, D(freq), 0, None, None, bank.phase.ptr, bank.output.ptr, pout, None), "osc")
        assert_bits_equal(result(), port.osc(8, freq, N)[0], "sinebuf")
        # maxiFilter lopass
        c = rng.uniform(0, 1, V)
        fb = mx.maxiFilterBank(V)
        chk(L.mxg_filter_render(3, V, N, pin, D(c), 0, None, 0, None, fb.state.ptr, pout, None), "lopass")
        assert_bits_equal(result(), port.filter(3, x, c, None)[0], "lopass")
        # maxiEnv adsr
        eb = mx.maxiEnvBank(V); eb.setAttack(5); eb.setDecay(20); eb.setSustain(0.5); eb.setRelease(40)
        dpar, dhold = eb._params()
        trig = ((np.arange(N) % 30) < 20).astype(np.int32)
        chk(L.mxg_env_render(0, V, N, pin, D(trig), 0, dpar.ptr, dhold.ptr, eb.dstate.ptr, eb.istate.ptr, pout,
                             None), "adsr")
        assert_bits_equal(result(), port.env(0, x, trig, eb.par, eb.holdtime)[0], "adsr")
        # maxiDelayline
        db = mx.maxiDelaylineBank(V, 64)
        size = np.full(V, 40, np.int32); fbk = np.full(V, 0.5)
        chk(L.mxg_delay_render(0, V, N, pin, D(size), D(fbk), None,
                               db.memory.ptr, 64, db.phase.ptr, pout, None), "dl")
        assert_bits_equal(result(), port.delay(0, x, size, fbk, 64)[0], "dl")
        # maxiSample playAtSpeed
        smp = rng.uniform(-1, 1, 5000)
        sb = mx.maxiSampleBank(V); sb.setSample(smp)
        pos0 = rng.uniform(0, 4000, V); sb.position.upload(pos0)
        sp = rng.uniform(0.5, 1.5, V)
        chk(L.mxg_sample_render(4, V, N, sb.d_samples, sb.length, 44100, D(sp), 0, None, None,
                                sb.position.ptr, pout, None), "playAtSpeed")
        assert_bits_equal(result(), port.sample(4, smp, N, pos0, a=sp)[0], "playAtSpeed")
        # maxiEnvGen
        eg = mx.maxiEnvGenBank(V); assert eg.setupADSR(2, 5, 0.4, 10)
        gate = np.where((np.arange(N) % 30) < 20, 1.0, -1.0)
        chk(L.mxg_envgen_render(V, N, D(gate), 0, eg.stages.ptr, 4, 0, 0, eg.dstate.ptr, eg.istate.ptr, pout,
                                None), "envgen")
        assert_bits_equal(result(), port.envgen(gate, [0, 1, 0.4, 0.4, 0], [2, 5, -46692.0, 10], [1, 1, 1, 1], V=V)[0], "envgen")
        # noise
        rnd, e = port.noise(77, V, N)
        chk(L.mxg_osc_noise(V, N, D(np.ascontiguousarray(rnd, np.int32)), None, pout, None), "noise")
        assert_bits_equal(result(), e, "noise")
    finally:
        L.mxg_tune(b"rw_store", prev)
        L.mxg_tune(b"osc_store", prev_osc)
