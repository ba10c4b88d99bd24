"""GPU parity (-m gpu) against tests/golden/extra.npz (values dumped from the compiled reference):
noise, quad/ambisonic bus, playOnZX*, playWithPhasor, FFT features, playAtPosition, maxiPitchShift.
Everything here is bit-exact except the mix sums (tree order) and the log/exp based features."""
import numpy as np
import pytest

from conftest import assert_bits_equal, mix_tol

pytestmark = pytest.mark.gpu

ZX = ["playOnZX", "playOnZXAtSpeed", "playOnZXAtSpeedFromOffset", "playOnZXAtSpeedBetweenPoints", "loopSetPosOnZX"]


def test_noise_golden(mx, golden):
    g = golden("extra.npz")
    N, V = g["noise_rand"].shape
    bank = mx.maxiOscBank(V)
    assert_bits_equal(bank.noise(g["noise_rand"]).numpy(), g["noise_out"], "noise")


@pytest.mark.parametrize("C", [2, 4, 8])
def test_mix_bus_golden(mx, golden, C):
    g = golden("extra.npz")
    x = g["bus_x"]
    N, V = x.shape
    bus = mx.DeviceBuffer((N, C, V))
    mix = mx.maxiMixBank(V).bus(C, mx.DeviceBuffer.from_numpy(x), g["bus_px"], g["bus_py"] if C >= 4 else None,
                                g["bus_pz"] if C == 8 else None, bus=bus).numpy()
    assert_bits_equal(bus.numpy(), g["bus_%d" % C], "bus")
    e = g["mix_%d" % C]
    ok = np.isfinite(e)                       # ambisonic with z < 0: NaN channels in the reference too
    assert np.array_equal(np.isnan(mix), np.isnan(e))
    np.testing.assert_allclose(mix[ok], e[ok], rtol=0, atol=mix_tol(V, np.abs(x).max()))


@pytest.mark.parametrize("mode", range(5))
def test_sample_zx_golden(mx, golden, mode):
    g = golden("extra.npz")
    trig = g["zx_trig"]
    V = trig.shape[1]
    h = trig.shape[0] // 2
    bank = mx.maxiSampleBank(V)
    bank.setSample(g["smp"])
    bank.position.upload(g["zx_pos0"])
    kw = dict(a=g["zx_a"] if mode in (1, 2, 3) else None, p0=g["zx_p0"] if mode >= 2 else None,
              p1=g["zx_p1"] if mode == 3 else None)
    o = np.concatenate([bank.render_trig(ZX[mode], trig[:h], **kw).numpy(),
                        bank.render_trig(ZX[mode], trig[h:], **kw).numpy()])
    assert_bits_equal(o, g["zx_out_%d" % mode], ZX[mode])
    assert_bits_equal(bank.position.numpy(), g["zx_pos_%d" % mode])
    assert_bits_equal(bank.zx_prev.numpy(), g["zx_prev_%d" % mode])
    assert np.array_equal(bank.zx_first.numpy(), g["zx_first_%d" % mode])


def test_sample_phasor_golden(mx, golden):
    g = golden("extra.npz")
    pha = g["phasor_in"]
    bank = mx.maxiSampleBank(pha.shape[1])
    bank.setSample(g["smp"])
    o = np.concatenate([bank.playWithPhasor(pha[:150]).numpy(), bank.playWithPhasor(pha[150:]).numpy()])
    assert_bits_equal(o, g["phasor_out"], "playWithPhasor")
    assert_bits_equal(bank.phasor_prev.numpy(), g["phasor_prev"])
    assert np.array_equal(bank.phasor_first.numpy(), g["phasor_first"])


def test_fft_features_golden(mx, golden):
    g = golden("extra.npz")
    f = mx.maxiFFT()
    f.setup(1024, 512, 1024)
    dm = mx.DeviceBuffer.from_numpy(g["feat_mags"])
    assert_bits_equal(f.spectralCentroid(dm).numpy().astype(np.float64), g["feat_centroid"].astype(np.float64))
    np.testing.assert_allclose(f.spectralFlatness(dm).numpy(), g["feat_flatness"], rtol=4e-6, atol=0)
    db, e = f.magsToDB(dm).numpy(), g["feat_db"]
    assert np.abs(db.view(np.int32).astype(np.int64) - e.view(np.int32).astype(np.int64)).max() <= 4


@pytest.mark.parametrize("chunked", [1, 0], ids=["chunked", "serial"])
def test_play_at_position_and_pitch_shift_golden(mx, golden, chunked):
    g = golden("extra.npz")
    prev = mx.lib().mxg_tune(b"grain_chunked", chunked)
    try:
        smp, pos = g["g_samples"], g["pap_pos"]
        T, S = pos.shape
        h = T // 2
        sb = mx.maxiSampleBank(1)
        sb.setSample(smp)
        bank = mx.maxiTimeStretchBank(S, sb, "hann")
        o = np.concatenate([bank.playAtPosition(pos[:h], 0.05, 4).numpy(), bank.playAtPosition(pos[h:], 0.05, 4).numpy()])
        assert_bits_equal(o, g["pap_out"], "playAtPosition")
        assert_bits_equal(bank.state.numpy(), g["pap_st"])
        assert_bits_equal(bank.grains.numpy(), g["pap_gst"])
        ps = mx.maxiPitchShiftBank(S, sb, "hann")
        ps.state.upload(g["ps_st0"])
        o = np.concatenate([ps.play(g["ps_speed"], 0.05, 3, h, posMod=g["ps_posmod"]).numpy(),
                            ps.play(g["ps_speed"], 0.05, 3, T - h, posMod=g["ps_posmod"]).numpy()])
        assert_bits_equal(o, g["ps_out"], "maxiPitchShift")
        assert_bits_equal(ps.state.numpy(), g["ps_st"])
        assert_bits_equal(ps.grains.numpy(), g["ps_gst"])
    finally:
        mx.lib().mxg_tune(b"grain_chunked", prev)


def test_extra2_golden_filters(mx, golden):
    """maxiDCBlocker / maxiSVF / maxiBiquad against the reference's own values (tests/golden/extra2.npz)."""
    g = golden("extra2.npz")
    x = g["f2_x"]
    V = x.shape[1]
    d = lambda a: mx.DeviceBuffer.from_numpy(np.ascontiguousarray(a))
    dc = mx.maxiDCBlockerBank(V)
    o = np.concatenate([dc.play(d(x[:100]), g["dc_par"][0]).numpy(), dc.play(d(x[100:]), g["dc_par"][0]).numpy()])
    assert_bits_equal(o, g["dc_out"], "dcblocker")
    p = g["svf_par"]
    svf = mx.maxiSVFBank(V); svf.setCutoff(p[0]); svf.setResonance(p[1])
    o = np.concatenate([svf.play(d(x[:100]), *p[2:]).numpy(), svf.play(d(x[100:]), *p[2:]).numpy()])
    assert_bits_equal(svf.coefficients(), g["svf_coef"], "svf coefficients")
    assert_bits_equal(o, g["svf_out"], "svf")
    assert_bits_equal(svf.state.numpy(), g["svf_st"])
    for t in range(7):
        p = g["bq_par_%d" % t]
        bq = mx.maxiBiquadBank(V); bq.set(t, p[1], p[2], p[3])
        o = np.concatenate([bq.play(d(x[:100])).numpy(), bq.play(d(x[100:])).numpy()])
        assert_bits_equal(bq.host_coef, g["bq_coef_%d" % t], "biquad coefficients %d" % t)
        assert_bits_equal(o, g["bq_out_%d" % t], "biquad %d" % t)
        assert_bits_equal(bq.state.numpy(), g["bq_st_%d" % t])


@pytest.mark.parametrize("name", ["ar", "adsr", "curved"])
def test_extra2_golden_envgen(mx, golden, name):
    g = golden("extra2.npz")
    trig = g["eg_trig"]
    lv, tm, cv = g["eg_levels_" + name], g["eg_times_" + name], g["eg_curves_" + name]
    for loop, retrig in ((0, 0), (1, 1)):
        bank = mx.maxiEnvGenBank(trig.shape[1])
        assert bank.setup(lv, tm, cv, bool(loop), bool(retrig))
        o = np.concatenate([bank.play(trig[:1400]).numpy(), bank.play(trig[1400:]).numpy()])
        tag = "%s_%d%d" % (name, loop, retrig)
        assert_bits_equal(bank.host_stages, g["eg_stages_" + name], "stage table")
        assert np.array_equal(bank.istate.numpy(), g["eg_ist_" + tag])
        if name == "curved":
            assert np.abs(o - g["eg_out_" + tag]).max() <= 4 * 2.2e-16
        else:
            assert_bits_equal(o, g["eg_out_" + tag], tag)
            assert_bits_equal(bank.dstate.numpy(), g["eg_dst_" + tag])


@pytest.mark.parametrize("sustain", [1, 0])
def test_extra2_golden_sampler(mx, golden, sustain):
    g = golden("extra2.npz")
    L = mx.lib()
    V, voices, N = g["smp_pitch"].size, 8, 900
    d = lambda a: mx.DeviceBuffer.from_numpy(np.ascontiguousarray(a))
    sb = mx.maxiSampleBank(1)
    sb.setSample(g["smp_samples"])
    freq = np.zeros(V)
    assert L.mxg_sampler_freq_host(V, g["smp_pitch"].ctypes.data, g["smp_samples"].size, freq.ctypes.data) == 0
    dpos, dtrig, dout = d(np.zeros(V)), d(g["smp_trig0"]), d(np.zeros(V))
    ddst, dist = d(np.zeros((2, V))), d(np.zeros((6, V), np.int64))
    dfreq, dgain, dpar, dhold = d(freq), d(g["smp_gain"]), d(g["smp_par"]), d(g["smp_hold"].astype(np.int64))
    for blk in "ab":
        if blk == "b":
            t = dtrig.numpy(); t[::3] = 0; t[1::7] = 1; dtrig.upload(t)
        mix, outputs = mx.DeviceBuffer((N, V // voices)), mx.DeviceBuffer((N, V))
        assert L.mxg_sampler_render(V, N, voices, sustain, sb.d_samples, g["smp_samples"].size, dfreq.ptr, dgain.ptr,
                                    dpar.ptr, dhold.ptr, dpos.ptr, dtrig.ptr, dout.ptr, ddst.ptr, dist.ptr, mix.ptr,
                                    outputs.ptr, None) == 0
        key = lambda nm: g["smp_%s_%s%d" % (nm, blk, sustain)]
        assert_bits_equal(mix.numpy(), key("mix"), "play() " + blk)
        assert_bits_equal(outputs.numpy(), key("outputs"))
        assert_bits_equal(dpos.numpy(), key("position"))
        assert np.array_equal(dtrig.numpy(), key("trigger"))
        assert_bits_equal(ddst.numpy(), key("dst"))
        assert np.array_equal(dist.numpy(), key("ist"))
