"""GPU parity (-m gpu) against tests/golden/extra.npz (values dumped from the compiled reference):
noise, quad/ambisonic bus, playOnZX*, playWithPhasor, FFT features, playAtPosition, maxiPitchShift.
Everything here is bit-exact except the mix sums (tree order) and the log/exp based features."""
import numpy as np
import pytest

from conftest import assert_bits_equal

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
    np.testing.assert_allclose(mix[ok], e[ok], rtol=0, atol=1e-12 * V)


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
