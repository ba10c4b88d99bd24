"""CPU tests (-m "not gpu") for the rows closed late in round 1: maxiOsc::noise, maxiMix quad/ambisonic,
the trigger-driven maxiSample players, playWithPhasor, magsToDB/spectralFlatness/spectralCentroid,
maxiTimeStretch::playAtPosition and maxiPitchShift.  The plain-C oracle against (a) tests/golden/extra.npz
(dumped from the compiled reference by oracle/gen_golden.py) and (b) the compiled reference on fresh
inputs when oracle/_ref is present.  Bit-exact throughout."""
import numpy as np
import pytest

from conftest import assert_bits_equal


def test_noise_golden(port, golden):
    g = golden("extra.npz")
    N, V = g["noise_rand"].shape
    rnd, out = port.noise(int(g["noise_seed"]), V, N)
    assert np.array_equal(rnd, g["noise_rand"])          # glibc rand() after srand(seed)
    assert_bits_equal(out, g["noise_out"], "noise")
    # the arithmetic alone, from the stored draws (what the HIP kernel is given)
    r = g["noise_rand"].astype(np.float32) / np.float32(2147483648.0)
    assert_bits_equal((r * np.float32(2) - np.float32(1)).astype(np.float64), g["noise_out"], "noise formula")


@pytest.mark.parametrize("C", [2, 4, 8])
def test_mix_bus_golden(port, golden, C):
    g = golden("extra.npz")
    mix, bus = port.mix_bus(C, g["bus_x"], g["bus_px"], g["bus_py"], g["bus_pz"], want_bus=True)
    assert_bits_equal(bus, g["bus_%d" % C], "bus")
    assert_bits_equal(mix, g["mix_%d" % C], "mix")
    if C == 2:   # same numbers as the older stereo entry point
        assert_bits_equal(port.mix_stereo(g["bus_x"], g["bus_px"]), mix)


@pytest.mark.parametrize("mode", range(5))
def test_sample_zx_golden(port, golden, mode):
    g = golden("extra.npz")
    trig = g["zx_trig"]
    h = trig.shape[0] // 2
    kw = dict(a=g["zx_a"], p0=g["zx_p0"], p1=g["zx_p1"])
    o1, p, zp, zf = port.sample_zx(mode, g["smp"], trig[:h], g["zx_pos0"], **kw)
    o2, p, zp, zf = port.sample_zx(mode, g["smp"], trig[h:], p, zx_prev=zp, zx_first=zf, **kw)
    assert_bits_equal(np.concatenate([o1, o2]), g["zx_out_%d" % mode])
    assert_bits_equal(p, g["zx_pos_%d" % mode])
    assert_bits_equal(zp, g["zx_prev_%d" % mode])
    assert np.array_equal(zf, g["zx_first_%d" % mode])


def test_sample_phasor_golden(port, golden):
    g = golden("extra.npz")
    pha = g["phasor_in"]
    o1, pp, pf = port.sample_phasor(g["smp"], pha[:150])
    o2, pp, pf = port.sample_phasor(g["smp"], pha[150:], phasor_prev=pp, phasor_first=pf)
    assert_bits_equal(np.concatenate([o1, o2]), g["phasor_out"])
    assert_bits_equal(pp, g["phasor_prev"])
    assert np.array_equal(pf, g["phasor_first"])


def test_fft_features_golden(port, golden):
    g = golden("extra.npz")
    flat, cen = port.fft_features(g["feat_mags"], 1024)
    assert_bits_equal(flat.astype(np.float64), g["feat_flatness"].astype(np.float64), "flatness")
    assert_bits_equal(cen.astype(np.float64), g["feat_centroid"].astype(np.float64), "centroid")
    assert_bits_equal(port.fft_to_db(g["feat_mags"]).astype(np.float64), g["feat_db"].astype(np.float64), "dB")


def test_play_at_position_and_pitch_shift_golden(port, golden):
    g = golden("extra.npz")
    smp, pos = g["g_samples"], g["pap_pos"]
    T = pos.shape[0]
    h = T // 2
    o1, st, gst, rc = port.granular(2, 0, smp, h, pos[:h], grainLength=0.05, overlaps=4)
    o2, st, gst, rc2 = port.granular(2, 0, smp, T - h, pos[h:], grainLength=0.05, overlaps=4, st=st, gst=gst)
    assert rc == 0 and rc2 == 0
    assert_bits_equal(np.concatenate([o1, o2]), g["pap_out"], "playAtPosition")
    assert_bits_equal(st, g["pap_st"])
    assert_bits_equal(gst, g["pap_gst"])
    kw = dict(posMod=g["ps_posmod"], grainLength=0.05, overlaps=3)
    o1, st, gst, rc = port.granular(3, 0, smp, h, g["ps_speed"], st=g["ps_st0"], **kw)
    o2, st, gst, rc2 = port.granular(3, 0, smp, T - h, g["ps_speed"], st=st, gst=gst, **kw)
    assert rc == 0 and rc2 == 0
    assert_bits_equal(np.concatenate([o1, o2]), g["ps_out"], "maxiPitchShift")
    assert_bits_equal(st, g["ps_st"])
    assert_bits_equal(gst, g["ps_gst"])


def test_extra_port_vs_reference(port, ref):
    """Fresh seeded inputs through both CPU checkers (only where oracle/_ref was built)."""
    rng = np.random.default_rng(99)
    V, N = 19, 257
    x = rng.uniform(-1, 1, (N, V))
    for C in (2, 4, 8):
        px, py, pz = rng.uniform(-0.2, 1.2, V), rng.uniform(-0.2, 1.2, V), rng.uniform(-0.3, 1.3, V)
        a, b = port.mix_bus(C, x, px, py, pz, True), ref.mix_bus(C, x, px, py, pz, True)
        assert_bits_equal(a[0], b[0])
        assert_bits_equal(a[1], b[1])
    for u, w in zip(port.noise(5, V, N), ref.noise(5, V, N)):
        assert np.array_equal(u, w)
    smp = rng.uniform(-1, 1, 1500)
    trig = np.sin(np.arange(N)[:, None] * rng.uniform(0.02, 0.3, V)[None, :] + rng.uniform(0, 6, V))
    sp = rng.uniform(0.3, 2.5, (N, V))
    for mode in range(5):
        kw = dict(a=sp, aps=True, p0=rng.uniform(0, 0.6, V), p1=rng.uniform(0.1, 0.5, V))
        pos0 = rng.uniform(0, 1499, V)
        for u, w in zip(port.sample_zx(mode, smp, trig, pos0, **kw), ref.sample_zx(mode, smp, trig, pos0, **kw)):
            assert_bits_equal(np.asarray(u, np.float64), np.asarray(w, np.float64))
    pha = rng.uniform(-0.1, 1.1, (N, V))
    pha[:, :8] = (np.arange(N)[:, None] * rng.uniform(0.0005, 0.01, 8)[None, :]) % 1.0
    for u, w in zip(port.sample_phasor(smp, pha), ref.sample_phasor(smp, pha)):
        assert_bits_equal(np.asarray(u, np.float64), np.asarray(w, np.float64))
    m = np.abs(rng.normal(0, 10, (50, 256))).astype(np.float32)
    for u, w in zip(port.fft_features(m, 512), ref.fft_features(m, 512)):
        assert np.array_equal(u.view(np.uint32), w.view(np.uint32))
    L, S, T = 20000, 9, 2500
    gs = rng.uniform(-1, 1, L)
    pos = rng.uniform(0, 1, (T, S))
    for u, w in zip(port.granular(2, 4, gs, T, pos, grainLength=0.04, overlaps=3),
                    ref.granular(2, 4, gs, T, pos, grainLength=0.04, overlaps=3)):
        assert_bits_equal(np.asarray(u, np.float64), np.asarray(w, np.float64))
    speed = rng.uniform(-2, 2.5, S)
    for u, w in zip(port.granular(3, 6, gs, T, speed, posMod=rng.uniform(-0.1, 0.1, S) * 0 + 0.05, grainLength=0.04, overlaps=3),
                    ref.granular(3, 6, gs, T, speed, posMod=np.full(S, 0.05), grainLength=0.04, overlaps=3)):
        assert_bits_equal(np.asarray(u, np.float64), np.asarray(w, np.float64))


def test_filter2_port_vs_reference(port, ref):
    """maxiDCBlocker / maxiSVF / maxiBiquad (H:1255-1486): restatement == compiled reference, incl. the
    coefficients setParams()/set() leave in the members."""
    rng = np.random.default_rng(4)
    V, N = 35, 500
    x = rng.uniform(-1, 1, (N, V))
    R = rng.uniform(0.9, 0.9999, (1, V))
    for u, w in zip(port.filter2(0, x, R), ref.filter2(0, x, R)):
        assert_bits_equal(u, w)
    par = np.stack([rng.uniform(20, 20000, V), rng.uniform(0, 12, V), *rng.uniform(0, 1, (4, V))])
    par[1, :3] = 0
    for u, w in zip(port.filter2(1, x, par), ref.filter2(1, x, par)):
        assert_bits_equal(u, w)
    for t in range(7):
        par = np.stack([np.full(V, float(t)), rng.uniform(30, 18000, V), rng.uniform(0.3, 8, V), rng.uniform(-18, 18, V)])
        for u, w in zip(port.filter2(2, x, par), ref.filter2(2, x, par)):
            assert_bits_equal(u, w)


def test_envgen_port_vs_reference(port, ref):
    """maxiEnvGen (H:2268-2547): restatement == compiled reference for AR/ASR/ADSR/curved shapes, loop and
    retrigger on/off, per-voice gates, impulses and a constant-1 trigger, state carried across two calls."""
    H = port.ENVGEN_HOLD
    rng = np.random.default_rng(6)
    V, N = 23, 4000
    n = np.arange(N)[:, None]
    trig = np.sign(np.sin(n * rng.uniform(0.002, 0.01, V)[None, :] + rng.uniform(0, 6, V)))
    trig[:, 3] = 1.0
    trig[:, 4] = (np.arange(N) % 700 < 5) * 1.0
    cases = [([0, 1, 0], [10, 40], [1, 1]), ([0, 1, 1, 0], [5, H, 30], [1, 1, 1]),
             ([0, 1, 0.4, 0.4, 0], [3, 12, H, 25], [1, 1, 1, 1]), ([0, 1, 0.2, 0], [7.3, 11.1, 20.7], [0.5, 2, 3])]
    for lv, tm, cv in cases:
        for loop in (0, 1):
            for retrig in (0, 1):
                a = port.envgen(trig[:N // 2], lv, tm, cv, loop, retrig)
                a2 = port.envgen(trig[N // 2:], lv, tm, cv, loop, retrig, dst=a[1], ist=a[2])
                b = ref.envgen(trig[:N // 2], lv, tm, cv, loop, retrig)
                b2 = ref.envgen(trig[N // 2:], lv, tm, cv, loop, retrig, dst=b[1], ist=b[2])
                for u, w in zip(a + a2, b + b2):
                    assert_bits_equal(np.asarray(u, np.float64), np.asarray(w, np.float64))


def test_sampler_port_vs_reference(port, ref):
    """maxiSampler::play (L/maxiSynths.cpp:289-312): restatement == compiled reference, 32/8/5 voices, sustain on/off,
    with note-offs and re-triggers between two blocks."""
    rng = np.random.default_rng(8)
    smp = rng.uniform(-1, 1, 3000)
    for voices in (32, 8, 5):
        NS, N = 3, 1200
        V = NS * voices
        pitch = rng.integers(-24, 25, V).astype(np.float64)
        gain = rng.uniform(0.2, 1.0, V)
        par = np.stack([rng.uniform(0.001, 0.2, V), rng.uniform(0.99, 0.9999, V), rng.uniform(0.3, 1.0, V), rng.uniform(0.99, 0.9999, V)])
        hold = rng.integers(1, 50, V)
        trig = (rng.uniform(size=V) < 0.6).astype(np.int32)
        for sustain in (1, 0):
            a = port.sampler(voices, smp, N, pitch, gain, par, hold, np.zeros(V), trig, sustain)
            b = ref.sampler(voices, smp, N, pitch, gain, par, hold, np.zeros(V), trig, sustain)
            t2 = a[3].copy(); t2[::3] = 0; t2[1::7] = 1
            a2 = port.sampler(voices, smp, N, pitch, gain, par, hold, a[2], t2, sustain, a[4], a[5], a[6])
            b2 = ref.sampler(voices, smp, N, pitch, gain, par, hold, b[2], t2, sustain, b[4], b[5], b[6])
            for u, w in zip(a + a2, b + b2):
                assert_bits_equal(np.asarray(u, np.float64), np.asarray(w, np.float64))


def test_extra2_golden_filters(port, golden):
    g = golden("extra2.npz")
    x = g["f2_x"]
    o1, st, _ = port.filter2(0, x[:100], g["dc_par"])
    o2, st, _ = port.filter2(0, x[100:], g["dc_par"], st)
    assert_bits_equal(np.concatenate([o1, o2]), g["dc_out"], "dcblocker")
    assert_bits_equal(st, g["dc_st"])
    o1, st, c = port.filter2(1, x[:100], g["svf_par"])
    o2, st, c = port.filter2(1, x[100:], g["svf_par"], st)
    assert_bits_equal(np.concatenate([o1, o2]), g["svf_out"], "svf")
    assert_bits_equal(st, g["svf_st"])
    assert_bits_equal(c, g["svf_coef"], "svf coefficients")
    for t in range(7):
        o1, st, c = port.filter2(2, x[:100], g["bq_par_%d" % t])
        o2, st, c = port.filter2(2, x[100:], g["bq_par_%d" % t], st)
        assert_bits_equal(np.concatenate([o1, o2]), g["bq_out_%d" % t], "biquad %d" % t)
        assert_bits_equal(st, g["bq_st_%d" % t])
        assert_bits_equal(c, g["bq_coef_%d" % t], "biquad coefficients %d" % t)


@pytest.mark.parametrize("name", ["ar", "adsr", "curved"])
def test_extra2_golden_envgen(port, golden, name):
    g = golden("extra2.npz")
    trig = g["eg_trig"]
    lv, tm, cv = g["eg_levels_" + name], g["eg_times_" + name], g["eg_curves_" + name]
    for loop, retrig in ((0, 0), (1, 1)):
        o1, ds, is_, stg = port.envgen(trig[:1400], lv, tm, cv, loop, retrig)
        o2, ds, is_, _ = port.envgen(trig[1400:], lv, tm, cv, loop, retrig, dst=ds, ist=is_)
        tag = "%s_%d%d" % (name, loop, retrig)
        assert_bits_equal(np.concatenate([o1, o2]), g["eg_out_" + tag], tag)
        assert_bits_equal(ds, g["eg_dst_" + tag])
        assert np.array_equal(is_, g["eg_ist_" + tag])
    assert_bits_equal(stg, g["eg_stages_" + name], "stage table")


@pytest.mark.parametrize("sustain", [1, 0])
def test_extra2_golden_sampler(port, golden, sustain):
    g = golden("extra2.npz")
    V = g["smp_pitch"].size
    args = (8, g["smp_samples"], 900, g["smp_pitch"], g["smp_gain"], g["smp_par"], g["smp_hold"])
    a = port.sampler(*args, np.zeros(V), g["smp_trig0"], sustain)
    t2 = a[3].copy(); t2[::3] = 0; t2[1::7] = 1
    b = port.sampler(*args, a[2], t2, sustain, a[4], a[5], a[6])
    for k, nm in enumerate(["mix", "outputs", "position", "trigger", "outhold", "dst", "ist"]):
        assert_bits_equal(np.asarray(a[k], np.float64), np.asarray(g["smp_%s_a%d" % (nm, sustain)], np.float64), nm + " a")
        assert_bits_equal(np.asarray(b[k], np.float64), np.asarray(g["smp_%s_b%d" % (nm, sustain)], np.float64), nm + " b")


def test_osc_tables_extension_is_sinebuf_when_every_table_is_sinebuffer(port):
    """The checker of the per-voice wavetable EXTENSION (mxo_osc_tables; no reference counterpart) restates C:266-274 with the voice's
    own table: with every table = sineBuffer it must give mxo_osc(sinebuf)'s bits -- which tests/test_oracle_golden.py pins to the
    reference -- over carried blocks, wraps at 511 and phases in (-1, 0) included."""
    rng = np.random.default_rng(77)
    V, N = 300, 700
    freq = rng.uniform(20, 20000, V)
    T = port.sine_table()
    assert T.shape == (514,) and abs(T[128] - 1.0) < 1e-4
    tabs = np.tile(T, (V, 1))
    o1, ph1, hd1 = port.osc_tables(freq, tabs, N)
    o2, ph2, hd2 = port.osc_tables(freq, tabs, N, phase=ph1, hold=hd1)
    e, eph, ehd = port.osc(8, freq, 2 * N)
    assert np.array_equal(np.concatenate([o1, o2]).view(np.uint64), e.view(np.uint64))
    assert np.array_equal(ph2.view(np.uint64), eph.view(np.uint64)) and np.array_equal(hd2.view(np.uint64), ehd.view(np.uint64))
    # and with tables of its own a voice follows ITS table: a constant table gives the constant
    tabs2 = np.tile(np.arange(V, dtype=np.float64)[:, None], (1, 514))
    o3, _, _ = port.osc_tables(freq, tabs2, 5)
    assert np.array_equal(o3, np.tile(np.arange(V, dtype=np.float64), (5, 1)))
