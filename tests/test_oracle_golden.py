"""CPU tests (-m "not gpu"): the plain-C oracle against (a) the golden vectors dumped from the
reference itself and (b) the compiled reference, when oracle/_ref is present.  Bit-exact."""
import numpy as np
import pytest

from conftest import assert_bits_equal

OSC = ["sinewave", "coswave", "phasor", "saw", "triangle", "square", "pulse", "impulse",
       "sinebuf", "sinebuf4", "sawn", "phasorBetween"]
FLT = ["lores", "hires", "bandpass", "lopass", "hipass"]


def test_guards_and_kind(port):
    assert port.kind == "port"
    assert port.L.mxo_sine_table_guard() == 0.0
    assert port.L.mxo_transition_guard() == 0.0


def test_reference_guards(ref):
    # what the compiled reference holds at sineBuffer[-1] / transition[1001] in this image
    assert ref.kind == "reference"
    assert ref.L.mxo_sine_table_guard() == 0.0
    assert ref.L.mxo_transition_guard() == 0.0


def test_kat_sinewave_440(port, golden):
    # first four samples of maxiOsc::sinewave(440) at 44.1 kHz
    out, _, _ = port.osc(0, np.array([440.0]), 4)
    assert [repr(float(x)) for x in out[:, 0]] == [
        "0.0", "0.06264832417874368", "0.1250505236945281", "0.18696144082725336"]
    assert_bits_equal(out[:, 0], golden("osc.npz")["ka_sinewave440"])


@pytest.mark.parametrize("wf", range(12))
def test_osc_golden(port, golden, wf):
    g = golden("osc.npz")
    name = OSC[wf]
    N = int(g["N"])
    a, b = (g["duty"], None) if name == "pulse" else (g["p1"], g["p2"])
    o1, ph, hd = port.osc(wf, g["freq"], N, p1=a, p2=b)
    o2, ph, hd = port.osc(wf, g["freq"], N, phase=ph, hold=hd, p1=a, p2=b)
    assert_bits_equal(np.concatenate([o1, o2]), g["out_" + name], name)
    assert_bits_equal(ph, g["phase_" + name], name + " phase")
    assert_bits_equal(hd, g["hold_" + name], name + " hold")


@pytest.mark.parametrize("name", ["sinebuf", "saw", "sawn"])
def test_osc_fm_golden(port, golden, name):
    g = golden("osc.npz")
    o, ph, _ = port.osc(OSC.index(name), g["fm"], int(g["N"]), per_sample=True)
    assert_bits_equal(o, g["fm_out_" + name], name)
    assert_bits_equal(ph, g["fm_phase_" + name], name)


@pytest.mark.parametrize("kind", range(5))
def test_filter_golden(port, golden, kind):
    g = golden("filter.npz")
    name = FLT[kind]
    c = g["lp"] if kind >= 3 else g["cutoff"]
    r = None if kind >= 3 else (g["bres"] if kind == 2 else g["res"])
    o1, st = port.filter(kind, g["x"][:128], c, r)
    o2, st = port.filter(kind, g["x"][128:], c, r, state=st)
    assert_bits_equal(np.concatenate([o1, o2]), g["out_" + name], name)
    assert_bits_equal(st, g["state_" + name], name + " state")
    if kind <= 2:
        assert_bits_equal(port.filter_coeffs(kind, c, r), g["coef_" + name], name + " coef")


def test_filter_modulated_golden(port, golden):
    g = golden("filter.npz")
    o, _ = port.filter(0, g["x"], g["cutoff_mod"], g["res"], cps=True)
    assert_bits_equal(o, g["out_lores_mod"])


def test_env_golden(port, golden):
    g = golden("env.npz")
    setters = np.array([[port.env_coeff(w, ms) for ms in g["setter_ms"]] for w in range(4)])
    assert_bits_equal(setters, g["setters"], "setters")
    for mode, name in enumerate(["adsr", "ar"]):
        o, dst, ist = port.env(mode, None, g["trig"], g["par"], g["hold"])
        assert_bits_equal(o, g["out_%s_gate" % name], name)
        assert_bits_equal(dst, g["dst_%s_gate" % name], name)
        assert np.array_equal(ist, g["ist_%s_gate" % name])
        o, dst, ist = port.env(mode, g["xin"], g["trig_v"], g["par"], g["hold"])
        assert_bits_equal(o, g["out_%s_pv" % name], name)
        assert_bits_equal(dst, g["dst_%s_pv" % name], name)
        assert np.array_equal(ist, g["ist_%s_pv" % name])


@pytest.mark.parametrize("mode", [0, 1])
def test_voice_golden(port, golden, mode):
    g = golden("voice.npz")
    cu = g["cutoff"] if mode == 0 else np.full(g["freq"].size, 10000.0)
    o, ost, fst, dst, ist = port.voice(mode, g["freq"], cu, g["res"], g["trig"], g["par"], g["hold"])
    assert_bits_equal(o, g["out_mode%d" % mode])
    assert_bits_equal(ost, g["ost_mode%d" % mode])
    assert_bits_equal(fst, g["fst_mode%d" % mode])
    assert_bits_equal(dst, g["dst_mode%d" % mode])
    assert np.array_equal(ist, g["ist_mode%d" % mode])


def test_mix_golden(port, golden):
    g = golden("mix.npz")
    assert_bits_equal(port.mix_stereo(g["x"], g["pan"]), g["mix"])


# ---- port vs compiled reference on fresh seeded inputs (only where _ref exists) -----------------
@pytest.mark.parametrize("wf", range(12))
def test_osc_port_vs_reference(port, ref, wf):
    rng = np.random.default_rng(100 + wf)
    V, N = 193, 1500
    freq = rng.uniform(0.05, 21000, V)
    p2 = rng.uniform(0.2, 1.0, V)
    p1 = p2 * rng.uniform(0, 0.95, V)
    if wf == 6:
        p1 = rng.uniform(-0.2, 1.2, V)
    ph0 = rng.uniform(0, 1, V) if wf not in (8, 9) else rng.uniform(0, 500, V)
    a = port.osc(wf, freq, N, phase=ph0, p1=p1, p2=p2)
    b = ref.osc(wf, freq, N, phase=ph0, p1=p1, p2=p2)
    for x, y, w in zip(a, b, ("out", "phase", "hold")):
        assert_bits_equal(x, y, "%s %s" % (OSC[wf], w))


def test_filter_env_voice_port_vs_reference(port, ref):
    rng = np.random.default_rng(7)
    V, N = 37, 700
    x = rng.uniform(-1, 1, (N, V))
    cutoff = rng.uniform(1, 30000, V)
    res = rng.uniform(0.2, 25, V)
    for kind in range(5):
        c = rng.uniform(0, 1, V) if kind >= 3 else cutoff
        r = None if kind >= 3 else (rng.uniform(0.01, 1.3, V) if kind == 2 else res)
        a, sa = port.filter(kind, x, c, r)
        b, sb = ref.filter(kind, x, c, r)
        assert_bits_equal(a, b, FLT[kind])
        assert_bits_equal(sa, sb, FLT[kind])
        if kind <= 2:
            assert_bits_equal(port.filter_coeffs(kind, c, r), ref.filter_coeffs(kind, c, r))
    par = np.stack([rng.uniform(1e-4, 0.5, V), rng.uniform(0.9, 0.9999, V), rng.uniform(0, 1, V),
                    rng.uniform(0.9, 0.9999, V)])
    hold = rng.integers(0, 50, V)
    trig = (rng.uniform(0, 1, (N, V)) < 0.01).astype(np.int32)
    for mode in (0, 1):
        a = port.env(mode, x, trig, par, hold)
        b = ref.env(mode, x, trig, par, hold)
        assert_bits_equal(a[0], b[0])
        assert_bits_equal(a[1], b[1])
        assert np.array_equal(a[2], b[2])
    freq = rng.uniform(20, 5000, V)
    gate = ((np.arange(N) % 300) < 120).astype(np.int32)
    for mode in (0, 1):
        a = port.voice(mode, freq, cutoff, res, gate, par, hold)
        b = ref.voice(mode, freq, cutoff, res, gate, par, hold)
        for i in range(4):
            assert_bits_equal(a[i], b[i], "voice mode %d item %d" % (mode, i))
        assert np.array_equal(a[4], b[4])
    pan = rng.uniform(-0.1, 1.1, V)
    assert_bits_equal(port.mix_stereo(x, pan), ref.mix_stereo(x, pan))


SMP = ["play", "playOnce", "playLoop", "playUntil", "playAtSpeed", "playOnceAtSpeed",
       "playUntilAtSpeed", "play4", "playAtSpeedBetweenPoints"]


def test_delay_golden(port, golden):
    g = golden("delay.npz")
    cap = int(g["cap"])
    for mode, name in enumerate(["dl", "dlFromPosition"]):
        o1, mem, ph = port.delay(mode, g["x"][:250], g["size"], g["fb"], cap, position=g["pos"])
        o2, mem, ph = port.delay(mode, g["x"][250:], g["size"], g["fb"], cap, position=g["pos"], mem=mem, phase=ph)
        assert_bits_equal(np.concatenate([o1, o2]), g["out_" + name], name)
        assert_bits_equal(mem, g["mem_" + name], name + " mem")
        assert np.array_equal(ph, g["phase_" + name])


@pytest.mark.parametrize("mode", range(9))
def test_sample_golden(port, golden, mode):
    g = golden("sample.npz")
    name = SMP[mode]
    N = int(g["N"])
    args = dict(a=g["a_" + name], start=g["start_" + name], end=g["end_" + name])
    o1, p = port.sample(mode, g["samples"], N // 2, g["pos0_" + name], **args)
    o2, p = port.sample(mode, g["samples"], N - N // 2, p, **args)
    assert_bits_equal(np.concatenate([o1, o2]), g["out_" + name], name)
    assert_bits_equal(p, g["pos_" + name], name + " position")


def test_sample_speed_mod_sr96k_golden(port, golden):
    g = golden("sample.npz")
    port.settings(96000, 2, 1024)
    try:
        o, p = port.sample(4, g["samples"], int(g["N"]), np.zeros(g["speed_mod"].shape[1]),
                           a=g["speed_mod"], aps=True, mySampleRate=44100)
    finally:
        port.settings(44100, 2, 1024)
    assert_bits_equal(o, g["out_speed_mod_sr96k"])
    assert_bits_equal(p, g["pos_speed_mod_sr96k"])


def test_delay_sample_port_vs_reference(port, ref):
    rng = np.random.default_rng(11)
    V, N, cap = 13, 900, 50
    x = rng.uniform(-1, 1, (N, V))
    size = rng.integers(1, cap + 1, V)
    fb = rng.uniform(0, 0.99, V)
    pos = rng.integers(0, cap + 5, V)
    for mode in (0, 1):
        a = port.delay(mode, x, size, fb, cap, position=pos)
        b = ref.delay(mode, x, size, fb, cap, position=pos)
        assert_bits_equal(a[0], b[0]); assert_bits_equal(a[1], b[1]); assert np.array_equal(a[2], b[2])
    Ls = 777
    smp = rng.uniform(-1, 1, Ls)
    for mode in range(9):
        pos0 = rng.uniform(1, Ls - 3, V)
        if mode in (0, 1):
            pos0 = np.floor(pos0)
        a_ = rng.uniform(0.1, 4.0, V)
        st, en = rng.uniform(0, 0.5, V), rng.uniform(0.4, 1.2, V)
        if mode == 2:
            en = np.minimum(en, 1.0)  # playLoop does not clamp `end`: > 1 is an OOB read in the reference
        if mode in (7, 8):
            a_ = rng.uniform(0.5, 80, V) * np.where(rng.uniform(0, 1, V) < 0.5, -1, 1)
            st, en = np.floor(rng.uniform(2, 300, V)), np.floor(rng.uniform(350, Ls - 1, V))
        a = port.sample(mode, smp, N, pos0, a=a_, start=st, end=en)
        b = ref.sample(mode, smp, N, pos0, a=a_, start=st, end=en)
        assert_bits_equal(a[0], b[0], SMP[mode]); assert_bits_equal(a[1], b[1], SMP[mode])


def test_survey_mfcc_anchor(port):
    """SURVEY 8(c)'s recorded anchor: the third frame of sawn(220) through fft.setup(1024, 512, 1024) and
    mfcc.setup(512, 42, 13, 20, 20000) (g++ 11.4 / glibc of this image)."""
    sig, _, _ = port.osc(10, np.array([220.0]), 1024 * 3)
    e = port.fft_stream(sig[:, 0].astype(np.float32), 1024, 512, 1024, want=("mags",))
    _, mf = port.mfcc(e["mags"], 42, 13, 20.0, 20000.0)
    assert mf[2, 0] == 0.40082902638055068 and mf[2, 1] == -0.31312622651102895 and mf[2, 12] == 0.32965843633589265
