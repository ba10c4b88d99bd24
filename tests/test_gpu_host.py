"""GPU (-m gpu): the C++ host facade (include/maximilian_bank.hpp) driven through the reference's
setup()/play() plugin shape by host/polysynth_host.cpp, compared bit-for-bit with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal

pytestmark = pytest.mark.gpu


def test_polysynth_host_matches_oracle(mx, port, tmp_path):
    exe = os.path.join(ROOT, "host", "polysynth_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")])
    V, frames = 48, 5000   # 5000 frames: 9 full 512-blocks + a partial one, 4 full 1024 buffers + a tail
    out = tmp_path / "poly.f64"
    r = subprocess.run([exe, str(V), str(frames), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float64).reshape(frames, 2)
    v = np.arange(V)
    freq = np.minimum(20.0 + (v * 97 % 16384) * 0.30517578125, 5000.0)
    cutoff = 200 + 4 * freq
    res = 1.0 + (v % 16)
    par = np.stack([np.full(V, port.env_coeff(0, 10)), np.full(V, port.env_coeff(1, 100)), np.full(V, 0.5),
                    np.full(V, port.env_coeff(2, 500))])
    # the host renders whole 512-frame blocks; the last one runs past `frames`
    nblk = (frames + 511) // 512 * 512
    gate = ((np.arange(nblk) % 4096) < 2048).astype(np.int32)
    voices = port.voice(0, freq, cutoff, res, gate, par, np.ones(V, np.int64))[0][:frames]
    pan = v / (V - 1.0)
    exp = port.mix_stereo(voices, pan)
    assert_bits_equal(got, exp, "polysynth host (per-sample facade, host-side voice sum)")


def test_polysynth_host_with_the_mixdown_on_the_device(mx, port, tmp_path):
    """The same patch with `gpumix`: play() does not sum the voices, the bank renders every block WITH the maxiMix::stereo mixdown fused
    into the voice kernel (maxiVoiceBank::renderMix / mixFrame -> mxg_voice_render_mix) and hands out the frame's two mix values: the
    tree-ordered sum of the same per-voice products, within mix_tol of the reference's voice-order sum; a bank wider than one workgroup."""
    from conftest import mix_tol
    exe = os.path.join(ROOT, "host", "polysynth_host")
    V, frames = 700, 3000
    out = tmp_path / "polymix.f64"
    r = subprocess.run([exe, str(V), str(frames), str(out), "gpumix"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float64).reshape(frames, 2)
    v = np.arange(V)
    freq = np.minimum(20.0 + (v * 97 % 16384) * 0.30517578125, 5000.0)
    par = np.stack([np.full(V, port.env_coeff(0, 10)), np.full(V, port.env_coeff(1, 100)), np.full(V, 0.5),
                    np.full(V, port.env_coeff(2, 500))])
    nblk = (frames + 511) // 512 * 512
    gate = ((np.arange(nblk) % 4096) < 2048).astype(np.int32)
    voices = port.voice(0, freq, 200 + 4 * freq, 1.0 + (v % 16), gate, par, np.ones(V, np.int64))[0][:frames]
    exp = port.mix_stereo(voices, v / (V - 1.0))
    fin = np.isfinite(exp)
    assert np.array_equal(fin, np.isfinite(got))
    assert np.abs(np.where(fin, got - exp, 0.0)).max() <= mix_tol(V, np.abs(np.where(np.isfinite(voices), voices, 0.0)).max(), sums=np.where(fin, exp, 0.0))
    assert np.abs(np.where(fin, exp, 0.0)).max() > 1e-3


def test_facade_smoke_matches_python_mirror(mx, tmp_path):
    """host/facade_smoke.cpp drives the newer C++ facade classes; the Python mirror makes the same C-ABI calls
    on inputs rebuilt with the same exact arithmetic, so every dumped block must be byte-identical."""
    exe = os.path.join(ROOT, "host", "facade_smoke")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")])
    wav = os.path.join(ROOT, "tests", "golden", "wav", "mono.wav")
    r = subprocess.run([exe, wav, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rd = lambda name, dt=np.float64: np.fromfile(tmp_path / (name + ".bin"), dt)
    mx.maxiSettings.setup(44100, 2, 512)
    try:
        V, N = 96, 300
        i = np.arange(N * V)
        x = (((i * 37) % 1000) / 1000.0 - 0.5) * 1.6
        v = np.arange(V)
        cutoff, q, gain, R = 100.0 + 37.0 * v, 0.5 + 0.05 * v, -12.0 + 0.25 * v, 0.99 + 0.0001 * v
        px = v / (V - 1)
        py = 1.0 - px * 0.5
        dx = mx.DeviceBuffer.from_numpy(x.reshape(N, V))
        svf = mx.maxiSVFBank(V); svf.setCutoff(cutoff); svf.setResonance(q)
        assert_bits_equal(svf.play(dx, 0.5, 0.25, 0.125, 1.0).numpy().ravel(), rd("svf"), "svf")
        bq = mx.maxiBiquadBank(V); bq.set(bq.PEAK, cutoff, q, gain)
        assert_bits_equal(bq.play(dx).numpy().ravel(), rd("biquad"), "biquad")
        assert_bits_equal(mx.maxiDCBlockerBank(V).play(dx, R).numpy().ravel(), rd("dcblock"), "dcblock")
        eg = mx.maxiEnvGenBank(V); eg.setupADSR(2, 3, 0.5, 4)
        gate = np.where(np.arange(N) % 200 < 120, 1.0, -1.0)
        assert_bits_equal(eg.play(gate).numpy().ravel(), rd("envgen"), "envgen")
        assert_bits_equal(mx.maxiMixBank(V).quad(dx, px, py).numpy().ravel(), rd("quad"), "quad")
        sb = mx.maxiSampleBank(V)
        assert sb.load(wav)
        n = np.arange(N)[:, None]
        trig = ((n * 3 + v[None, :] * 7) % 64) / 32.0 - 1.0
        assert_bits_equal(sb.playOnZX(trig).numpy().ravel(), rd("playonzx"), "playOnZX")
        fs, hop, nfr = 256, 64, 40
        k = np.arange(fs + hop * (nfr - 1))
        sig = (((k * 29) % 200) / 200.0 - 0.5).astype(np.float32)
        f = mx.maxiFFT(); f.setup(fs, hop, fs)
        f.process_frames(mx.DeviceBuffer.from_numpy(sig), hop, nfr)
        assert np.array_equal(f.getMagnitudes().numpy().ravel().view(np.uint32), rd("mags", np.float32).view(np.uint32))
        assert np.array_equal(f.spectralCentroid().numpy().view(np.uint32), rd("centroid", np.float32).view(np.uint32))
        inv = mx.maxiIFFT(); inv.setup(fs, hop, fs)
        y = inv.process_frames(f.getMagnitudes(), f.getPhases()).numpy()
        assert np.array_equal(y.view(np.uint32), rd("resynth", np.float32).view(np.uint32))
        one = mx.maxiSampleBank(1)
        assert one.load(wav)
        ps = mx.maxiPitchShiftBank(32, one, "hann")
        assert_bits_equal(ps.play(0.5 + 0.05 * np.arange(32), 0.01, 3, 1500).numpy().ravel(), rd("pitchshift"), "pitchshift")
        # round 3: maxiStretchBank, maxiConvolveBlock, maxiSamplerBank of include/maximilian_bank.hpp
        sk = mx.maxiStretchBank(32, one, "triangle")
        assert_bits_equal(sk.play(0.5 + 0.05 * np.arange(32), 0.25 + 0.05 * np.arange(32), 0.02, 2, 1500).numpy().ravel(), rd("stretch"),
                          "stretch")
        cv = mx.maxiConvolve()
        cv.setup(one.amplitudes(), fs, hop)
        cin = (((np.arange(6 * fs) * 13) % 97) / 97.0 - 0.5).astype(np.float32)
        assert np.array_equal(cv.play(cin, mode=1).numpy().view(np.uint32), rd("convolve", np.float32).view(np.uint32))
        sp = mx.maxiSamplerBank(3, 4)
        assert sp.load(wav)
        for k in range(3):                      # midiNoteOn / trigger on sampler k only (the bank's methods act on every sampler's
            sp.pitch[k * 4 + sp.currentVoice[k]] = -2.0 + 3.0 * k   # current slot at once: set the slots directly)
            sp.gain[k * 4 + sp.currentVoice[k]] = (100.0 + k) / 128
            sp._pull()
            sl = k * 4 + sp.currentVoice[k]
            sp.trigger_state[sl] = 1
            sp.position[sl] = 0.0
            sp.currentVoice[k] = (sp.currentVoice[k] + 1) % 4
            sp._state_dirty = True
        a = sp.play(300).numpy()
        sp._pull()
        for i in range(4):                      # midiNoteOff(1, 1.0)
            if sp.pitch[4 + i] == 1.0:
                sp.trigger_state[4 + i] = 0
        sp._state_dirty = True
        sl = 0 * 4 + sp.currentVoice[0]         # midiNoteOn(0, 4.0, 64.0); trigger(0)
        sp.pitch[sl] = 4.0
        sp.gain[sl] = 64.0 / 128
        sp.trigger_state[sl] = 1
        sp.position[sl] = 0.0
        sp.currentVoice[0] = (sp.currentVoice[0] + 1) % 4
        b = sp.play(600).numpy()
        assert_bits_equal(np.concatenate([a, b]).ravel(), rd("sampler"), "sampler bank")
    finally:
        mx.maxiSettings.setup(44100, 2, 1024)
