"""GPU parity (-m gpu): mxg_sample_load_wav / mxg_sample_save_wav against the reference's results for the
fixtures in tests/golden/wav/ (tests/golden/wav.npz)."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

WAVDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wav")
CASES = [("mono", 0), ("list", 0), ("fmt18", 0), ("stereo", 0), ("stereo", 1)]


@pytest.mark.parametrize("name,ch", CASES)
def test_load_wav_golden(mx, golden, name, ch):
    g = golden("wav.npz")
    bank = mx.maxiSampleBank(3)
    assert bank.load(os.path.join(WAVDIR, name + ".wav"), ch) is True
    e = g["amp_%s_%d" % (name, ch)]
    amp = bank.amplitudes()
    assert bank.getLength() == e.size
    assert np.array_equal(bank.wav_header, g["hdr_%s_%d" % (name, ch)])
    assert bank.mySampleRate == int(g["hdr_%s_%d" % (name, ch)][4])
    assert np.array_equal(bank.position.numpy(), np.full(3, float(e.size)))   # C:681
    if name == "stereo":
        k = int(g["defined_%s_%d" % (name, ch)])
        assert_bits_equal(amp[:k], e[:k])
        tail = (2 * amp.size + 6 - 2 * ch + 3) // 4
        assert_bits_equal(amp[tail:], e[tail:])
    else:
        assert_bits_equal(amp, e)
    # the loaded buffer drives the players: first play() after a load wraps to sample 0
    o = bank.play(4).numpy()
    assert_bits_equal(o[1:, 0], amp[:3])


def test_load_wav_errors(mx, tmp_path):
    bank = mx.maxiSampleBank(1)
    assert bank.load("/nonexistent/file.wav") is False
    assert b"cannot open" in mx.lib().mxg_last_error()
    (tmp_path / "short.wav").write_bytes(b"RIFF1234WAVE")
    assert bank.load(str(tmp_path / "short.wav")) is False
    (tmp_path / "nodata.wav").write_bytes(b"RIFF" + bytes(60))
    assert bank.load(str(tmp_path / "nodata.wav")) is False


def test_save_wav_golden(mx, golden, tmp_path):
    g = golden("wav.npz")
    bank = mx.maxiSampleBank(1)
    bank.setSample(g["save_amp"])
    bank.wav_header = g["save_hdr"]
    out = tmp_path / "gpu.wav"
    assert bank.save(str(out))
    assert out.read_bytes() == open(os.path.join(WAVDIR, "saved_by_reference.wav"), "rb").read()
    # round trip: load what was saved, quantisation error <= half an LSB
    b2 = mx.maxiSampleBank(1)
    assert b2.load(str(out))
    assert np.abs(b2.amplitudes() - g["save_amp"]).max() <= 0.5 / 32767 + 1e-12


def test_save_wav_out_of_range(mx, port, tmp_path):
    """static_cast<short>(round(a*32767)) outside [-1, 1]: the x86 conversion wraps modulo 2^16 (or gives 0
    beyond int32 / NaN); same bytes as the oracle on the host."""
    amp = np.array([1.5, -1.7, 3.0, 70000.0, -70000.0, 1e12, -1e12, np.nan, np.inf, 0.0])
    hdr = np.array([36 + 20, 16, 1, 1, 44100, 88200, 2, 16], np.int32)
    bank = mx.maxiSampleBank(1)
    bank.setSample(amp)
    bank.wav_header = hdr
    bank.save(str(tmp_path / "g.wav"))
    port.wav_save(str(tmp_path / "p.wav"), amp, hdr)
    assert (tmp_path / "g.wav").read_bytes() == (tmp_path / "p.wav").read_bytes()
