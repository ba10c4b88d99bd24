"""CPU tests (-m "not gpu"): the oracle's restatement of maxiSample::load/read/save (C:605-725) against the
values the compiled reference produced for the synthesised fixtures in tests/golden/wav/ (oracle/gen_golden.py),
and against the compiled reference itself where oracle/_ref exists."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

WAVDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wav")
CASES = [("mono", 0), ("list", 0), ("fmt18", 0), ("stereo", 0), ("stereo", 1)]


@pytest.mark.parametrize("name,ch", CASES)
def test_wav_load_golden(port, golden, name, ch):
    g = golden("wav.npz")
    amp, hdr, pos = port.wav_load(os.path.join(WAVDIR, name + ".wav"), ch)
    e = g["amp_%s_%d" % (name, ch)]
    assert amp.size == e.size and pos == float(g["pos_%s_%d" % (name, ch)]) == amp.size   # C:681
    assert np.array_equal(hdr, g["hdr_%s_%d" % (name, ch)])
    if name == "stereo":
        # C:667-674: every 4th short lands at the front; the reference then reads past its vector, and
        # the tail keeps the interleaved data
        k = int(g["defined_%s_%d" % (name, ch)])
        assert_bits_equal(amp[:k], e[:k])
        tail = (2 * amp.size + 6 - 2 * ch + 3) // 4
        assert_bits_equal(amp[tail:], e[tail:])
    else:
        assert_bits_equal(amp, e)
        assert amp[0] == 1.0 and amp[1] == -32768 / 32767.0


def test_wav_save_golden(port, golden, tmp_path):
    g = golden("wav.npz")
    out = tmp_path / "port.wav"
    assert port.wav_save(str(out), g["save_amp"], g["save_hdr"]) == 0
    assert out.read_bytes() == open(os.path.join(WAVDIR, "saved_by_reference.wav"), "rb").read()
    amp, hdr, _ = port.wav_load(str(out))
    assert np.array_equal(hdr, g["save_hdr"])
    assert np.abs(amp - g["save_amp"]).max() <= 0.5 / 32767 + 1e-12


def test_wav_missing_file(port):
    assert port.wav_load("/nonexistent/file.wav") is None


def test_wav_port_vs_reference(port, ref, tmp_path):
    import struct
    rng = np.random.default_rng(8)
    for channels in (1, 2, 3):
        data = (rng.uniform(-1, 1, 999 * channels) * 32767).astype("<i2").tobytes()
        fmt = struct.pack("<HHIIHH", 1, channels, 48000, 48000 * channels * 2, channels * 2, 16)
        body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(data)) + data
        p = tmp_path / ("c%d.wav" % channels)
        p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
        for ch in range(channels):
            a, b = port.wav_load(str(p), ch), ref.wav_load(str(p), ch)
            n = a[0].size
            assert n == b[0].size and np.array_equal(a[1], b[1]) and a[2] == b[2]
            k = len(range(ch * 2, n, channels * 2)) if channels > 1 else n
            assert_bits_equal(a[0][:k], b[0][:k])
    amp = rng.uniform(-1.2, 1.2, 500)
    hdr = np.array([1036, 16, 1, 1, 8000, 16000, 2, 16], np.int32)
    port.wav_save(str(tmp_path / "p.wav"), amp, hdr)
    ref.wav_save(str(tmp_path / "r.wav"), amp, hdr)
    assert (tmp_path / "p.wav").read_bytes() == (tmp_path / "r.wav").read_bytes()
