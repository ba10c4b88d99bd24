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
