"""CPU (-m "not gpu"): maxiOsc's waveforms as the device runs them per lane (mxg_osc.h) compiled for the host and compared
bit for bit with the oracle.  The ramp family (phasor, saw, triangle, square, pulse, impulse, phasorBetween) is plain
compare-and-add arithmetic, defined for ANY phase / held output and any frequency sign, so it starts from arbitrary
states; the table oscillators (sinebuf, sinebuf4, sawn) start anywhere inside the range their own wrap keeps them in."""
import ctypes
import os
import subprocess

import numpy as np
import pytest


from conftest import HOST_OPT, ROOT, assert_bits_equal

NAMES = {2: "phasor", 3: "saw", 4: "triangle", 5: "square", 6: "pulse", 7: "impulse", 8: "sinebuf", 9: "sinebuf4",
         10: "sawn", 11: "phasorBetween"}


@pytest.fixture(scope="module")
def osc_host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("osc") / "libosc_host.so")
    subprocess.check_call(["g++", "-std=c++17"] + HOST_OPT + ["-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_osc.cpp")])
    lib = ctypes.CDLL(so)
    lib.osc_host.restype = ctypes.c_int
    lib.osc_host.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int] + [ctypes.c_void_p] * 6
    return lib


@pytest.mark.parametrize("wf", sorted(NAMES))
def test_waveforms_on_host_match_oracle(osc_host, port, wf):
    rng = np.random.default_rng(700 + wf)
    V, N = 20000, 300
    if wf in (8, 9, 10):                      # table oscillators: inside their own invariants (a3, a4, a6)
        freq = rng.uniform(0.0, 22050.0, V)
        freq[:3] = [0.0, 22050.0, 440.0]
        if wf == 8:
            ph0 = rng.uniform(-1.0, 511.0, V)
        elif wf == 9:
            ph0 = rng.uniform(0.0, 511.0, V)  # (-1, 0) makes sinebuf4 read sineBuffer[-2]: undefined in the reference
            freq = rng.uniform(0.0, 43.0, V)  # one table step per sample at most, so the wrap never lands in (-1, 0)
            freq[:2] = [0.0, 43.06640625]
        else:
            ph0 = rng.uniform(-0.5, 0.5, V)
            freq = rng.uniform(20.0, 22050.0, V)
        hold0 = rng.uniform(-1, 1, V)
    else:                                     # ramps: any state, either direction
        freq = rng.uniform(-30000.0, 60000.0, V)
        freq[:4] = [0.0, 44100.0, -44100.0, 1e-9]
        ph0 = rng.uniform(-3.0, 3.0, V)
        ph0[4:8] = [0.0, 0.5, 1.0, -1.0]
        hold0 = rng.uniform(-2, 2, V)
    p1 = rng.uniform(-0.3, 1.3, V) if wf == 6 else rng.uniform(-1.0, 0.5, V)
    p2 = p1 + rng.uniform(-0.5, 2.0, V)
    ph, hd, out = ph0.copy(), hold0.copy(), np.empty((N, V))
    rc = osc_host.osc_host(wf, V, N, 44100, freq.ctypes.data, p1.ctypes.data, p2.ctypes.data, ph.ctypes.data,
                           hd.ctypes.data, out.ctypes.data)
    assert rc == 0
    e, eph, ehd = port.osc(wf, freq, N, phase=ph0, hold=hold0, p1=p1, p2=p2)
    assert_bits_equal(out, e, NAMES[wf])
    assert_bits_equal(ph, eph, NAMES[wf] + " phase")
    assert_bits_equal(hd, ehd, NAMES[wf] + " held output")
