"""CPU (-m "not gpu"): the maxiSample players as the device runs them per lane (mxg_smp.h: smp_gen / smp_eval), compiled
for the host (tests/host_smp.cpp) and compared bit for bit with the oracle over many more voices, buffer lengths, block
lengths, speeds (incl. reverse and per-sample modulation) and loop points than the GPU parity tests afford."""
import ctypes
import os
import subprocess

import numpy as np
import pytest


from conftest import HOST_OPT, ROOT, assert_bits_equal

MODES = ["play", "playOnce", "playLoop", "playUntil", "playAtSpeed", "playOnceAtSpeed", "playUntilAtSpeed", "play4",
         "playAtSpeedBetweenPoints"]


@pytest.fixture(scope="module")
def smp_host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("smp") / "libsmp_host.so")
    subprocess.check_call(["g++", "-std=c++17"] + HOST_OPT + ["-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_smp.cpp")])
    lib = ctypes.CDLL(so)
    lib.smp_host.restype = ctypes.c_int
    lib.smp_host.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                             ctypes.c_int] + [ctypes.c_void_p] * 1 + [ctypes.c_int] + [ctypes.c_void_p] * 4
    return lib


def _p(a):
    return None if a is None else a.ctypes.data


@pytest.mark.parametrize("mode", range(9))
@pytest.mark.parametrize("Ls", [64, 1500, 40001])
def test_players_on_host_match_oracle(smp_host, port, mode, Ls):
    rng = np.random.default_rng(1000 * mode + Ls)
    V = 3000
    smp = rng.uniform(-1, 1, Ls)
    g = port.guarded(smp)                     # [0, samples..., 0, 0]: the guards the upload adds on the device
    for N, per_sample in ((1, False), (37, False), (300, False), (300, True)):
        pos0 = rng.uniform(1, Ls - 5, V)
        pos0[:4] = [Ls - 1.0, 0.0, 1.0, Ls - 2.5]
        if mode in (0, 1):
            pos0 = np.floor(pos0)
        a = rng.uniform(0.05, 4.0, V)
        st, en = rng.uniform(0, 0.45, V), rng.uniform(0.5, 1.1, V)
        if mode == 2:
            en = np.minimum(en, 1.0)          # playLoop does not clamp `end` (C:960-967): > 1 reads past the buffer
        if mode in (7, 8):                    # frequency-driven loops between two absolute points, both directions
            lo = np.floor(rng.uniform(2, max(3, 0.3 * Ls), V))
            st, en = lo, np.minimum(lo + np.floor(rng.uniform(8, 0.6 * Ls, V)), Ls - 1.0)
            # play4 (C:884-956) reads buffer[(long)position] unguarded: it is defined while one step, (end-start)*f/sr,
            # stays below two samples (the head then never leaves [start-2, end+2)); beyond that the reference reads
            # outside its vector.  Keep the step <= 1.2 (x1.5 with per-sample modulation).
            a = rng.uniform(0.05, 1.2, V) * 44100.0 / (en - st) * np.where(rng.uniform(0, 1, V) < 0.4, -1, 1)
            if mode == 8:                     # playAtSpeedBetweenPoints guards every index: any frequency is defined
                a = rng.uniform(0.5, 90, V) * np.where(rng.uniform(0, 1, V) < 0.4, -1, 1)
        aps = per_sample and mode >= 4
        if aps:
            a = a[None, :] * rng.uniform(0.5, 1.5, (N, V))
            if mode >= 7:
                a = a * np.where(rng.uniform(size=(N, V)) < 0.15, -1.0, 1.0)   # direction flips
        a = np.ascontiguousarray(a)
        pos = pos0.copy()
        out = np.empty((N, V))
        rc = smp_host.smp_host(mode, V, N, g.ctypes.data + 8, Ls, 44100, 44100, _p(a), int(aps), _p(st), _p(en),
                               _p(pos), _p(out))
        assert rc == 0
        e, ep = port.sample(mode, smp, N, pos0, a=a, start=st, end=en, aps=aps)
        assert_bits_equal(out, e, "%s Ls=%d N=%d aps=%d" % (MODES[mode], Ls, N, aps))
        assert_bits_equal(pos, ep, "%s position" % MODES[mode])
