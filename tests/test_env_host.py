"""CPU (-m "not gpu"): maxiEnv as the device runs it per lane (mxg_env.h) compiled for the host and compared bit for bit
with the oracle from ARBITRARY states -- phase flags drawn from {0, 1, 2} in any combination, any amplitude / holdcount --
under gates that toggle at random, both through the state machine alone and with the steady-state sustain / release ticks
taken whenever their entry tests hold (the tests must only admit states the machine itself would leave unchanged)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal


@pytest.fixture(scope="module")
def env_host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("env") / "libenv_host.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_env.cpp")])
    lib = ctypes.CDLL(so)
    lib.env_host.restype = ctypes.c_int
    lib.env_host.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_int] + [ctypes.c_void_p] * 5
    return lib


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("seed", range(4))
def test_env_on_host_matches_oracle_from_arbitrary_states(env_host, port, mode, fast, seed):
    rng = np.random.default_rng(100 * mode + 10 * fast + seed)
    V, N = 20000, 120
    x = rng.uniform(-1, 1, (N, V))
    par = np.stack([rng.uniform(1e-4, 0.3, V), rng.uniform(0.9, 0.99999, V), rng.uniform(0.0, 1.0, V),
                    rng.uniform(0.9, 0.99999, V)])
    hold = rng.integers(0, 60, V).astype(np.int64)
    d0 = np.stack([rng.uniform(-0.1, 1.3, V), rng.uniform(-1, 1, V)])
    d0[0, ::7] = 0.0
    i0 = np.concatenate([rng.integers(0, 90, (1, V)), rng.integers(0, 3, (5, V))]).astype(np.int64)
    # gates: held, released, toggling every few samples, and values that are neither 0 nor 1
    trig = np.zeros((N, V), np.int32)
    trig[:, 0::4] = 1
    trig[:, 2::4] = (rng.uniform(size=(N, (V + 1) // 4 if V % 4 > 2 else V // 4)) < 0.5)
    trig[:, 3::4] = rng.integers(-1, 3, (N, V // 4))
    dst, ist, out = d0.copy(), i0.copy(), np.empty((N, V))
    rc = env_host.env_host(mode, fast, V, N, x.ctypes.data, trig.ctypes.data, 1, par.ctypes.data, hold.ctypes.data,
                           dst.ctypes.data, ist.ctypes.data, out.ctypes.data)
    assert rc == 0
    e, ed, ei = port.env(mode, x, trig, par, hold, dstate=d0, istate=i0)
    assert_bits_equal(out, e, "env mode %d fast %d" % (mode, fast))
    assert_bits_equal(dst, ed, "amplitude / output")
    assert np.array_equal(ist, ei), "holdcount / flags"
