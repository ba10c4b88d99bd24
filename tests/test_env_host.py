"""CPU (-m "not gpu"): maxiEnv as the device runs it per lane (mxg_env.h) compiled for the host and compared bit for bit
with the oracle from ARBITRARY states -- phase flags drawn from {0, 1, 2} in any combination, any amplitude / holdcount --
under gates that toggle at random, both through the state machine alone and with the steady-state sustain / release ticks
taken whenever their entry tests hold (the tests must only admit states the machine itself would leave unchanged)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest


from conftest import HOST_OPT, ROOT, assert_bits_equal


@pytest.fixture(scope="module")
def env_host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("env") / "libenv_host.so")
    subprocess.check_call(["g++", "-std=c++17"] + HOST_OPT + ["-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_env.cpp")])
    lib = ctypes.CDLL(so)
    lib.env_host.restype = ctypes.c_int
    lib.env_host.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_int] + [ctypes.c_void_p] * 5
    return lib


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("fast", [0, 1, 2])
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
    assert rc >= 0
    e, ed, ei = port.env(mode, x, trig, par, hold, dstate=d0, istate=i0)
    assert_bits_equal(out, e, "env mode %d fast %d" % (mode, fast))
    assert_bits_equal(dst, ed, "amplitude / output")
    assert np.array_equal(ist, ei), "holdcount / flags"


@pytest.mark.parametrize("seed", range(3))
def test_env_steady_chunks_over_long_gates(env_host, port, seed):
    """env_steady_chunk (the general steady chunk of the ADSR: attack / decay / hold / sustain / release / idle, each lane in its
    own stage) over gates that stay constant for tens to hundreds of samples: stage exits anywhere inside a chunk (the chunk
    must be refused and redone by the state machine), saturating hold counts, amplitudes reaching exactly 0 or 1, slow and fast
    envelopes, zero hold times, parameters outside (0, 1]."""
    rng = np.random.default_rng(4242 + seed)
    V, N = 6000, 1200
    x = rng.uniform(-1, 1, (N, V))
    par = np.stack([10.0 ** rng.uniform(-3.5, -0.5, V), 1.0 - 10.0 ** rng.uniform(-4, -1, V), rng.uniform(0.0, 0.9, V),
                    1.0 - 10.0 ** rng.uniform(-4, -1, V)])
    par[1, ::50] = rng.uniform(1.0, 1.01, par[1, ::50].size)      # decay >= 1: never steady in decay
    par[3, 1::50] = 0.0                                            # release 0: amplitude hits 0 at once
    par[0, 2::50] = 0.0                                            # attack 0: stuck in attack
    par[3, 3::50] = 1e-160                                         # amplitude underflows to 0 within a few samples
    hold = rng.integers(0, 300, V).astype(np.int64)
    hold[::9] = 0
    d0 = np.stack([rng.uniform(0.0, 1.0, V), rng.uniform(-1, 1, V)])
    d0[0, ::7] = 0.0
    i0 = np.zeros((6, V), np.int64)
    if seed == 2:                                                  # arbitrary flags: the entry tests must sort them out
        d0[0] = rng.uniform(-0.1, 1.3, V)
        i0 = np.concatenate([rng.integers(0, 300, (1, V)), rng.integers(0, 3, (5, V))]).astype(np.int64)
    period = rng.integers(40, 500, V)
    duty = rng.uniform(0.1, 0.9, V)
    n = np.arange(N)[:, None]
    trig = (((n + rng.integers(0, 500, V)[None, :]) % period[None, :]) < (duty * period)[None, :]).astype(np.int32)
    trig[:, ::11] *= 2                                             # a gate value that is neither 0 nor 1 counts as off
    dst, ist, out = d0.copy(), i0.copy(), np.empty((N, V))
    rc = env_host.env_host(0, 2, V, N, x.ctypes.data, trig.ctypes.data, 1, par.ctypes.data, hold.ctypes.data,
                           dst.ctypes.data, ist.ctypes.data, out.ctypes.data)
    print("steady chunks: %.1f %% of all chunks" % (rc / 10.0))
    assert rc >= 500, "the test is meant to spend most chunks on the steady path"
    e, ed, ei = port.env(0, x, trig, par, hold, dstate=d0, istate=i0)
    assert_bits_equal(out, e, "general steady chunk")
    assert_bits_equal(dst, ed, "amplitude / output")
    assert np.array_equal(ist, ei), "holdcount / flags"
