"""CPU (-m "not gpu"): maxiEnvGen::play as the device runs it per lane (mxg_envgen.h) compiled for the host and compared
with the oracle from arbitrary states -- any of the three play states, any stage, any counter / level, detector history
of either sign, first-trigger flags set or not -- under random per-voice triggers, for every loop / retrigger setting."""
import ctypes
import os
import subprocess

import numpy as np
import pytest


from conftest import HOST_OPT, ROOT, assert_bits_equal

H = -46692.0
CASES = {
    "AR": ([0, 1, 0], [10, 40], [1, 1]),
    "ADSR": ([0, 1, 0.4, 0.4, 0], [3, 12, H, 25], [1, 1, 1, 1]),
    "curved": ([0, 1, 0.2, 0], [7.3, 11.1, 20.7], [0.5, 2, 3]),
}


@pytest.fixture(scope="module")
def eg_host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("eg") / "libenvgen_host.so")
    subprocess.check_call(["g++", "-std=c++17"] + HOST_OPT + ["-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_envgen.cpp")])
    lib = ctypes.CDLL(so)
    lib.envgen_host.restype = ctypes.c_int
    lib.envgen_host.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return lib


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("loop,retrig", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("fast", [0, 1])
def test_envgen_on_host_matches_oracle_from_arbitrary_states(eg_host, port, name, loop, retrig, fast):
    lv, tm, cv = CASES[name]
    S = len(tm)
    rng = np.random.default_rng(hash((name, loop, retrig)) % 2 ** 32)
    V, N = 6000, 400
    # triggers: sign flips at random rates, exact zeros, constant voices
    trig = np.sign(np.sin(np.arange(N)[:, None] * rng.uniform(0.01, 0.4, V)[None, :] + rng.uniform(0, 6, V)))
    trig[:, ::11] = 1.0
    trig[:, 1::11] = 0.0
    trig[:, 2::11] *= rng.uniform(0.1, 3.0, (N, len(range(2, V, 11))))
    d0 = np.zeros((5, V))
    d0[0] = rng.uniform(0, 1, V)                         # envval
    d0[1] = rng.uniform(0, 1.1, V)                       # currentlevel of the current stage
    d0[2:5] = rng.choice([-1.0, 0.0, 0.5, 1.0], (3, V))  # detector history
    i0 = np.zeros((7, V), np.int64)
    i0[0] = rng.integers(0, S, V)                        # phase (a valid stage)
    i0[0, ::17] = S                                      # ... or parked on the end-of-envelope test (an uploaded state, H:2349-2355)
    i0[1] = rng.integers(0, 3, V)                        # WAITING / TRIGGERED / HOLDING
    i0[1, ::17] = 0                                      # (WAITING, and not triggered on the first sample: any other way the
    trig[0, ::17] = -1.0                                 # reference would index past its stage vector before the test resets it)
    i0[2] = rng.integers(0, 2, V)                        # nxcHappened
    i0[3] = rng.integers(0, 30, V)                       # counter
    i0[4:7] = rng.integers(0, 2, (3, V))                 # firstTrigger flags
    _, _, _, stages = port.envgen(trig[:1], lv, tm, cv, loop, retrig)
    stages = np.ascontiguousarray(stages)
    dst, ist, out = d0.copy(), i0.copy(), np.empty((N, V))
    rc = eg_host.envgen_host(V, N, trig.ctypes.data, 1, stages.ctypes.data, S, loop, retrig, dst.ctypes.data,
                             ist.ctypes.data, out.ctypes.data, fast)
    assert rc >= 0
    e, ed, ei, _ = port.envgen(trig, lv, tm, cv, loop, retrig, dst=d0, ist=i0)
    assert_bits_equal(out, e, name)                       # (host pow == the oracle's pow: the curved case is exact here too)
    assert np.array_equal(ist, ei), "phase / state / nxc / counter / firstTrigger"
    assert_bits_equal(dst, ed, "envval, currentlevel, detector history")


@pytest.mark.parametrize("loop,retrig", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_envgen_steady_chunks_over_long_stages(eg_host, port, loop, retrig):
    """envgen_steady_chunk (ramp / hold / wait, each lane in its own) on envelopes whose stages last tens to thousands of
    samples under gates that stay put for a while: stage ends, hold releases, triggers and retriggers anywhere inside a
    chunk, counters close to the stage length, exact-zero gate values."""
    rng = np.random.default_rng(31 + 2 * loop + retrig)
    lv, tm, cv = [0, 1, 0.45, 0.45, 0], [4.0, 21.3, H, 37.7], [1, 1, 1, 1]
    S = len(tm)
    V, N = 3000, 2400
    period = rng.integers(60, 1500, V)
    n = np.arange(N)[:, None]
    trig = np.where(((n + rng.integers(0, 1500, V)[None, :]) % period[None, :]) < (rng.uniform(0.05, 0.95, V) * period)[None, :],
                    1.0, -1.0)
    trig[:, ::13] = np.where(trig[:, ::13] > 0, 0.7, 0.0)           # gates that rest on exactly 0
    d0 = np.zeros((5, V)); i0 = np.zeros((7, V), np.int64)
    i0[4:7] = 1                                                        # fresh detectors (firstTrigger set)
    _, _, _, stages = port.envgen(trig[:1], lv, tm, cv, loop, retrig)
    stages = np.ascontiguousarray(stages)
    dst, ist, out = d0.copy(), i0.copy(), np.empty((N, V))
    rc = eg_host.envgen_host(V, N, trig.ctypes.data, 1, stages.ctypes.data, S, loop, retrig, dst.ctypes.data,
                             ist.ctypes.data, out.ctypes.data, 1)
    print("steady chunks: %.1f %%" % (rc / 10.0))
    assert rc >= 500
    e, ed, ei, _ = port.envgen(trig, lv, tm, cv, loop, retrig, dst=d0, ist=i0)
    assert_bits_equal(out, e, "steady chunks")
    assert np.array_equal(ist, ei), "phase / state / nxc / counter / firstTrigger"
    assert_bits_equal(dst, ed, "envval, currentlevel, detector history")
