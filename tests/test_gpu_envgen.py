"""GPU parity (-m gpu): maxiEnvGen bank (H:2268-2547) through the C-ABI vs the oracle.  The state machine
(phase, state, counters, the three zero-crossing detectors) is bit-exact; the value is bit-exact for
curve == 1 (setupAR/ASR/ADSR) and within 4 ULP-of-1.0 x level range for other curves (device pow)."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

H = -46692.0
CASES = {
    "AR": ([0, 1, 0], [10, 40], [1, 1]),
    "ASR": ([0, 1, 1, 0], [5, H, 30], [1, 1, 1]),
    "ADSR": ([0, 1, 0.4, 0.4, 0], [3, 12, H, 25], [1, 1, 1, 1]),
    "curved": ([0, 1, 0.2, 0], [7.3, 11.1, 20.7], [0.5, 2, 3]),
}


def _trig(V, N, seed):
    rng = np.random.default_rng(seed)
    n = np.arange(N)[:, None]
    t = np.sign(np.sin(n * rng.uniform(0.002, 0.01, V)[None, :] + rng.uniform(0, 6, V)))
    t[:, 3] = 1.0                                  # constant 1: fires once (firstTrigger), never releases
    t[:, 4] = (np.arange(N) % 700 < 5) * 1.0       # short impulses with exact zeros between
    t[:, 5] = 0.0
    return t


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("loop,retrig", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_envgen_per_voice_triggers(mx, port, name, loop, retrig):
    lv, tm, cv = CASES[name]
    V, N = 300, 3001
    trig = _trig(V, 2 * N, 80)
    bank = mx.maxiEnvGenBank(V)
    assert bank.setup(lv, tm, cv, bool(loop), bool(retrig))
    o = np.concatenate([bank.play(trig[:N]).numpy(), bank.play(trig[N:]).numpy()])
    e1, d, i, stages = port.envgen(trig[:N], lv, tm, cv, loop, retrig)
    e2, d, i, _ = port.envgen(trig[N:], lv, tm, cv, loop, retrig, dst=d, ist=i)
    e = np.concatenate([e1, e2])
    assert_bits_equal(bank.host_stages, stages, "stage table")
    assert np.array_equal(bank.istate.numpy(), i), "phase/state/nxc/counter/firstTrigger"
    assert_bits_equal(bank.dstate.numpy()[2:], d[2:], "detector previousValue")
    if name == "curved":
        tol = 4 * 2.2e-16 * 1.0
        assert np.abs(o - e).max() <= tol
        assert np.abs(bank.dstate.numpy()[:2] - d[:2]).max() <= tol
    else:
        assert_bits_equal(o, e, name)
        assert_bits_equal(bank.dstate.numpy()[:2], d[:2], "envval, currentlevel")
    assert e.max() == 1.0 and (e[:, 5] == 0).all()


def test_envgen_shared_gate_and_helpers(mx, port):
    V, N = 130, 5000
    gate = ((np.arange(N) % 1800) < 900) * 2.0 - 1.0
    bank = mx.maxiEnvGenBank(V)
    assert bank.setupADSR(2, 8, 0.3, 20)
    o = bank.play(gate).numpy()
    e, d, i, _ = port.envgen(gate, [0, 1, 0.3, 0.3, 0], [2, 8, H, 20], [1, 1, 1, 1], V=V)
    assert_bits_equal(o, e, "shared gate ADSR")
    assert np.array_equal(bank.istate.numpy(), i)
    assert bank.setupAR(1, 2) and bank.setupASR(1, 2)
    assert bank.setup([0, 1], [1, 2], [1], False) is False                 # size mismatch (H:2395)
    assert bank.setup([0, 1, 1, 0], [H, H, 5], [1, 1, 1], False) is False  # two HOLD stages (H:2377-2381)


@pytest.mark.parametrize("shape", ["AR", "ADSR"])
def test_envgen_shared_gate_steady_states(mx, port, shape):
    """The wave-uniform steady-state paths with a shared gate: HOLDING under a held gate, WAITING under a low gate, and
    WAITING under a gate that stays up after the envelope has finished (no zero crossing => no retrigger), across
    launches of 64-chunk groups and ragged lengths."""
    V, N = 200, 12000
    gate = np.ones(N)
    gate[:37] = -1.0
    gate[6000:6100] = 0.0          # exact zeros: a release / a re-arm
    gate[9000:9003] = -2.5
    lv, tm, cv = CASES[shape]
    bank = mx.maxiEnvGenBank(V)
    assert bank.setup(lv, tm, cv, False, False)
    cuts = [0, 5, 517, 1029, 6050, 9001, N]
    o = np.concatenate([bank.play(gate[a:b]).numpy() for a, b in zip(cuts[:-1], cuts[1:])])
    e, d, i, _ = port.envgen(gate, lv, tm, cv, 0, 0, V=V)
    assert_bits_equal(o, e, shape)
    assert np.array_equal(bank.istate.numpy(), i)
    assert_bits_equal(bank.dstate.numpy(), d)


@pytest.mark.parametrize("loop", [0, 1])
def test_envgen_uploaded_state_parked_on_the_end_test(mx, port, loop):
    """The end-of-envelope test `phase == stages.size` (H:2349-2355) runs after the switch on every sample, whatever the
    state.  A host-uploaded state that is WAITING with phase already on the end must be reset / re-armed by it, also
    under a shared gate that keeps every wavefront on the steady-state test."""
    lv, tm, cv = CASES["ADSR"]
    V, N = 192, 700
    bank = mx.maxiEnvGenBank(V)
    assert bank.setup(lv, tm, cv, bool(loop), False)
    d0, i0 = port.envgen_fresh(V)
    rng = np.random.default_rng(99)
    i0[0, :128] = 4                       # phase == number of stages, state WAITING (two whole wavefronts)
    d0[0, :128] = rng.uniform(0, 1, 128)  # some envval left over
    i0[4, ::3] = 0                        # trigDetector.firstTrigger already consumed on a third of the voices
    d0[2, ::3] = rng.choice([-1.0, 0.5], V)[::3]
    bank.dstate.upload(d0); bank.istate.upload(i0)
    gate = -np.ones(N)
    gate[300:] = 1.0
    o = bank.play(gate).numpy()
    e, d, i, _ = port.envgen(gate, lv, tm, cv, loop, 0, dst=d0, ist=i0, V=V)
    assert_bits_equal(o, e, "uploaded state")
    assert np.array_equal(bank.istate.numpy(), i)
    assert_bits_equal(bank.dstate.numpy(), d)
