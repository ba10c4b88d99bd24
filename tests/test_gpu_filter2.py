"""GPU parity (-m gpu): maxiDCBlocker / maxiSVF / maxiBiquad banks (H:1255-1486) through the C-ABI vs the
oracle.  Coefficients are host-libm (must equal the reference's members bit for bit), recurrences bit-exact."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def _x(V, N, seed):
    return np.random.default_rng(seed).uniform(-1, 1, (N, V))


@pytest.mark.parametrize("N", [1, 7, 64, 515])
def test_dcblocker(mx, port, N):
    V = 777
    rng = np.random.default_rng(N)
    x = _x(V, 2 * N, 1) + 0.3
    R = rng.uniform(0.9, 0.9999, V)
    bank = mx.maxiDCBlockerBank(V)
    o = np.concatenate([bank.play(mx.DeviceBuffer.from_numpy(x[:N]), R).numpy(),
                        bank.play(mx.DeviceBuffer.from_numpy(x[N:]), R).numpy()])
    e, st, _ = port.filter2(0, x, R[None, :])
    assert_bits_equal(o, e, "dcblocker")
    assert_bits_equal(bank.state.numpy()[:2], st[:2], "xm1, ym1")


@pytest.mark.parametrize("N", [5, 300])
def test_svf(mx, port, N):
    V = 1000
    rng = np.random.default_rng(10 + N)
    x = _x(V, 2 * N, 2)
    cutoff, res = rng.uniform(20, 20000, V), rng.uniform(0, 12, V)
    res[:5] = 0.0                                         # damping = 0 branch (H:1325)
    mix = rng.uniform(0, 1, (4, V))
    bank = mx.maxiSVFBank(V)
    e0, _, c0 = port.filter2(1, x[:1], np.stack([np.full(V, 1000.0), np.ones(V), *mix]))
    assert_bits_equal(bank.coefficients(), c0, "ctor setParams(1000, 1)")
    bank.setCutoff(cutoff)
    bank.setResonance(res)
    o = np.concatenate([bank.play(mx.DeviceBuffer.from_numpy(x[:N]), *mix).numpy(),
                        bank.play(mx.DeviceBuffer.from_numpy(x[N:]), *mix).numpy()])
    e, st, coef = port.filter2(1, x, np.stack([cutoff, res, *mix]))
    assert_bits_equal(bank.coefficients(), coef, "g1..g4, k")
    assert_bits_equal(o, e, "svf")
    assert_bits_equal(bank.state.numpy(), st, "v0z, v1, v2")


@pytest.mark.parametrize("ftype", range(7))
def test_biquad(mx, port, ftype):
    V, N = 900, 257
    rng = np.random.default_rng(20 + ftype)
    x = _x(V, 2 * N, 3)
    cutoff, Q, gain = rng.uniform(30, 18000, V), rng.uniform(0.3, 8, V), rng.uniform(-18, 18, V)
    gain[:3] = [0.0, -0.0, 6.0]
    bank = mx.maxiBiquadBank(V)
    bank.set(ftype, cutoff, Q, gain)
    o = np.concatenate([bank.play(mx.DeviceBuffer.from_numpy(x[:N])).numpy(),
                        bank.play(mx.DeviceBuffer.from_numpy(x[N:])).numpy()])
    e, st, coef = port.filter2(2, x, np.stack([np.full(V, float(ftype)), cutoff, Q, gain]))
    assert_bits_equal(bank.host_coef, coef, "a0, a1, a2, b1, b2")
    assert_bits_equal(o, e, "biquad")
    assert_bits_equal(bank.state.numpy(), st, "v[0..2]")


def test_biquad_mixed_types_and_errors(mx, port):
    V, N = 70, 100
    rng = np.random.default_rng(33)
    x = _x(V, N, 4)
    t = (np.arange(V) % 7).astype(np.int32)
    cutoff, Q, gain = rng.uniform(30, 18000, V), rng.uniform(0.3, 8, V), rng.uniform(-18, 18, V)
    bank = mx.maxiBiquadBank(V)
    bank.set(t, cutoff, Q, gain)
    e, _, _ = port.filter2(2, x, np.stack([t.astype(np.float64), cutoff, Q, gain]))
    assert_bits_equal(bank.play(mx.DeviceBuffer.from_numpy(x)).numpy(), e)
    with pytest.raises(mx.MaxiGpuError):
        bank.set(9, cutoff, Q, gain)
    assert mx.lib().mxg_filter2_render(3, V, N, 1, 1, 1, 1, None) == -1


@pytest.mark.parametrize("rw", [1, 2, 3, 4])
@pytest.mark.parametrize("V,N", [(700, 300), (64, 8), (2050, 514), (4096, 1000), (701, 64), (256, 7)])
def test_pair_row_streams_same_bits(mx, port, rw, V, N):
    """The read + write kernels' 16-byte pair-row streams (knob rw_store; csrc/filter2.hip filter2_pairs_kernel): a lane pair swaps one
    value on the way in and one on the way out, nothing else changes -- every kind, partial last wavefronts, blocks that are not a
    multiple of the 8-sample chunk; odd banks / odd blocks fall back to the 8-byte kernels.  All against the oracle, state carried."""
    L = mx.lib()
    rng = np.random.default_rng(V + N)
    x = _x(V, 2 * N, V)
    prev = L.mxg_tune(b"rw_store", rw)
    try:
        R = rng.uniform(0.9, 0.9999, V)
        b = mx.maxiDCBlockerBank(V)
        o = np.concatenate([b.play(mx.DeviceBuffer.from_numpy(x[:N]), R).numpy(), b.play(mx.DeviceBuffer.from_numpy(x[N:]), R).numpy()])
        e, st, _ = port.filter2(0, x, R[None, :])
        assert_bits_equal(o, e, "dcblocker rw_store=%d" % rw)
        assert_bits_equal(b.state.numpy()[:2], st[:2])
        cutoff, res = rng.uniform(20, 20000, V), rng.uniform(0.3, 12, V)
        b = mx.maxiSVFBank(V); b.setCutoff(cutoff); b.setResonance(res)
        o = np.concatenate([b.play(mx.DeviceBuffer.from_numpy(x[:N]), 0.5, 0.25, 0.125, 0.6).numpy(),
                            b.play(mx.DeviceBuffer.from_numpy(x[N:]), 0.5, 0.25, 0.125, 0.6).numpy()])
        e, st, _ = port.filter2(1, x, np.stack([cutoff, res, np.full(V, 0.5), np.full(V, 0.25), np.full(V, 0.125), np.full(V, 0.6)]))
        assert_bits_equal(o, e, "svf rw_store=%d" % rw)
        assert_bits_equal(b.state.numpy(), st)
        typ = (np.arange(V) % 7).astype(np.float64)
        cut, q, gain = rng.uniform(100, 8000, V), rng.uniform(0.4, 3, V), rng.uniform(-9, 9, V)
        b = mx.maxiBiquadBank(V); b.set(typ.astype(np.int32), cut, q, gain)
        o = np.concatenate([b.play(mx.DeviceBuffer.from_numpy(x[:N])).numpy(), b.play(mx.DeviceBuffer.from_numpy(x[N:])).numpy()])
        e, st, _ = port.filter2(2, x, np.stack([typ, cut, q, gain]))
        assert_bits_equal(o, e, "biquad rw_store=%d" % rw)
    finally:
        L.mxg_tune(b"rw_store", prev)
