"""GPU (-m gpu): the TIME-PARALLEL tolerance mode for small banks of linear filters (knob "time_parallel", csrc/scan.hip): a
wavefront takes one voice, its 64 lanes take consecutive time segments, the segment states are joined by a Kogge-Stone scan over
wavefront shuffles (north_star: "wavefront shuffles for the biquad recurrence").  The arithmetic is reordered, so the mode is not
bit-exact: stated tolerance |error| <= SCAN_RTOL (1e-10) x the voice's peak over the carried blocks, against the oracle's sequential
recurrence.  With the knob off (the default) the same calls are bit-exact -- asserted here too, so the knob cannot leak."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

SCAN_RTOL = 1e-10  # x per-voice peak of |reference output| (measured on MI355X: <= 5e-12, the worst a 2048-sample block of high-Q biquads)


def _scaled_err(got, exp):
    peak = np.maximum(np.abs(exp).max(axis=0), 1e-300)
    return float((np.abs(got - exp) / peak).max())


@pytest.fixture
def scan_on(mx):
    prev = mx.lib().mxg_tune(b"time_parallel", 1)
    yield
    mx.lib().mxg_tune(b"time_parallel", prev)


@pytest.mark.parametrize("V,N", [(1, 64), (6, 512), (70, 2048), (6, 128), (3, 1024)])
@pytest.mark.parametrize("kind", ["dc", "svf", "biquad"])
def test_filter2_time_parallel_within_tolerance(mx, port, scan_on, kind, V, N):
    rng = np.random.default_rng(V * 1000 + N)
    blocks = 3
    x = rng.uniform(-1, 1, (blocks * N, V))
    v = np.arange(V)
    if kind == "dc":
        R = 0.99 + 0.009 * rng.uniform(0, 1, V)
        bank = mx.maxiDCBlockerBank(V)
        run = lambda xb: bank.play(mx.DeviceBuffer.from_numpy(xb), R).numpy()
        exp, est, _ = port.filter2(0, x, R[None, :])
    elif kind == "svf":
        cut, q = 80.0 + 900.0 * rng.uniform(0, 1, V), 0.5 + 4.0 * rng.uniform(0, 1, V)
        bank = mx.maxiSVFBank(V)
        bank.setCutoff(cut); bank.setResonance(q)
        run = lambda xb: bank.play(mx.DeviceBuffer.from_numpy(xb), 0.5, 0.25, 0.125, 0.6).numpy()
        exp, est, _ = port.filter2(1, x, np.stack([cut, q, np.full(V, 0.5), np.full(V, 0.25), np.full(V, 0.125), np.full(V, 0.6)]))
    else:
        typ = (v % 7).astype(np.float64)
        cut, q, gain = 100.0 + 3000.0 * rng.uniform(0, 1, V), 0.4 + 3.0 * rng.uniform(0, 1, V), -9.0 + 18.0 * rng.uniform(0, 1, V)
        bank = mx.maxiBiquadBank(V)
        bank.set(typ.astype(np.int32), cut, q, gain)
        run = lambda xb: bank.play(mx.DeviceBuffer.from_numpy(xb)).numpy()
        exp, est, _ = port.filter2(2, x, np.stack([typ, cut, q, gain]))
    got = np.concatenate([run(x[b * N:(b + 1) * N]) for b in range(blocks)])   # state carried from block to block
    err = _scaled_err(got, exp)
    print("%s V=%d N=%d: max |err| / peak = %.3e (tolerance %.0e)" % (kind, V, N, err, SCAN_RTOL))
    assert err <= SCAN_RTOL
    assert np.abs(bank.state.numpy()[:est.shape[0]] - est).max() <= SCAN_RTOL * max(1.0, np.abs(est).max())


@pytest.mark.parametrize("kind", ["lores", "hires"])
@pytest.mark.parametrize("V,N", [(6, 512), (33, 256)])
def test_lores_time_parallel_within_tolerance(mx, port, scan_on, kind, V, N):
    rng = np.random.default_rng(V + N)
    x = rng.uniform(-1, 1, (3 * N, V))
    cut, res = 200.0 + 5000.0 * rng.uniform(0, 1, V), 1.0 + 12.0 * rng.uniform(0, 1, V)
    bank = mx.maxiFilterBank(V)
    got = np.concatenate([bank.render(kind, mx.DeviceBuffer.from_numpy(x[b * N:(b + 1) * N]), cut, res).numpy() for b in range(3)])
    exp, est = port.filter(0 if kind == "lores" else 1, x, cut, res)
    err = _scaled_err(got, exp)
    print("%s V=%d N=%d: max |err| / peak = %.3e" % (kind, V, N, err))
    assert err <= SCAN_RTOL


def test_knob_off_is_bit_exact_and_odd_shapes_fall_back(mx, port):
    """Default knob: the exact kernels.  With the knob on, shapes the scan does not take (N not a multiple of 64) still go through the
    exact kernels and stay bit-exact."""
    rng = np.random.default_rng(7)
    V, N = 6, 512
    x = rng.uniform(-1, 1, (N, V))
    typ, cut, q, gain = np.zeros(V), np.full(V, 1200.0), np.full(V, 0.7), np.zeros(V)
    exp, _, _ = port.filter2(2, x, np.stack([typ, cut, q, gain]))
    bank = mx.maxiBiquadBank(V)
    bank.set(typ.astype(np.int32), cut, q, gain)
    assert_bits_equal(bank.play(mx.DeviceBuffer.from_numpy(x)).numpy(), exp, "knob off")
    prev = mx.lib().mxg_tune(b"time_parallel", 1)
    try:
        bank = mx.maxiBiquadBank(V)
        bank.set(typ.astype(np.int32), cut, q, gain)
        assert_bits_equal(bank.play(mx.DeviceBuffer.from_numpy(x[:500])).numpy(), exp[:500], "N = 500 falls back to the exact kernel")
    finally:
        mx.lib().mxg_tune(b"time_parallel", prev)
