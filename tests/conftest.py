import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build the checker (plain-C oracle) and, if missing, the HIP library.  Building the
    checker is not using it; the product never loads it."""
    from oracle import pyoracle
    if not os.path.exists(pyoracle.PORT_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    import maximilian_amd
    if not os.path.exists(maximilian_amd.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "maximilian_amd", "csrc"), "-j4"])


@pytest.fixture(scope="session")
def port():
    _ensure_built()
    from oracle import pyoracle
    o = pyoracle.port()
    o.settings(44100, 2, 1024)
    return o


@pytest.fixture(scope="session")
def ref():
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref/libmaxiref.so not present (built only where /root/reference exists)")
    o = pyoracle.reference()
    o.settings(44100, 2, 1024)
    return o


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def mx():
    """The product package on a GPU box (HIP library loaded, device initialised)."""
    _ensure_built()
    import maximilian_amd as m
    m._lib.check(m.lib().mxg_init(-1), "mxg_init")
    m.maxiSettings.setup(44100, 2, 1024)
    return m


# Optimisation / instrumentation flags of the host builds of the device headers (tests/host_*.cpp).  MXG_HOST_UBSAN=1 swaps -O2
# for UndefinedBehaviorSanitizer (+ float-cast-overflow), aborting on the first report: tests/test_host_ubsan.py re-runs those
# suites that way, so the arithmetic the kernels share with their host harnesses is checked for signed overflow, out-of-range
# float -> int casts, misaligned or out-of-bounds accesses the compiler can see, ...
HOST_OPT = (["-O1", "-fsanitize=undefined,float-cast-overflow", "-fno-sanitize-recover=all"]
            if os.environ.get("MXG_HOST_UBSAN") else ["-O2"])


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def mix_tol(units, peak=1.0, sums=None):
    """Absolute tolerance of a maxiMix mixdown over `units` voices/streams.  The only non-bit-exact step of the path:
    the products x*gain are the reference's bits, the sum is tree-ordered on the device and sequential in the
    reference (src/maximilian.cpp:404-410 applied unit after unit), so the two differ by the rounding of partial sums.  Each of the
    `units` additions rounds by at most eps/2 of the running sum S, the roundings add up like a random walk: ~ eps * sqrt(units) * S.
    For voices with unrelated phases S ~ sqrt(units) * peak, which gives the linear form eps * units * peak -- measured <= 2.2e-17 *
    units * peak on the GPU (r02: 65536 voices, 2048 streams); 1e-15 leaves ~50x.  Where the voices are COHERENT (a bank started from
    phase 0: the first samples of all its voices have the same sign, S ~ units * peak / 2) the reference's own sequential sum is that
    much less accurate; pass the expected sums (`sums`) and the bound scales with their magnitude:
    1e-15 * sqrt(units) * max(|sums|, sqrt(units) * peak)."""
    u = float(units)
    scale = max(1.0, float(peak)) * np.sqrt(u)
    if sums is not None:
        scale = max(scale, float(np.abs(sums).max()))
    return 1e-15 * np.sqrt(u) * scale


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # NaN == NaN whatever its sign/payload: IEEE 754 leaves the sign of an arithmetic NaN
    # unspecified (x86 SSE produces 0xFFF8..., gfx950 propagates/negates operand NaNs); the
    # reference's behaviour at such a sample is "NaN" (e.g. lores with cutoff clamped to sr,
    # where r = 0/0, src/maximilian.cpp:461).
    neq = (a.view(np.uint64) != b.view(np.uint64)) & ~(np.isnan(a) & np.isnan(b))
    if neq.any():
        idx = np.argwhere(neq)[0]
        raise AssertionError("%s: %d of %d values differ bitwise; first at %s: %r vs %r" % (
            what, int(neq.sum()), a.size, tuple(idx), a[tuple(idx)], b[tuple(idx)]))


def ulp_diff(a, b):
    """Distance in units of the last place between two float64 arrays (same sign regime)."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return np.abs(ia - ib)


def assert_close_scaled(o, e, rtol, what=""):
    """|o-e| <= rtol * (per-voice peak of |e|); NaN positions must coincide (see assert_bits_equal)."""
    o = np.asarray(o, np.float64)
    e = np.asarray(e, np.float64)
    nan_e, nan_o = np.isnan(e), np.isnan(o)
    assert np.array_equal(nan_e, nan_o), "%s: NaN positions differ" % what
    scale = np.nanmax(np.abs(np.where(nan_e, 0.0, e)), axis=0, keepdims=True)
    err = np.where(nan_e, 0.0, np.abs(o - e))
    bad = err > rtol * np.maximum(scale, 1e-300)
    assert not bad.any(), "%s: max scaled error %.3e > %.1e" % (
        what, float((err / np.maximum(scale, 1e-300)).max()), rtol)
    return float((err / np.maximum(scale, 1e-300)).max())
