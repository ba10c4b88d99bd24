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


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    neq = a.view(np.uint64) != b.view(np.uint64)
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
