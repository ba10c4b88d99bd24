"""CPU tests (-m "not gpu"): maxiGrains oracle vs golden + reference; product window tables."""
import numpy as np
import pytest

from conftest import assert_bits_equal

CASES = {"ts_hann": (0, 0, 0.05, 4, True), "ts_hamming_norand": (0, 1, 0.03, 3, False),
         "st_hann": (1, 0, 0.05, 2, True), "st_gauss": (1, 8, 0.021, 5, True)}


@pytest.mark.parametrize("name", list(CASES))
def test_granular_golden(port, golden, name):
    g = golden("grains.npz")
    mode, w, gl, ov, use_rnd = CASES[name]
    T = int(g["T"])
    h = T // 2
    rnd = g["rnd"] if use_rnd else None
    o1, st, gst, rc = port.granular(mode, w, g["samples"], h, g["speed"], b=g["timestretch"], rnd=rnd,
                                    grainLength=gl, overlaps=ov, st=g["st0"])
    assert rc == 0
    o2, st, gst, rc = port.granular(mode, w, g["samples"], T - h, g["speed"], b=g["timestretch"], rnd=rnd,
                                    grainLength=gl, overlaps=ov, st=st, gst=gst)
    assert rc == 0
    assert_bits_equal(np.concatenate([o1, o2]), g["out_" + name], name)
    assert_bits_equal(st, g["st_" + name], name + " state")
    assert_bits_equal(gst, g["gst_" + name], name + " grains")


def test_windows_golden_and_product(port, golden):
    import maximilian_amd as mx
    g = golden("grains.npz")
    mx.lib().mxg_settings(44100, 2, 1024)
    for k in range(9):
        assert_bits_equal(port.grain_window(k, 2205), g["windows_2205"][k], "window %d" % k)
        p = mx.lib().mxg_grain_plan_create(k, 0.05, 44100)   # host table; works without a device
        assert p
        w = np.zeros(2205)
        assert mx.lib().mxg_grain_plan_window(p, w.ctypes.data) == 2205
        assert_bits_equal(w, g["windows_2205"][k], "product window %d" % k)
        mx.lib().mxg_grain_plan_destroy(p)
    assert not mx.lib().mxg_grain_plan_create(0, 0.6, 44100)   # >= 500 ms: beyond the reference's cache
    assert not mx.lib().mxg_grain_plan_create(9, 0.05, 44100)


def test_granular_port_vs_reference(port, ref):
    rng = np.random.default_rng(23)
    Ls = 20000
    smp = rng.uniform(-1, 1, Ls)
    S, T = 9, 4000
    a = rng.uniform(-2, 2, S)
    b = rng.uniform(0.2, 2.0, S)
    pm = rng.uniform(-0.2, 0.2, S)
    rnd = rng.integers(0, 10, (S, 64))
    st0 = np.zeros((4, S)); st0[0] = rng.uniform(0, Ls - 1, S)
    for mode in (0, 1):
        for w, gl, ov in [(0, 0.05, 4), (6, 0.013, 2), (4, 0.1, 6)]:
            A = port.granular(mode, w, smp, T, a, b=b, posMod=pm, rnd=rnd, grainLength=gl, overlaps=ov, st=st0)
            B = ref.granular(mode, w, smp, T, a, b=b, posMod=pm, rnd=rnd, grainLength=gl, overlaps=ov, st=st0)
            assert A[3] == 0 and B[3] == 0
            for i in range(3):
                assert_bits_equal(A[i], B[i], "mode %d w %d item %d" % (mode, w, i))
    # too many overlapping grains for 8 slots -> both report it
    assert port.granular(0, 0, smp, 8000, a, rnd=None, grainLength=0.2, overlaps=12, st=st0)[3] == -3
    assert ref.granular(0, 0, smp, 8000, a, rnd=None, grainLength=0.2, overlaps=12, st=st0)[3] == -3
