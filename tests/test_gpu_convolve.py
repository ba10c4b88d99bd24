"""GPU (-m gpu): maxiConvolve (src/libs/maxiConvolve.cpp) and maxiIFFT's COMPLEX mode through the C-ABI, bit-exact against
the oracle -- golden fixtures dumped from the compiled reference (tests/golden/convolve.npz) and the plain-C port on fresh
inputs.  Mode 0 is the reference verbatim (its COMPLEX-mode inverse transform never sees the sums: silence); mode 1 routes
the sums to the transform's inputs, checked against the reference's own calcIFFT fed that way."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def f32bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_convolve_golden(mx, golden, tag):
    g = golden("convolve.npz")
    F, H = (int(v) for v in g["cfg_" + tag])
    pcm, x = g["pcm_" + tag], g["x_" + tag]
    c = mx.maxiConvolve()
    c.setup(pcm / 32767.0, F, H)
    assert c.frames == g["impR_" + tag].shape[0]
    r, i = c.impulse()
    assert np.array_equal(f32bits(r), f32bits(g["impR_" + tag])) and np.array_equal(f32bits(i), f32bits(g["impI_" + tag]))
    for mode in (0, 1):
        c.reset()
        # in two calls (4 + 5 blocks): the delay line, the pending sums and the output buffer carry over
        o = np.concatenate([c.play(x[:4 * F], mode).numpy(), c.play(x[4 * F:], mode).numpy()])
        assert np.array_equal(f32bits(o), f32bits(g["out%d_%s" % (mode, tag)])), "mode %d" % mode
        if mode == 0:
            assert not o.any()
        else:
            assert np.abs(o).max() > 0.01


@pytest.mark.parametrize("Li,F,H,nb", [(700, 64, 16, 40), (3000, 512, 128, 3), (900, 1024, 256, 2), (40000, 1024, 512, 6),
                                       (4096, 2048, 256, 5)])
def test_convolve_vs_port(mx, port, Li, F, H, nb):
    """Fresh inputs against the plain-C port: impulse shorter than a frame (no impulse frame at all), an impulse of 39
    frames, block-by-block calls."""
    rng = np.random.default_rng(Li + F)
    pcm = (rng.uniform(-1, 1, Li) * np.exp(-np.arange(Li) / (Li / 5.0)) * 25000).astype(np.int16)
    x = rng.uniform(-1, 1, F * nb).astype(np.float32)
    c = mx.maxiConvolve()
    c.setup(pcm / 32767.0, F, H)
    for mode in (1, 0):
        e, er, ei = port.convolve(pcm, x, F, H, mode)
        assert c.frames == er.shape[0]
        if c.frames:
            r, i = c.impulse()
            assert np.array_equal(f32bits(r), f32bits(er)) and np.array_equal(f32bits(i), f32bits(ei))
        c.reset()
        o = np.concatenate([c.play(x[k * F:(k + 1) * F], mode).numpy() for k in range(nb)])
        assert np.array_equal(f32bits(o), f32bits(e)), "mode %d, one block per call" % mode
        c.reset()
        o = c.play(x, mode).numpy()
        assert np.array_equal(f32bits(o), f32bits(e)), "mode %d, one call" % mode


def test_convolve_rejects_bad_arguments(mx):
    L = mx.lib()
    a = np.zeros(100)
    assert not L.mxg_convolve_create(a.ctypes.data, 100, 100.0, 1000, 256)      # not a power of two
    assert not L.mxg_convolve_create(a.ctypes.data, 100, 100.0, 1024, 2048)     # hop > fft
    assert not L.mxg_convolve_create(None, 100, 100.0, 1024, 256)
    assert not L.mxg_convolve_create(a.ctypes.data, 0, 0.0, 1024, 256)
    c = mx.maxiConvolve()
    c.setup(np.ones(3000), 1024, 256)
    b = mx.DeviceBuffer(1024, np.float32)
    assert L.mxg_convolve_play(c.h, b.ptr, 1, b.ptr, 2, None) == -1               # unknown mode
    assert L.mxg_convolve_play(c.h, b.ptr, 0, b.ptr, 0, None) == 0                # empty call
    assert L.mxg_convolve_play(c.h, None, 1, b.ptr, 0, None) == -1
