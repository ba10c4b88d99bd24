"""CPU tests (-m "not gpu"): FFT/MFCC oracle vs golden + reference, host tables, stream framing."""
import numpy as np
import pytest

from conftest import assert_bits_equal

CASES = [(1024, 1024, 1024), (1024, 256, 0), (512, 128, 512), (2048, 1024, 2048), (64, 64, 64)]


def sig_slice(sig, fs, hop):
    return sig[:fs * 4] if fs > 1024 else sig[:sig.size // (2 if hop < 1024 else 1)]


def f32bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("fs,hop,win", CASES)
def test_fft_golden(port, golden, fs, hop, win):
    g = golden("spectral.npz")
    r = port.fft_stream(sig_slice(g["signal"], fs, hop), fs, hop, win)
    for key in ("real", "imag", "mags", "phases"):
        e = g["%s_%d_%d" % (key, fs, hop)]
        assert r[key].shape == e.shape
        assert np.array_equal(f32bits(r[key]), f32bits(e)), key


def test_mfcc_golden(port, golden):
    g = golden("spectral.npz")
    mags = g["mags_1024_1024"]
    assert np.array_equal(f32bits(port.fft_to_db(mags[0])), f32bits(g["db_1024_1024"]))
    for nf, nc in [(42, 13), (256, 13), (40, 20)]:
        mel, mf = port.mfcc(mags, nf, nc, 20.0, 20000.0)
        assert_bits_equal(mel, g["melbands_%d_%d" % (nf, nc)])
        assert_bits_equal(mf, g["mfcc_%d_%d" % (nf, nc)])


def test_fft_mfcc_port_vs_reference(port, ref):
    rng = np.random.default_rng(17)
    sig = rng.uniform(-1, 1, 5000).astype(np.float32)
    for fs, hop, win in [(1024, 512, 1024), (256, 64, 0), (8, 4, 8), (4096, 4096, 4096)]:
        a, b = port.fft_stream(sig, fs, hop, win), ref.fft_stream(sig, fs, hop, win)
        for key in ("real", "imag", "mags", "phases"):
            assert a[key].shape == b[key].shape and a[key].shape[0] == sig.size // hop
            assert np.array_equal(f32bits(a[key]), f32bits(b[key])), (fs, key)
    mags = port.fft_stream(sig, 1024, 512, 1024)["mags"]
    for cfg in [(42, 13, 20.0, 20000.0), (256, 13, 20.0, 20000.0), (30, 12, 300.0, 8000.0)]:
        x, y = port.mfcc(mags, *cfg), ref.mfcc(mags, *cfg)
        assert_bits_equal(x[0], y[0]); assert_bits_equal(x[1], y[1])
        assert_bits_equal(port.mfcc_tables(512, *cfg)[0], ref.mfcc_tables(512, *cfg)[0])
        assert_bits_equal(port.mfcc_tables(512, *cfg)[1], ref.mfcc_tables(512, *cfg)[1])


def test_product_mfcc_host_tables(port):
    """mxg_mfcc_plan_create builds its tables on the host libm: bit-identical to the oracle's."""
    import maximilian_amd as mx
    mx.lib().mxg_settings(44100, 2, 1024)
    for cfg in [(512, 42, 13, 20.0, 20000.0), (512, 256, 13, 20.0, 20000.0), (256, 40, 20, 100.0, 30000.0)]:
        m = mx.maxiMFCC()
        m.setup(*cfg)
        W, D, used = m.tables()
        We, De = port.mfcc_tables(*cfg)
        assert_bits_equal(W, We); assert_bits_equal(D, De)
        nz = np.nonzero(We.reshape(cfg[0], cfg[1]).any(axis=1))[0]
        assert used == nz.max() + 1
        # the exact kernel's premise: every filter's support is one contiguous run of bins
        Wm = We.reshape(cfg[0], cfg[1])
        for f in range(cfg[1]):
            idx = np.nonzero(Wm[:, f])[0]
            assert idx.size == 0 or (np.diff(idx) == 1).all()
        assert (Wm >= 0).all() and (Wm[:, 0] == 0).all()
    with pytest.raises(ValueError):
        mx.maxiMFCC().setup(512, 1, 13, 20.0, 20000.0)


def test_stream_framing_matches_oracle(port):
    """frames_in_stream / padded_stream reproduce maxiFFT::process()'s hop buffer (maxiFFT.cpp:56-87)."""
    import maximilian_amd as mx
    rng = np.random.default_rng(2)
    for n, fs, hop in [(5000, 1024, 256), (1023, 1024, 1024), (1024, 1024, 1024), (300, 64, 16), (10, 64, 16)]:
        sig = rng.uniform(-1, 1, n).astype(np.float32)
        r = port.fft_stream(sig, fs, hop, fs)
        assert mx.frames_in_stream(n, hop, fs) == r["mags"].shape[0]
        p = mx.padded_stream(sig, hop, fs)
        nfr = r["mags"].shape[0]
        assert nfr == 0 or (nfr - 1) * hop + fs <= p.size
        # frame k of the padded stream, transformed alone (hop = fs), equals frame k of the stream
        for k in ([0, nfr - 1] if nfr else []):
            one = port.fft_stream(p[k * hop:k * hop + fs], fs, fs, fs)
            assert np.array_equal(f32bits(one["mags"][0]), f32bits(r["mags"][k]))


IFFT_TAGS = [(1024, 512, 0), (1024, 256, 1024), (64, 16, 48)]


@pytest.mark.parametrize("fs,hop,win", IFFT_TAGS)
def test_ifft_golden(port, golden, fs, hop, win):
    """maxiIFFT (L/maxiFFT.cpp:140-192): the restatement against the values dumped from the compiled reference,
    incl. the overlap-add buffer carried across two calls."""
    g = golden("ifft.npz")
    tag = "%d_%d_%d" % (fs, hop, win)
    m, ph = g["mags_" + tag], g["phases_" + tag]
    o1, io1, buf = port.ifft_stream(m[:3], ph[:3], fs, hop, win)
    o2, io2, buf = port.ifft_stream(m[3:], ph[3:], fs, hop, win, buffer=buf)
    for got, name in ((np.concatenate([o1, o2]), "signal_"), (np.concatenate([io1, io2]), "ifftout_"), (buf, "buffer_")):
        assert np.array_equal(got.view(np.uint32), g[name + tag].view(np.uint32)), name


def test_ifft_port_vs_reference(port, ref):
    rng = np.random.default_rng(12)
    for (fs, hop, win) in [(512, 128, 0), (2048, 2048, 0), (8, 2, 6), (256, 64, 100)]:
        m = np.abs(rng.normal(0, 2, (6, fs // 2))).astype(np.float32)
        ph = rng.uniform(-4, 4, (6, fs // 2)).astype(np.float32)
        for u, w in zip(port.ifft_stream(m, ph, fs, hop, win), ref.ifft_stream(m, ph, fs, hop, win)):
            assert np.array_equal(u.view(np.uint32), w.view(np.uint32))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_convolve_port_matches_golden(port, golden, tag):
    """maxiConvolve: the plain-C restatement against the fixtures dumped from the compiled reference, both modes
    (0 = the reference verbatim: silence; 1 = sums routed to the inverse transform's inputs)."""
    g = golden("convolve.npz")
    F, H = (int(v) for v in g["cfg_" + tag])
    for mode in (0, 1):
        o, r, i = port.convolve(g["pcm_" + tag], g["x_" + tag], F, H, mode)
        assert np.array_equal(o.view(np.uint32), g["out%d_%s" % (mode, tag)].view(np.uint32))
    assert np.array_equal(r.view(np.uint32), g["impR_" + tag].view(np.uint32))
    assert np.array_equal(i.view(np.uint32), g["impI_" + tag].view(np.uint32))
    assert not g["out0_" + tag].any() and np.abs(g["out1_" + tag]).max() > 0.01


def test_convolve_port_matches_reference(port, ref):
    rng = np.random.default_rng(99)
    for Li, F, H in [(700, 64, 16), (6000, 512, 128), (2049, 1024, 256), (100, 1024, 256)]:
        pcm = (rng.uniform(-1, 1, Li) * 20000).astype(np.int16)
        x = rng.uniform(-1, 1, F * 5 + 7).astype(np.float32)
        for mode in (0, 1):
            a, b = ref.convolve(pcm, x, F, H, mode), port.convolve(pcm, x, F, H, mode)
            for which, (u, v) in enumerate(zip(a, b)):
                same = np.array_equal(u.view(np.uint32), v.view(np.uint32))
                assert same, ("impulse %d, fft %d, hop %d, mode %d, array %d (0 out, 1 impR, 2 impI): %d of %d differ, "
                              "max |ref| %.3g, max |port| %.3g" % (Li, F, H, mode, which, int((u != v).sum()), u.size,
                                                                  np.abs(u).max(), np.abs(v).max()))


def _mfma_4x4x4_4b(a, b, c):
    """v_mfma_f64_4x4x4_4b_f64 on 64-lane operand vectors, as measured on gfx950 (tools/ubench/mfma_probe.hip): four independent
    4 x 4 x 4 blocks; A lane 16 k + 4 blk + i = A_blk[i][k], B lane 16 k + 4 blk + j = B_blk[k][j], D lane 16 i + 4 blk + j."""
    d = np.array(c, np.float64).copy()
    for blk in range(4):
        for i in range(4):
            for j in range(4):
                acc = d[16 * i + 4 * blk + j]
                for k in range(4):
                    acc += a[16 * k + 4 * blk + i] * b[16 * k + 4 * blk + j]
                d[16 * i + 4 * blk + j] = acc
    return d


@pytest.mark.parametrize("cfg", [(42, 13, 20.0, 20000.0), (48, 16, 20.0, 20000.0), (13, 5, 20.0, 20000.0), (26, 13, 300.0, 8000.0),
                                 (30, 12, 50.0, 22050.0)])
def test_matrix_pipe_tables_replay(port, cfg):
    """The quad tables of knob fused_mel (mxg_mfcc_plan_matrix_tables): replaying the fused kernel's lane program on the host --
    the same lane -> (quarter k, quad gs, frame half fh, row i / column j) maps, the same batches -- reproduces the oracle's band
    sums and mfcc within the stated tolerances.  What this pins without a GPU: the tables cover every non-zero weight exactly
    once, every read stays inside a 260-float magnitude row, and the D layout of the mel product IS the B layout of the DCT."""
    import ctypes
    import maximilian_amd as mx
    nf, nc, lo_f, hi_f = cfg
    mx.lib().mxg_settings(44100, 2, 1024)
    port.settings(44100, 2, 1024)
    m = mx.maxiMFCC()
    m.setup(512, nf, nc, lo_f, hi_f)
    nb = (ctypes.c_int * 6)()
    base = (ctypes.c_int * 12)()
    W = np.zeros(64 * 128)
    D = np.zeros(4 * 6 * 32)
    batches = mx.lib().mxg_mfcc_plan_matrix_tables(m.plan, nb, base, W.ctypes.data, W.size, D.ctypes.data)
    assert batches == sum(nb) and batches > 0
    rng = np.random.default_rng(nf)
    mags = (np.abs(rng.normal(size=(8, 512))) * 10.0 ** rng.uniform(-3, 1.5, (8, 1))).astype(np.float32)
    M = np.zeros((8, 260), np.float32)
    M[:, 1:257] = mags[:, 1:257]          # the tile the post-pass leaves: bins 1 .. 256, bin 0 kept at zero
    lanes = np.arange(64)
    k, gs, fh, ij = lanes >> 4, (lanes >> 3) & 1, (lanes >> 2) & 1, lanes & 3
    lane32 = 8 * k + 4 * gs + ij
    acc = np.zeros((6, 64))
    t0 = 0
    for p in range(6):
        a = np.zeros(64)
        for s4 in range(nb[p]):
            for u in range(4):
                h, e = u >> 1, u & 1
                w = W[(((t0 + s4) * 2 + h) * 32 + lane32) * 2 + e]
                col = np.array([base[2 * p + g] for g in gs]) + k * 4 * nb[p] + 4 * s4 + u
                assert col.min() >= 0 and col.max() < 260
                x = M[4 * fh + ij, col].astype(np.float64)
                a = _mfma_4x4x4_4b(w, x, a)
        acc[p] = a
        t0 += nb[p]
    emel_raw = np.zeros((8, nf))
    Wd, Dd = port.mfcc_tables(512, nf, nc, lo_f, hi_f)
    Wd = np.asarray(Wd).reshape(512, nf)
    for f in range(nf):
        emel_raw[:, f] = mags.astype(np.float64) @ Wd[:, f]
    got_raw = np.zeros((8, 48))
    for p in range(6):
        got_raw[4 * fh + ij, 4 * (2 * p + gs) + k] = acc[p]
    assert np.abs(got_raw[:, :nf] - emel_raw).max() <= 1e-13 * np.abs(emel_raw).max()
    assert (got_raw[:, nf:] == 0).all()
    lg = np.where(acc > 0.000001, np.log(np.maximum(acc, 1e-6) ** 2), 0.0)
    out = np.zeros((8, 16))
    for q in range(4):
        c = np.zeros(64)
        for p in range(6):
            c = _mfma_4x4x4_4b(D[(q * 6 + p) * 32 + lane32], lg[p], c)
        c = c + c[lanes ^ 8]                                    # row_ror:8: the other quad of the pair
        sel = gs == 0
        out[(4 * fh + ij)[sel], (4 * q + k)[sel]] = c[sel] / nc
    emel, emf = port.mfcc(mags, nf, nc, lo_f, hi_f)
    assert np.abs(out[:, :nc] - emf).max() <= 1e-11 * max(1.0, np.abs(emel).max())
    assert (out[:, nc:] == 0).all()
