"""GPU parity (-m gpu) at the lengths SURVEY.md 8(d) specifies, every sample of every unit compared with the CPU oracle:

  config 2   65 536-voice sinebuf / sinebuf4 bank, 64 consecutive blocks of 512 with state carried (32 768 samples per
             voice, 2.1 G samples per waveform), bit-exact
  config 3   65 536 subtractive voices, 128 consecutive blocks (65 536 samples per voice: two attacks/decays, sustain,
             release), gate(n) = (n mod 44100) < 22050; mode A (hoisted coefficients) bit-exact, mode B (14.monosynth
             order, per-sample device cos/sqrt coefficients) within the stated tolerance, measured maximum printed
  config 4   the dense MFMA mel contraction over all 1 048 576 frames against the exact sparse path
  config 5   all 2048 grain streams of one GPU's share, bit-exact (outputs, scheduler state, live grains)

The oracle (plain-C port) is single-threaded per call; units are independent, so the host shards them over a thread pool
(ctypes releases the GIL) -- the comparison is still one oracle value per device value."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import assert_bits_equal, ulp_diff

pytestmark = pytest.mark.gpu

V, B = 65536, 512
NTHREADS = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))


def shards(n, parts):
    edges = [n * i // parts for i in range(parts + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def count_mismatch(a, b):
    """Bitwise mismatches between two float64 arrays (NaN == NaN)."""
    ne = (a.view(np.uint64) != b.view(np.uint64)) & ~(np.isnan(a) & np.isnan(b))
    return int(ne.sum())


@pytest.mark.parametrize("wf,name", [(8, "sinebuf"), (9, "sinebuf4")])
def test_config2_64_blocks_every_sample(mx, port, wf, name):
    K = 64
    freq = 20.0 + np.arange(V) * 0.30517578125
    bank = mx.maxiOscBank(V)
    out = mx.DeviceBuffer((B, V), zero=False)
    sh = shards(V, NTHREADS)
    state = [(None, None)] * len(sh)
    bad = 0
    with ThreadPoolExecutor(len(sh)) as pool:
        for k in range(K):
            bank.render(wf, freq, B, out=out)
            got = out.numpy()

            def one(i):
                a, b = sh[i]
                e, ph, hd = port.osc(wf, freq[a:b], B, phase=state[i][0], hold=state[i][1])
                state[i] = (ph, hd)
                return count_mismatch(got[:, a:b], e)
            bad += sum(pool.map(one, range(len(sh))))
            assert bad == 0, "%s: %d samples differ by block %d" % (name, bad, k)
    eph = np.concatenate([s[0] for s in state])
    assert_bits_equal(bank.phase.numpy(), eph, "phase after 64 blocks")
    print("config 2 %s: %d blocks x %d samples x %d voices = %.2f G samples, all bit-identical to the oracle"
          % (name, K, B, V, K * B * V / 1e9))


@pytest.mark.parametrize("wf,name", [(0, "sinewave"), (1, "coswave")])
def test_config2_64_blocks_trig_every_sample(mx, port, wf, name):
    """maxiOsc::sinewave / coswave (C:228-235, 276-283) over the full config-2 length: the phase recurrence is bit-exact
    (it is carried from block to block, so a single wrong bit would show in every later block) and EVERY output sample is
    within 1 ULP of the oracle's glibc sin / cos of the same phase -- no absolute escape.  At 65 536 voices the launch takes
    the two-part time split (osc.hip: split = 2 below 2048 wavefronts), so this is also that path against the oracle."""
    K = 64
    freq = 20.0 + np.arange(V) * 0.30517578125
    bank = mx.maxiOscBank(V)
    out = mx.DeviceBuffer((B, V), zero=False)
    sh = shards(V, NTHREADS)
    state = [(None, None)] * len(sh)
    hist = np.zeros(3, np.int64)  # samples at 0 ULP, at 1 ULP, above
    with ThreadPoolExecutor(len(sh)) as pool:
        for k in range(K):
            bank.render(wf, freq, B, out=out)
            got = out.numpy()
            gph = bank.phase.numpy()

            def one(i):
                a, b = sh[i]
                e, ph, hd = port.osc(wf, freq[a:b], B, phase=state[i][0], hold=state[i][1])
                state[i] = (ph, hd)
                d = ulp_diff(got[:, a:b], e)
                return (int((d == 0).sum()), int((d == 1).sum()), int((d > 1).sum()),
                        count_mismatch(gph[a:b], ph))
            r = np.array(list(pool.map(one, range(len(sh)))))
            hist += r[:, :3].sum(axis=0)
            assert r[:, 3].sum() == 0, "%s: phase differs after block %d" % (name, k)
            assert hist[2] == 0, "%s: %d samples above 1 ULP by block %d" % (name, hist[2], k)
    total = K * B * V
    assert hist.sum() == total
    print("config 2 %s: %.2f G samples: %d identical to glibc, %d at 1 ULP (%.4f %%), 0 above; phase bit-identical in every block"
          % (name, total / 1e9, hist[0], hist[1], 100.0 * hist[1] / total))


@pytest.mark.parametrize("voices,blocks", [(131072, 4), (1048576, 4)])
@pytest.mark.parametrize("wf,name", [(8, "sinebuf"), (9, "sinebuf4"), (0, "sinewave")])
def test_large_banks_every_sample(mx, port, voices, blocks, wf, name):
    """The north star's >= 10^5-voice banks: 131 072 voices (a 537 MB block: the non-temporal store flavour, osc.hip picks
    it by block size) and 1 048 576 voices (4.3 GB block, plain stores, 16 384 wavefronts = 16 per SIMD), carried
    blocks, every sample against the oracle: wavetable forms bit-exact, sinewave <= 1 ULP with a bit-exact phase."""
    if voices > 200000 and wf != 8:
        pytest.skip("the 1 M-voice bank is checked for sinebuf (the store path is shared)")
    freq = 20.0 + (np.arange(voices) % 65536) * 0.30517578125 + (np.arange(voices) // 65536) * 0.001953125
    bank = mx.maxiOscBank(voices)
    out = mx.DeviceBuffer((B, voices), zero=False)
    sh = shards(voices, max(NTHREADS, 8))
    state = [(None, None)] * len(sh)
    ones = 0
    with ThreadPoolExecutor(min(len(sh), NTHREADS)) as pool:
        for k in range(blocks):
            bank.render(wf, freq, B, out=out)
            got = out.numpy()

            def one(i):
                a, b = sh[i]
                e, ph, hd = port.osc(wf, freq[a:b], B, phase=state[i][0], hold=state[i][1])
                state[i] = (ph, hd)
                if wf == 0:
                    d = ulp_diff(got[:, a:b], e)
                    return int((d > 1).sum()), int((d == 1).sum())
                return count_mismatch(got[:, a:b], e), 0
            r = np.array(list(pool.map(one, range(len(sh)))))
            assert r[:, 0].sum() == 0, "%s, %d voices: %d samples off in block %d" % (name, voices, r[:, 0].sum(), k)
            ones += int(r[:, 1].sum())
            del got
    assert_bits_equal(bank.phase.numpy(), np.concatenate([s[0] for s in state]), "phase after %d blocks" % blocks)
    print("%s, %d voices x %d blocks x %d samples = %.2f G samples: %s" % (
        name, voices, blocks, B, voices * blocks * B / 1e9,
        "all within 1 ULP (%d at 1 ULP), phase bit-identical" % ones if wf == 0 else "all bit-identical to the oracle"))


VOICE_B_RTOL = 1e-11  # mode B: device cos/sqrt coefficients through a recursive filter, scaled by the voice's peak


@pytest.mark.parametrize("mode", [0, 1])
def test_config3_128_blocks_every_sample(mx, port, mode):
    K = 128
    freq = np.minimum(20.0 + np.arange(V) * 0.30517578125, 5000.0)
    cutoff = (200 + 4 * freq) if mode == 0 else np.full(V, 10000.0)   # mode B: cutoff = adsr * 10000 (14.monosynth:53)
    res = 1.0 + (np.arange(V) % 16)
    vb = mx.maxiVoiceBank(V)
    vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
    gate = ((np.arange(K * B) % 44100) < 22050).astype(np.int32)
    out = mx.DeviceBuffer((B, V), zero=False)
    sh = shards(V, NTHREADS)
    st = [None] * len(sh)
    par, hold = vb.env.par, vb.env.holdtime
    bad, worst, peak = 0, np.zeros(V), np.zeros(V)
    with ThreadPoolExecutor(len(sh)) as pool:
        for k in range(K):
            trig = gate[k * B:(k + 1) * B]
            vb.render(mode, freq, cutoff, res, trig, B, out=out)
            got = out.numpy()

            def one(i):
                a, b = sh[i]
                s = st[i] or (None, None, None, None)
                e, ost, fst, dst, ist = port.voice(mode, freq[a:b], cutoff[a:b], res[a:b], trig, par[:, a:b], hold[a:b],
                                                   ost=s[0], fst=s[1], dstate=s[2], istate=s[3])
                st[i] = (ost, fst, dst, ist)
                if mode == 0:
                    return count_mismatch(got[:, a:b], e), None, None
                fin = np.isfinite(e)
                assert np.array_equal(fin, np.isfinite(got[:, a:b]))
                err = np.where(fin, np.abs(got[:, a:b] - e), 0.0).max(axis=0)
                return 0, err, np.where(fin, np.abs(e), 0.0).max(axis=0)
            for i, (nb, err, pk) in enumerate(pool.map(one, range(len(sh)))):
                bad += nb
                if err is not None:
                    a, b = sh[i]
                    worst[a:b] = np.maximum(worst[a:b], err)
                    peak[a:b] = np.maximum(peak[a:b], pk)
            assert bad == 0, "mode A: %d samples differ by block %d" % (bad, k)
    if mode == 0:
        dst = np.concatenate([s[2] for s in st], axis=1)
        ist = np.concatenate([s[3] for s in st], axis=1)
        assert_bits_equal(vb.env.dstate.numpy(), dst, "envelope amplitude/output after 128 blocks")
        assert np.array_equal(vb.env.istate.numpy(), ist)
        print("config 3 mode A: %d blocks x %d x %d = %.2f G samples, all bit-identical to the oracle" % (K, B, V, K * B * V / 1e9))
    else:
        scaled = worst / np.maximum(peak, 1e-300)
        print("config 3 mode B: max |err| / per-voice peak = %.3e over %.2f G samples (tolerance %.0e)"
              % (scaled.max(), K * B * V / 1e9, VOICE_B_RTOL))
        assert scaled.max() <= VOICE_B_RTOL


def test_config3_mixdown_full_bank(mx, port):
    """Config 3's N > 1 step at its full size (65 536 voices x 512, K2f with the maxiMix::stereo mixdown fused: mxg_voice_render_mix_rows),
    16 carried blocks through attack / decay / sustain / release: every block and every state array bit-identical to mxg_voice_render's
    (itself bit-identical to the oracle over 128 blocks: test_config3_128_blocks_every_sample), the rows' sum within mix_tol of the
    reference's voice-after-voice sum (C:503-509, 15.polysynth/main.cpp:54-70) over the same per-voice values."""
    from conftest import mix_tol
    K = 16
    freq = np.minimum(20.0 + np.arange(V) * 0.30517578125, 5000.0)
    cutoff, res = 200 + 4 * freq, 1.0 + (np.arange(V) % 16)
    pan = np.arange(V) / (V - 1.0)
    gate = ((np.arange(K * B) % 4410) < 2205).astype(np.int32)   # (ten times faster than config 3's gate: every stage within 16 blocks)
    banks = [mx.maxiVoiceBank(V), mx.maxiVoiceBank(V)]
    for vb in banks:
        vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
    L = mx.lib()
    G = L.mxg_osc_mix_groups(V)
    rows = mx.DeviceBuffer((G, B, 2), np.float64)
    mix = mx.DeviceBuffer((B, 2), np.float64)
    worst = 0.0
    for k in range(K):
        trig = gate[k * B:(k + 1) * B]
        a = banks[0].render(0, freq, cutoff, res, trig, B).numpy()
        o, _ = banks[1].render_mix(0, freq, cutoff, res, trig, pan, B, rows=rows)
        assert count_mismatch(o.numpy(), a) == 0, "block %d" % k
        mx._lib.check(L.mxg_mix_rows_sum(G, B * 2, rows.ptr, mix.ptr, None), "mxg_mix_rows_sum")
        em = port.mix_stereo(a, pan)
        gm = mix.numpy()
        # (config 3's high cutoffs x resonances let some voices run away to Inf / NaN within a few blocks -- in the reference too: a row
        # that holds one is NaN on both sides)
        fin = np.isfinite(em)
        assert np.array_equal(fin, np.isfinite(gm)), "block %d: non-finite mix rows differ" % k
        if not fin.any():
            continue
        af = np.where(np.isfinite(a), a, 0.0)
        err = np.abs(np.where(fin, gm - em, 0.0)).max()
        tol = mix_tol(V, np.abs(af).max(), sums=np.where(fin, em, 0.0))
        worst = max(worst, err / tol)
        assert err <= tol, "mix of block %d: %.3e > %.3e" % (k, err, tol)
    for x, y, name in ((banks[0].osc_state, banks[1].osc_state, "osc"), (banks[0].flt_state, banks[1].flt_state, "filter"),
                       (banks[0].env.dstate, banks[1].env.dstate, "env")):
        assert_bits_equal(y.numpy(), x.numpy(), name + " state after %d blocks" % K)
    assert np.array_equal(banks[0].env.istate.numpy(), banks[1].env.istate.numpy())
    print("config 3 mixdown form: %d blocks x %d x %d bit-identical to the plain render, mix within %.2f of mix_tol" % (K, B, V, worst))


def test_config4_mfma_all_frames(mx, port):
    """The dense fp64 MFMA mel contraction (method 1) over 1 048 576 frames: every frame against the exact sparse path
    on the device, a strided sample against the oracle."""
    import torch
    N = 1 << 20
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0x4D415849)
    # magnitudes with the dynamic range of real spectra: |N(0,1)| * 10^U(-3,1.5)
    mags = (torch.randn((N, 512), device=dev, generator=g).abs() *
            torch.pow(10.0, torch.rand((N, 1), device=dev, generator=g) * 4.5 - 3.0)).to(torch.float32)
    L = mx.lib()
    m = mx.maxiMFCC()
    m.setup(512, 42, 13, 20.0, 20000.0)
    exact = torch.empty((N, 13), dtype=torch.float64, device=dev)
    dense = torch.empty((N, 13), dtype=torch.float64, device=dev)
    rawx = torch.empty((N, 42), dtype=torch.float64, device=dev)
    rawd = torch.empty((N, 42), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    assert L.mxg_mfcc_batch(m.plan, mags.data_ptr(), 512, N, rawx.data_ptr(), None, exact.data_ptr(), 0, None) == 0
    assert L.mxg_mfcc_batch(m.plan, mags.data_ptr(), 512, N, rawd.data_ptr(), None, dense.data_ptr(), 1, None) == 0
    L.mxg_sync()
    # band sums: relative to the frame's largest band (fused / reordered fp64 sums of 512 non-negative terms)
    rel = ((rawd - rawx).abs().amax(dim=1) / rawx.abs().amax(dim=1).clamp_min(1e-300)).max().item()
    err = (dense - exact).abs().max().item()
    print("config 4 MFMA vs exact over %d frames: band sums rel %.3e, mfcc abs %.3e" % (N, rel, err))
    assert rel <= 1e-13
    assert err <= 1e-11
    sel = torch.arange(0, N, 9973, device=dev)
    emel, emf = port.mfcc(mags[sel].cpu().numpy(), 42, 13, 20.0, 20000.0)
    assert np.abs(dense[sel].cpu().numpy() - emf).max() <= 1e-11 * max(1.0, np.abs(emel).max())


MM_FORM_ATOL = 1e-13  # matrix-pipe form against the vector form on the same device (same logs within 1 ULP, sums in another order): measured 1.4e-15
MFCC_RTOL = 1e-12  # x the frame set's largest band log-energy: device log (mxg_log.h, < 0.75 ULP) vs glibc log, then the 42-term DCT


def test_config4_all_frames_vs_oracle(mx, port):
    """Config 4 at its full size, EVERY frame against the oracle: 1 048 576 frames x 1024 points of SURVEY 8d's signal through
    the fused kernel (mxg_fft_mfcc_batch, magnitudes requested as well); per frame the oracle runs maxiFFT(1024,1024,1024)
    (window, packed real FFT with its fp32 twiddle recurrences, cartToPol) and maxiMFCC(512,42,13,20,20000):
    magnitudes bit-exact (fp32), mfcc within MFCC_RTOL.  The host side shards the frames over its cores."""
    import torch
    N = 1 << 20
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0x4D415849)
    sig = torch.empty(N * 1024, dtype=torch.float32, device=dev)
    chunk = 1 << 16
    for c0 in range(0, N, chunk):
        n = torch.arange(c0 * 1024, (c0 + chunk) * 1024, dtype=torch.float64, device=dev)
        k = torch.div(n, 1024, rounding_mode="floor")
        x = (0.4 * torch.sin(2 * np.pi * 220 * n / 44100) + 0.3 * torch.sin(2 * np.pi * (440 + 0.01 * k) * n / 44100)
             + 0.1 * (2 * torch.rand(n.numel(), dtype=torch.float64, device=dev, generator=g) - 1))
        sig[c0 * 1024:(c0 + chunk) * 1024] = x.to(torch.float32)
        del n, k, x
    mags = torch.empty((N, 512), dtype=torch.float32, device=dev)
    mfcc = torch.empty((N, 13), dtype=torch.float64, device=dev)
    L = mx.lib()
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC()
    m.setup(512, 42, 13, 20.0, 20000.0)
    torch.cuda.synchronize()
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, mags.data_ptr(), None, None, mfcc.data_ptr(), None) == 0
    L.mxg_sync()
    # the call without magnitudes (the half-spectrum kernel): the vector form (fused_mel 1) must give the same coefficients bit for bit;
    # the DEFAULT form (mel contraction + DCT on the matrix pipe when only the coefficients are requested) every one of the 2^20 frames
    # within the matrix form's stated distance of them
    mfcc2 = torch.empty_like(mfcc)
    prev = L.mxg_tune(b"fused_mel", 1)
    try:
        assert L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, None, None, None, mfcc2.data_ptr(), None) == 0
        L.mxg_sync()
    finally:
        L.mxg_tune(b"fused_mel", prev)
    assert torch.equal(mfcc, mfcc2)
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, None, None, None, mfcc2.data_ptr(), None) == 0
    L.mxg_sync()
    d_auto = (mfcc2 - mfcc).abs().max().item()
    print("config 4, default form (matrix pipe) vs vector form over %d frames: max |difference| %.3e (bound %.0e)" % (N, d_auto, MM_FORM_ATOL))
    assert d_auto <= MM_FORM_ATOL  # (measured 1.4e-15: the bound is what the code does, ~50 x)
    task = 4096
    bad_mags, worst, worst_auto, scale = 0, 0.0, 0.0, 0.0
    with ThreadPoolExecutor(NTHREADS) as pool:
        for c0 in range(0, N, chunk):
            hs = sig[c0 * 1024:(c0 + chunk) * 1024].cpu().numpy()
            hm = mags[c0:c0 + chunk].cpu().numpy()
            hc = mfcc[c0:c0 + chunk].cpu().numpy()
            ha = mfcc2[c0:c0 + chunk].cpu().numpy()  # the DEFAULT form (matrix pipe), compared with the oracle directly as well

            def one(t0):
                e = port.fft_stream(hs[t0 * 1024:(t0 + task) * 1024], 1024, 1024, 1024, want=("mags",))["mags"]
                assert e.shape == (task, 512)
                nb = int((e.view(np.uint32) != hm[t0:t0 + task].view(np.uint32)).sum())
                emel, emf = port.mfcc(e, 42, 13, 20.0, 20000.0)
                return nb, float(np.abs(hc[t0:t0 + task] - emf).max()), float(np.abs(emel).max()), float(np.abs(ha[t0:t0 + task] - emf).max())
            for nb, err, sc, err_a in pool.map(one, range(0, chunk, task)):
                bad_mags += nb
                worst = max(worst, err)
                worst_auto = max(worst_auto, err_a)
                scale = max(scale, sc)
            assert bad_mags == 0, "magnitudes differ in frames [%d, %d)" % (c0, c0 + chunk)
    print("config 4: %d frames, all %d magnitudes bit-identical to the oracle; mfcc max |err| %.3e (tolerance %.1e x %.2f)"
          % (N, N * 512, worst, MFCC_RTOL, scale))
    assert worst <= MFCC_RTOL * scale
    print("config 4, DEFAULT form (mel contraction + DCT on the matrix pipe) against the oracle, every frame: mfcc max |err| %.3e (same tolerance)"
          % worst_auto)
    assert worst_auto <= MFCC_RTOL * scale


def test_config5_all_streams(mx, port):
    S, T, Ls = 2048, 70560, 4410000
    rng = np.random.default_rng(0x4D415849)
    n = np.arange(Ls)
    smp = 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) \
        + 0.05 * rng.uniform(-1, 1, Ls)
    sb = mx.maxiSampleBank(1)
    sb.setSample(smp)
    bank = mx.maxiTimeStretchBank(S, sb, "hann")
    pos01 = np.arange(S) / S
    bank.setPosition(pos01)
    speed = 0.25 + 1.5 * (np.arange(S) % 97) / 96
    got = bank.play(speed, 0.05, 4, T).numpy()
    gst, gg = bank.state.numpy(), bank.grains.numpy()
    sh = shards(S, max(NTHREADS, 8))

    def one(ab):
        a, b = ab
        st0 = np.zeros((4, b - a))
        st0[0] = np.clip(pos01[a:b] * Ls, 0, Ls - 1)
        e, st, g, rc = port.granular(0, 0, smp, T, speed[a:b], grainLength=0.05, overlaps=4, st=st0)
        assert rc == 0
        return (count_mismatch(got[:, a:b], e), count_mismatch(gst[:, a:b], st), count_mismatch(gg[:, :, a:b], g),
                int((np.diff((e != 0).astype(np.int8), axis=0) != 0).sum()))
    with ThreadPoolExecutor(len(sh)) as pool:
        r = np.array(list(pool.map(one, sh)))
    assert r[:, 0].sum() == 0 and r[:, 1].sum() == 0 and r[:, 2].sum() == 0, r[:, :3].sum(axis=0)
    print("config 5: %d streams x %d samples = %.1f M stream-samples, outputs / scheduler state / live grains all "
          "bit-identical to the oracle" % (S, T, S * T / 1e6))
