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

from conftest import assert_bits_equal

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
