"""GPU parity (-m gpu) at BASELINE.json's FULL sizes for configs 4 and 5 (configs 2 and 3 have theirs in
test_gpu_osc.py / test_gpu_voice.py): the whole workload runs once through the C-ABI, a strided sample of
units is compared with the oracle, and size-independent properties cover the rest."""
import numpy as np
import pytest

from conftest import assert_bits_equal, mix_tol

pytestmark = pytest.mark.gpu

MFCC_RTOL = 1e-12


def test_config4_full_size_fft_mfcc(mx, port):
    """Config 4: 1 048 576 frames x 1024 points, maxiFFT(1024, 1024, 1024) -> maxiMFCC(512, 42, 13, 20, 20000).
    The signal (two sines + uniform noise) is synthesised on the device with torch; 97 frames spread
    over the batch go through the oracle: magnitudes bit-exact, mfcc within the log tolerance; the first and
    last 4 KiB-aligned copies of a repeated segment must give identical bits."""
    import torch
    N = 1 << 20
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0x4D415849)
    sig = torch.empty(N * 1024, dtype=torch.float32, device=dev)
    chunk = 1 << 16                                              # frames per generation chunk
    for c0 in range(0, N, chunk):
        n = torch.arange(c0 * 1024, (c0 + chunk) * 1024, dtype=torch.float64, device=dev)
        k = torch.div(n, 1024, rounding_mode="floor")
        x = (0.4 * torch.sin(2 * np.pi * 220 * n / 44100) + 0.3 * torch.sin(2 * np.pi * (440 + 0.01 * k) * n / 44100)
             + 0.1 * (2 * torch.rand(n.numel(), dtype=torch.float64, device=dev, generator=g) - 1))
        sig[c0 * 1024:(c0 + chunk) * 1024] = x.to(torch.float32)
        del n, k, x
    sig[(N - 4) * 1024:] = sig[:4 * 1024]                        # last 4 frames repeat the first 4
    mags = torch.empty((N, 512), dtype=torch.float32, device=dev)
    mfcc = torch.empty((N, 13), dtype=torch.float64, device=dev)
    L = mx.lib()
    f = mx.maxiFFT()
    f.setup(1024, 1024, 1024)
    m = mx.maxiMFCC()
    m.setup(512, 42, 13, 20.0, 20000.0)
    torch.cuda.synchronize()
    assert L.mxg_fft_batch(f.plan, sig.data_ptr(), 1024, N, None, None, mags.data_ptr(), None, None) == 0
    assert L.mxg_mfcc_batch(m.plan, mags.data_ptr(), 512, N, None, None, mfcc.data_ptr(), 0, None) == 0
    L.mxg_sync()
    sel = np.unique(np.concatenate([np.arange(0, N, 10847), [N - 1, N - 4, 4, 65535, 65536]]))
    tsel = torch.from_numpy(sel).to(dev)
    hs = sig.view(N, 1024)[tsel].cpu().numpy()
    hm, hc = mags[tsel].cpu().numpy(), mfcc[tsel].cpu().numpy()
    for i in range(sel.size):
        e = port.fft_stream(hs[i], 1024, 1024, 1024, want=("mags",))["mags"]
        assert e.shape[0] == 1
        assert np.array_equal(hm[i].view(np.uint32), e[0].view(np.uint32)), "magnitudes of frame %d" % sel[i]
    emel, emf = port.mfcc(hm, 42, 13, 20.0, 20000.0)
    assert np.abs(hc - emf).max() <= MFCC_RTOL * np.abs(emel).max()
    assert torch.equal(mags[:4], mags[N - 4:]) and torch.equal(mfcc[:4], mfcc[N - 4:])
    assert bool(torch.isfinite(mfcc).all())
    # the fused kernel over the same 1 M frames: every coefficient and every magnitude identical to the two-kernel path
    mags_f = torch.empty((N, 512), dtype=torch.float32, device=dev)
    mfcc_f = torch.empty((N, 13), dtype=torch.float64, device=dev)
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, mags_f.data_ptr(), None, None, mfcc_f.data_ptr(), None) == 0
    L.mxg_sync()
    assert torch.equal(mags_f, mags) and torch.equal(mfcc_f, mfcc)
    mfcc_f.zero_()
    prev = L.mxg_tune(b"fused_mel", 1)  # the vector form of the half-spectrum kernel: the two-kernel path's bits
    try:
        assert L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, None, None, None, mfcc_f.data_ptr(), None) == 0
        L.mxg_sync()
    finally:
        L.mxg_tune(b"fused_mel", prev)
    assert torch.equal(mfcc_f, mfcc)
    assert L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, None, None, None, mfcc_f.data_ptr(), None) == 0  # the default form
    L.mxg_sync()
    assert (mfcc_f - mfcc).abs().max().item() <= 1e-11


def test_config5_full_size_granular_share(mx, port):
    """Config 5, one GPU's share: 2048 maxiTimeStretch<hann> streams over a 100 s sample, 70 560 samples each,
    grainLength 0.05, overlaps 4 (~2.6e5 grains).  33 streams spread over the bank are replayed by the oracle:
    output, scheduler state and live grains bit-exact; the stereo mixdown of the whole block is checked
    against the per-stream outputs it was made from."""
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
    out = bank.play(speed, 0.05, 4, T)
    sel = np.unique(np.concatenate([np.arange(0, S, 67), [S - 1, 96, 97]]))
    ho = out.numpy()[:, sel]
    st0 = np.zeros((4, sel.size))
    st0[0] = np.clip(pos01[sel] * Ls, 0, Ls - 1)
    e, st, gst, rc = port.granular(0, 0, smp, T, speed[sel], grainLength=0.05, overlaps=4, st=st0)
    assert rc == 0
    assert_bits_equal(ho, e, "sampled streams")
    assert_bits_equal(bank.state.numpy()[:, sel], st, "scheduler state")
    assert_bits_equal(bank.grains.numpy()[:, :, sel], gst, "live grains")
    # mixdown of the first 4096 samples: pan = s/(S-1); tree sum vs float64 sum of the same products
    pan = np.arange(S) / (S - 1.0)
    B = 4096
    blk = mx.DeviceBuffer.from_numpy(out.numpy()[:B])
    mix = mx.maxiMixBank(S).stereo(blk, pan).numpy()
    x = blk.numpy()
    ref = np.stack([(x * np.sqrt(1.0 - pan)).sum(1), (x * np.sqrt(pan)).sum(1)], axis=1)
    assert np.abs(mix - ref).max() <= mix_tol(S, np.abs(x).max())
