"""GPU parity (-m gpu): maxiOsc bank through the C-ABI vs the oracle and the golden vectors."""
import numpy as np
import pytest

from conftest import assert_bits_equal, ulp_diff, mix_tol

pytestmark = pytest.mark.gpu

OSC = ["sinewave", "coswave", "phasor", "saw", "triangle", "square", "pulse", "impulse",
       "sinebuf", "sinebuf4", "sawn", "phasorBetween"]
EXACT = [w for w in range(12) if w not in (0, 1)]  # everything but sinewave/coswave is bit-exact
# sinewave/coswave: device sin/cos vs glibc sin/cos of the same rounded argument
TRIG_MAX_ULP = 1


def _render(mx, wf, freq, N, blocks=1, phase0=None, p1=None, p2=None, per_sample=False):
    V = freq.shape[-1]
    bank = mx.maxiOscBank(V)
    if phase0 is not None:
        bank.phaseReset(phase0)
    outs = []
    for b in range(blocks):
        f = freq[b * N:(b + 1) * N] if per_sample else freq
        outs.append(bank.render(wf, f, N, p1=p1, p2=p2, per_sample=per_sample).numpy())
    return np.concatenate(outs), bank.phase.numpy(), bank.output.numpy()


@pytest.mark.parametrize("wf", range(12))
def test_osc_golden(mx, golden, wf):
    g = golden("osc.npz")
    name = OSC[wf]
    N = int(g["N"])
    a, b = (g["duty"], None) if name == "pulse" else (g["p1"], g["p2"])
    out, ph, hd = _render(mx, wf, g["freq"], N, blocks=2, p1=a, p2=b)
    if wf in EXACT:
        assert_bits_equal(out, g["out_" + name], name)
        assert_bits_equal(ph, g["phase_" + name], name + " phase")
        assert_bits_equal(hd, g["hold_" + name], name + " hold")
    else:
        assert ulp_diff(out, g["out_" + name]).max() <= TRIG_MAX_ULP   # every sample, zero crossings included
        assert_bits_equal(ph, g["phase_" + name], name + " phase")


def test_kat_sinewave_440(mx, golden):
    out, _, _ = _render(mx, 0, np.array([440.0]), 4)
    exp = golden("osc.npz")["ka_sinewave440"]
    assert ulp_diff(out[:, 0], exp).max() <= TRIG_MAX_ULP


@pytest.mark.parametrize("name", ["sinebuf", "saw", "sawn"])
def test_osc_fm_golden(mx, golden, name):
    g = golden("osc.npz")
    out, ph, _ = _render(mx, OSC.index(name), g["fm"], int(g["N"]), per_sample=True)
    assert_bits_equal(out, g["fm_out_" + name], name)
    assert_bits_equal(ph, g["fm_phase_" + name], name)


@pytest.mark.parametrize("wf", EXACT)
@pytest.mark.parametrize("V,N,blocks", [(1, 7, 3), (63, 33, 2), (64, 512, 2), (130, 100, 4), (1000, 257, 2)])
def test_osc_vs_oracle_state_carry(mx, port, wf, V, N, blocks):
    rng = np.random.default_rng(1000 * wf + V)
    freq = rng.uniform(0.05, 21000, V)
    p2 = rng.uniform(0.2, 1.0, V)
    p1 = rng.uniform(-0.2, 1.2, V) if wf == 6 else p2 * rng.uniform(0, 0.95, V)
    ph0 = rng.uniform(0, 500, V) if wf in (8, 9) else rng.uniform(0, 1, V)
    out, ph, hd = _render(mx, wf, freq, N, blocks=blocks, phase0=ph0, p1=p1, p2=p2)
    eo, eph, ehd = port.osc(wf, freq, N * blocks, phase=ph0, p1=p1, p2=p2)
    assert_bits_equal(out, eo, OSC[wf])
    assert_bits_equal(ph, eph, OSC[wf] + " phase")
    assert_bits_equal(hd, ehd, OSC[wf] + " hold")


@pytest.mark.parametrize("wf", [0, 1])
def test_trig_osc_ulp(mx, port, wf):
    rng = np.random.default_rng(5 + wf)
    V, N = 512, 2048
    freq = np.concatenate([[440.0, 0.2, 20000.0, 11025.0], rng.uniform(0.1, 21000, V - 4)])
    out, ph, _ = _render(mx, wf, freq, N)
    eo, eph, _ = port.osc(wf, freq, N)
    assert_bits_equal(ph, eph, "phase")
    d = ulp_diff(out, eo)
    # <= 1 ULP of the RESULT on every sample: no absolute-error allowance near the zero crossings (round 1 needed one:
    # a wrong third reduction constant left k*2e-21 in the remainder).  11025 Hz = sr/4 puts phases exactly on 0.25,
    # 0.5, 0.75: x = fl(k*pi/2), the hardest arguments.
    assert int(d.max()) <= TRIG_MAX_ULP, (int((d > TRIG_MAX_ULP).sum()), int(d.max()), float(np.abs(out - eo).max()))
    print("sinewave/coswave wf=%d: %d of %d samples differ from glibc (all by 1 ULP), 0 above 1 ULP"
          % (wf, int((d > 0).sum()), d.size))


@pytest.mark.parametrize("vpl,nt,block", [(1, 0, 64), (2, 0, 64), (2, 1, 256), (1, 1, 128)])
def test_tuning_does_not_change_results(mx, port, vpl, nt, block):
    L = mx.lib()
    prev = [L.mxg_tune(b"osc_vpl", vpl), L.mxg_tune(b"osc_nt", nt), L.mxg_tune(b"osc_block", block)]
    try:
        rng = np.random.default_rng(77)
        V, N = 4096 + 2, 300
        freq = rng.uniform(20, 20000, V)
        for wf in (8, 9, 10, 3):
            out, ph, hd = _render(mx, wf, freq, N, blocks=2)
            eo, eph, _ = port.osc(wf, freq, 2 * N)
            assert_bits_equal(out, eo, OSC[wf])
            assert_bits_equal(ph, eph)
    finally:
        L.mxg_tune(b"osc_vpl", prev[0]); L.mxg_tune(b"osc_nt", prev[1]); L.mxg_tune(b"osc_block", prev[2])


@pytest.mark.parametrize("vpl,store,block,xcd", [(1, 3, 256, 1), (1, 4, 256, 2), (1, 5, 128, 1), (1, 4, 64, 1), (2, 3, 256, 2),
                                                 (2, 1, 512, 2), (1, 1, 256, 2), (1, 4, 1024, 2), (0, 0, 256, 0)])
@pytest.mark.parametrize("V,N", [(4096 + 2, 301), (512, 2), (8192, 64), (4097, 33)])
def test_store_streams_do_not_change_results(mx, port, vpl, store, block, xcd, V, N):
    """The store stream of K1 is a launch knob (osc_store: 8-byte plain / nt, the 16-byte pair-row exchange in three flavours,
    16-byte stores of two voices per lane; osc_xcd: XCD-contiguous workgroup numbering): every choice must give the oracle's
    bits, odd block lengths (a trailing single row), odd banks (pair rows fall back to 8-byte stores) and time parts included."""
    L = mx.lib()
    prev = [L.mxg_tune(b"osc_vpl", vpl), L.mxg_tune(b"osc_store", store), L.mxg_tune(b"osc_block", block), L.mxg_tune(b"osc_xcd", xcd)]
    try:
        rng = np.random.default_rng(V + N)
        freq = rng.uniform(20, 20000, V)
        for wf in (8, 9, 2, 0):
            out, ph, hd = _render(mx, wf, freq, N, blocks=2)
            eo, eph, _ = port.osc(wf, freq, 2 * N)
            if wf == 0:
                assert ulp_diff(out, eo).max() <= TRIG_MAX_ULP
            else:
                assert_bits_equal(out, eo, OSC[wf])
            assert_bits_equal(ph, eph)
    finally:
        L.mxg_tune(b"osc_vpl", prev[0]); L.mxg_tune(b"osc_store", prev[1]); L.mxg_tune(b"osc_block", prev[2])
        L.mxg_tune(b"osc_xcd", prev[3])


@pytest.mark.parametrize("vpl,store,passes,V,N", [(1, 4, 2, 4096 + 2, 301), (2, 3, 3, 70000, 33), (1, 2, 5, 4097, 64), (1, 4, 64, 512, 8),
                                                    (2, 3, 2, 131072, 16), (1, 4, 3, 98304, 16)])
def test_passes_same_bits(mx, port, vpl, store, passes, V, N):
    """Knob osc_passes: the grid covers 1 / passes of the bank and every wavefront renders `passes` voice groups one after the other --
    the oracle's bits and carried state for every grouping, ragged last groups and more passes than workgroups included."""
    L = mx.lib()
    prev = [L.mxg_tune(b"osc_vpl", vpl), L.mxg_tune(b"osc_store", store), L.mxg_tune(b"osc_passes", passes)]
    try:
        rng = np.random.default_rng(V + passes)
        freq = rng.uniform(20, 20000, V)
        for wf in ((8, 2, 0) if V < 50000 else (8,)):
            out, ph, hd = _render(mx, wf, freq, N, blocks=2)
            eo, eph, _ = port.osc(wf, freq, 2 * N)
            if wf == 0:
                assert ulp_diff(out, eo).max() <= TRIG_MAX_ULP
            else:
                assert_bits_equal(out, eo, OSC[wf])
            assert_bits_equal(ph, eph, "phase")
    finally:
        L.mxg_tune(b"osc_vpl", prev[0]); L.mxg_tune(b"osc_store", prev[1]); L.mxg_tune(b"osc_passes", prev[2])


@pytest.mark.parametrize("wf,V,N", [(8, 2 * 98304 + 4098, 256), (3, 98304 + 65536 + 2, 300), (8, 131072, 352), (10, 3 * 98304, 176),
                                      (8, 73728 + 2, 384), (8, 81920, 320), (10, 81920, 320), (8, 98304 + 73728, 256), (8, 86016, 300)])
def test_large_bank_launch_plan_same_bits(mx, port, wf, V, N):
    """Banks beyond ~350 MB per block are rendered as a PLAN of launches (osc.hip: passes of 98 304 voices on a grid of three wavefronts
    per CU, then the remainder in the shape that suits its size): two carried blocks, every 499th voice and the voices around every
    boundary of the plan against the oracle, the carried phase of all of them.  (Sizes between 65 536 and 98 304 voices: the two-pass
    non-temporal form of sinebuf, sawn's pair rows, on their own and as the remainder of a plan.)"""
    rng = np.random.default_rng(V)
    freq = rng.uniform(20, 20000, V)
    bank = mx.maxiOscBank(V)
    o1 = bank.render(wf, freq, N).numpy()
    o2 = bank.render(wf, freq, N).numpy()
    edges = [k * 98304 + d for k in range(1, V // 98304 + 1) for d in (-2, -1, 0, 1) if 0 <= k * 98304 + d < V]
    edges += [V // 2 + d for d in (-130, -129, -128, -2, -1, 0, 1, 127, 128, 129)]  # (a two-pass launch splits the range near its middle)
    sel = np.unique(np.concatenate([np.arange(0, V, 499), edges, [V - 2, V - 1]]).astype(np.int64))
    eo, eph, ehd = port.osc(wf, freq[sel], 2 * N)
    assert_bits_equal(np.concatenate([o1[:, sel], o2[:, sel]]), eo, OSC[wf])
    assert_bits_equal(bank.phase.numpy()[sel], eph, "phase")
    assert_bits_equal(bank.output.numpy()[sel], ehd, "output member")


@pytest.mark.parametrize("wf,V,N", [(3, 98304 + 2, 200), (5, 131072, 136), (4, 196608 + 64, 77), (2, 262144, 64), (6, 131074, 100), (11, 100000, 130),
                                      (8, 131072, 120), (8, 122880 + 6, 90), (8, 327680, 40)])
def test_paced_launch_same_bits(mx, port, wf, V, N):
    """The table-free waveforms at 90 112 ... 327 680 voices (sinebuf from 122 880) are rendered by ONE simple launch on the paced schedule (csrc/mxg_pace.h:
    eight samples every P ticks of the 100 MHz counter, P from a controller in device scratch that the kernel updates): three carried
    blocks against the round-4 launch rules (knob osc_pace 1) bit for bit -- the schedule is timing only -- with a fixed period as
    well, ragged block lengths, and a subsample of voices against the oracle."""
    L = mx.lib()
    rng = np.random.default_rng(V + wf)
    freq = rng.uniform(20, 20000, V)
    p1 = rng.uniform(0.1, 0.9, V) if wf in (6, 11) else None
    p2 = rng.uniform(0.1, 0.9, V) if wf == 11 else None
    if wf == 11:
        p1, p2 = np.minimum(p1, p2), np.maximum(p1, p2) + 0.05

    def run(pace):
        prev = L.mxg_tune(b"osc_pace", pace)
        try:
            bank = mx.maxiOscBank(V)
            outs = [bank.render(wf, freq, N, p1=p1, p2=p2).numpy() for _ in range(3)]
            return np.concatenate(outs), bank.phase.numpy().copy(), bank.output.numpy().copy()
        finally:
            L.mxg_tune(b"osc_pace", prev)
    ref = run(1)
    for pace in (0, 60):
        got = run(pace)
        for a, b, what in zip(ref, got, ("blocks", "phase", "output member")):
            assert_bits_equal(b, a, "osc_pace=%d, %s" % (pace, what))
    sel = np.unique(np.concatenate([np.arange(0, V, 997), [V - 2, V - 1]]).astype(np.int64))
    eo, eph, _ = port.osc(wf, freq[sel], 3 * N, p1=None if p1 is None else p1[sel], p2=None if p2 is None else p2[sel])
    assert_bits_equal(ref[0][:, sel], eo, OSC[wf])
    assert_bits_equal(ref[1][sel], eph, "phase")


def test_empty_and_invalid(mx):
    L = mx.lib()
    bank = mx.maxiOscBank(4)
    out = mx.DeviceBuffer((4, 4))
    f = mx.DeviceBuffer.from_numpy(np.full(4, 100.0))
    assert L.mxg_osc_render(8, 4, 0, f.ptr, 0, None, None, bank.phase.ptr, bank.output.ptr, out.ptr, None) == 0
    assert L.mxg_osc_render(8, 0, 4, f.ptr, 0, None, None, bank.phase.ptr, bank.output.ptr, out.ptr, None) == 0
    assert L.mxg_osc_render(12, 4, 4, f.ptr, 0, None, None, bank.phase.ptr, bank.output.ptr, out.ptr, None) == -1
    assert L.mxg_osc_render(6, 4, 4, f.ptr, 0, None, None, bank.phase.ptr, bank.output.ptr, out.ptr, None) == -1
    assert L.mxg_osc_render(8, 4, 4, None, 0, None, None, bank.phase.ptr, bank.output.ptr, out.ptr, None) == -1
    assert b"null" in L.mxg_last_error()


def test_config2_full_size_properties(mx, port):
    """BASELINE config 2 at full size: 65 536 voices, B=512, several blocks with state carried.
    Checked through size-independent properties + a strided sample of voices against the oracle."""
    V, B, K = 65536, 512, 4
    freq = 20.0 + np.arange(V) * 0.30517578125
    bank = mx.maxiOscBank(V)
    blocks = [bank.sinebuf(freq, B).numpy() for _ in range(K)]
    full = np.concatenate(blocks)
    # (1) splitting the render differently gives the same stream (state carry is exact)
    bank2 = mx.maxiOscBank(V)
    a = bank2.sinebuf(freq, 3 * B).numpy()
    b = bank2.sinebuf(freq, B).numpy()
    assert_bits_equal(np.concatenate([a, b]), full, "split invariance")
    assert_bits_equal(bank.phase.numpy(), bank2.phase.numpy(), "phase")
    # (2) range: the table spans [-0.99997, 0.99997], linear interpolation cannot exceed it
    assert np.abs(full).max() <= 0.99997
    # (3) every 97th voice (+ first/last) against the oracle, every sample
    sel = np.unique(np.concatenate([np.arange(0, V, 97), [V - 1]]))
    eo, eph, _ = port.osc(8, freq[sel], K * B)
    assert_bits_equal(full[:, sel], eo, "sampled voices")
    assert_bits_equal(bank.phase.numpy()[sel], eph, "sampled phases")
    # (4) phase invariant of sinebuf: always in [-1, 511)
    ph = bank.phase.numpy()
    assert ph.min() >= -1.0 and ph.max() < 511.0


@pytest.mark.parametrize("wf,V,N", [(8, 300, 100), (3, 64, 16), (10, 1000, 37), (9, 4096 + 7, 512), (6, 256, 1), (8, 700, 1100),
                                      (4, 130, 530), (8, 40000, 512), (10, 33000, 48)])
def test_render_mix_fused(mx, port, wf, V, N):
    """mxg_osc_render_mix: per-voice block bit-exact (same as the plain render), mix within the
    stated tolerance of the reference's sequential sum, store=False gives the same mix bits."""
    rng = np.random.default_rng(900 + wf)
    freq = rng.uniform(20, 15000, V)
    pan = rng.uniform(-0.1, 1.1, V)
    p1 = rng.uniform(0.1, 0.9, V)
    bank = mx.maxiOscBank(V)
    out1, mix1 = bank.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
    out2, mix2 = bank.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
    o = np.concatenate([out1.numpy(), out2.numpy()])
    m = np.concatenate([mix1.numpy(), mix2.numpy()])
    eo, eph, _ = port.osc(wf, freq, 2 * N, p1=p1, p2=np.ones(V))
    assert_bits_equal(o, eo, OSC[wf])
    assert_bits_equal(bank.phase.numpy(), eph, "phase")
    em = port.mix_stereo(eo, pan)
    assert np.abs(m - em).max() <= mix_tol(V, np.abs(eo).max(), sums=em)  # (voices started together: coherent first samples)
    bank2 = mx.maxiOscBank(V)
    none, mix3 = bank2.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V), store=False)
    assert none is None
    assert_bits_equal(mix3.numpy(), mix1.numpy(), "mix-only mode")
    # launch knobs of K1m: the per-voice block as pair rows of 16-byte stores or plain stores, one to four time parts --
    # the block, the carried phase AND the mix (same additions in the same tree) must not change by a bit
    L = mx.lib()
    for store, split in ((2, 1), (1, 1), (2, 2), (1, 3), (2, 4)):
        prev = [L.mxg_tune(b"osc_mix_store", store), L.mxg_tune(b"osc_mix_split", split)]
        try:
            bank3 = mx.maxiOscBank(V)
            o3, m3 = bank3.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
            assert_bits_equal(o3.numpy(), out1.numpy(), "store %d split %d" % (store, split))
            assert_bits_equal(m3.numpy(), mix1.numpy(), "mix, store %d split %d" % (store, split))
            o3, m3 = bank3.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
            assert_bits_equal(o3.numpy(), out2.numpy(), "second block, store %d split %d" % (store, split))
            assert_bits_equal(bank3.phase.numpy(), eph, "phase, store %d split %d" % (store, split))
        finally:
            L.mxg_tune(b"osc_mix_store", prev[0]); L.mxg_tune(b"osc_mix_split", prev[1])
    # the two forms of the kernel -- one wavefront does everything (osc_mix_pc 1) / producer + consumer wavefront pairs (2) -- add the
    # same products in the same tree: the same bits for the block, the carried state and the mix
    for pc, pcwin in ((1, 0), (2, 256), (2, 512)):   # (osc_mix_pcwin: the producer / consumer form's combine window)
        prev = L.mxg_tune(b"osc_mix_pc", pc)
        prev_win = L.mxg_tune(b"osc_mix_pcwin", pcwin)
        try:
            bank6 = mx.maxiOscBank(V)
            o6, m6 = bank6.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
            assert_bits_equal(o6.numpy(), out1.numpy(), "osc_mix_pc %d" % pc)
            assert_bits_equal(m6.numpy(), mix1.numpy(), "mix, osc_mix_pc %d" % pc)
            o6, m6 = bank6.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
            assert_bits_equal(o6.numpy(), out2.numpy(), "second block, osc_mix_pc %d" % pc)
            assert_bits_equal(m6.numpy(), mix2.numpy(), "second mix, osc_mix_pc %d" % pc)
            assert_bits_equal(bank6.phase.numpy(), eph, "phase, osc_mix_pc %d" % pc)
            none, m7 = bank6.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V), store=False)
        finally:
            L.mxg_tune(b"osc_mix_pc", prev)
            L.mxg_tune(b"osc_mix_pcwin", prev_win)
    # passes (a workgroup renders several groups of 256 voices one after the other): the same rows, the same bits
    for mp in (2, 5):
        prev = L.mxg_tune(b"osc_mix_passes", mp)
        try:
            bank5 = mx.maxiOscBank(V)
            o5, m5 = bank5.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
            assert_bits_equal(o5.numpy(), out1.numpy(), "passes %d" % mp)
            assert_bits_equal(m5.numpy(), mix1.numpy(), "mix, passes %d" % mp)
            o5, m5 = bank5.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V))
            assert_bits_equal(o5.numpy(), out2.numpy(), "second block, passes %d" % mp)
            assert_bits_equal(bank5.phase.numpy(), eph, "phase, passes %d" % mp)
        finally:
            L.mxg_tune(b"osc_mix_passes", prev)
    # the rows form (what a grouped mix queue's slot receives): rows [ceil(V / 256)][N][2], added by mxg_mix_rows_sum = the same mix
    # bits; the combine window (osc_mix_win 128 / 256) changes neither the rows nor the block
    G = L.mxg_osc_mix_groups(V)
    assert G == (V + 255) // 256
    for win in (128, 256):
        prev = L.mxg_tune(b"osc_mix_win", win)
        try:
            bank4 = mx.maxiOscBank(V)
            rows = mx.DeviceBuffer((G, N, 2))
            o4, none = bank4.render_mix(wf, freq, pan, N, p1=p1, p2=np.ones(V), rows=rows)
            assert none is None
            assert_bits_equal(o4.numpy(), out1.numpy(), "rows form, window %d" % win)
            msum = mx.DeviceBuffer((N, 2))
            mx._lib.check(L.mxg_mix_rows_sum(G, N * 2, rows.ptr, msum.ptr, None), "mxg_mix_rows_sum")
            assert_bits_equal(msum.numpy(), mix1.numpy(), "sum of the rows, window %d" % win)
            r = rows.numpy()
            for g in range(G):  # every row is the mix of its own 256 voices
                sl = slice(256 * g, min(V, 256 * g + 256))
                eg = port.mix_stereo(eo[:N, sl], pan[sl])
                assert np.abs(r[g] - eg).max() <= mix_tol(256, np.abs(eo).max())
        finally:
            L.mxg_tune(b"osc_mix_win", prev)


@pytest.mark.parametrize("wf,V,N", [(0, 1000, 601), (1, 64, 3), (8, 300, 512), (6, 129, 77), (10, 70, 1), (5, 4096, 130)])
def test_osc_time_split_same_bits(mx, wf, V, N):
    """mxg_tune("osc_split"): rendering a block in 2 or 4 time parts (part p first advances the recurrence over the
    earlier parts' samples without producing them) gives the same samples, phases and output members as one part --
    two consecutive blocks, so the state written by the last part is exercised."""
    rng = np.random.default_rng(wf * 7 + N)
    freq, p1 = rng.uniform(20, 15000, V), rng.uniform(0.1, 0.9, V)
    L = mx.lib()
    prev = L.mxg_tune(b"osc_split", 1)
    ref = None
    try:
        for split in (1, 2, 4, 0):
            L.mxg_tune(b"osc_split", split)
            bank = mx.maxiOscBank(V)
            o = np.concatenate([bank.render(wf, freq, N, p1=p1, p2=np.ones(V)).numpy() for _ in range(2)])
            cur = (o, bank.phase.numpy(), bank.output.numpy())
            if ref is None:
                ref = cur
            else:
                for a, b, what in zip(cur, ref, ("samples", "phase", "output member")):
                    assert_bits_equal(a, b, "%s, split %d" % (what, split))
    finally:
        L.mxg_tune(b"osc_split", prev)


@pytest.mark.parametrize("V,N,rw", [(96, 257, 0), (96, 257, 2), (96, 257, 3), (100, 33, 4), (98, 40, 3), (4, 1, 3), (2, 2, 4), (97, 12, 3), (4096, 513, 4), (700, 16, 2), (700, 17, 3)])
def test_noise_from_rand_draws(mx, port, V, N, rw):
    """maxiOsc::noise (C:214-220): the caller supplies the rand() draws, the float arithmetic is exact.  rw: knob rw_store -- the
    column walk of round 4 (a lane owns two voices and every other row; odd banks keep the element-wise kernel)."""
    rnd, e = port.noise(1234 + V, V, N)
    bank = mx.maxiOscBank(V)
    prev = mx.lib().mxg_tune(b"rw_store", rw)
    prev_chunk = mx.lib().mxg_tune(b"rw_chunk", (0, 0, 4, 16, 32)[rw])   # (rows per lane and chunk of the column walk)
    try:
        o = bank.noise(rnd).numpy()
    finally:
        mx.lib().mxg_tune(b"rw_store", prev)
        mx.lib().mxg_tune(b"rw_chunk", prev_chunk)
    assert_bits_equal(o, e, "noise")
    assert_bits_equal(bank.output.numpy(), e[-1], "noise output member")
    # extremes of the int -> float conversion: 0, RAND_MAX (rounds to 2^31 -> r == 1), odd ties
    edge = np.array([[0, 2147483647, 2147483583, 2147483584, 16777217, 16777219, 1, 33554434]], np.int32)
    o = mx.maxiOscBank(8).noise(edge).numpy()
    r = (edge.astype(np.float32) / np.float32(2147483648.0))
    assert_bits_equal(o, (r * np.float32(2) - np.float32(1)).astype(np.float64), "noise edges")


def test_render_and_mixdown_are_graph_capturable(mx):
    """The per-block launch sequence (K1 render + K3 mixdown: gains kernel, mix kernel) makes no synchronising or
    allocating HIP call in steady state, so it can be captured into a hipGraph on the caller's stream and
    replayed; K replays must leave the same state and mix as K eager launches."""
    import torch
    L = mx.lib()
    V, B, K = 4096, 128, 5
    dev = torch.device("cuda", 0)
    freq = torch.from_numpy(20 + np.arange(V) * 4.8828125).to(dev)
    pan = torch.from_numpy(np.arange(V) / (V - 1.0)).to(dev)

    def fresh():
        return (torch.zeros(V, dtype=torch.float64, device=dev), torch.zeros(V, dtype=torch.float64, device=dev),
                torch.empty((B, V), dtype=torch.float64, device=dev), torch.zeros((B, 2), dtype=torch.float64, device=dev))

    def block(st, phase, hold, out, mix):
        assert L.mxg_osc_render(8, V, B, freq.data_ptr(), 0, None, None, phase.data_ptr(), hold.data_ptr(),
                                out.data_ptr(), st) == 0
        assert L.mxg_mix_stereo(V, B, out.data_ptr(), pan.data_ptr(), mix.data_ptr(), st) == 0

    s = torch.cuda.Stream(device=dev)
    st = s.cuda_stream
    # eager reference: K blocks
    e = fresh()
    with torch.cuda.stream(s):
        for _ in range(K):
            block(st, *e)
    s.synchronize()
    # captured: one warm-up block on scratch state (allocates the library scratch for this stream), then capture
    g_state = fresh()
    warm = fresh()
    with torch.cuda.stream(s):
        block(st, *warm)
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        block(st, *g_state)
    for _ in range(K):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_state[0], e[0]), "phase after K replays"
    assert torch.equal(g_state[2], e[2]), "last block"
    assert torch.equal(g_state[3], e[3]), "last mix"


def test_headline_trial_same_bits(mx, port):
    """sinebuf at the headline's size is launched as ONE kernel that holds the free-running pair-row stream and the paced 8-byte one; a trial
    on the device (csrc/mxg_pace.h, PaceTrial: phases of 32 launches: free-running, then descending periods for as long as each is faster than the one before) picks by the
    measured durations.  180 carried blocks through the trial and its verdict against the free-running kernel alone (knob osc_pace 1):
    every block of the run and the state, bit for bit; the verdict is there within 160 launches; a subsample against the oracle."""
    import ctypes
    L = mx.lib()
    V, N, K = 65536, 512, 180       # (a block the automatic rule streams through pair rows: 268 MB; the verdict comes within 160 launches)
    rng = np.random.default_rng(5)
    freq = rng.uniform(20, 20000, V)

    def run(pace):
        prev = L.mxg_tune(b"osc_pace", pace)
        try:
            bank = mx.maxiOscBank(V)
            acc = np.zeros(V)
            last = None
            for k in range(K):
                o = bank.render(8, freq, N)
                if k % 16 == 5 or k == K - 1:                # (blocks of every phase of the trial, and the last one)
                    last = o.numpy()
                    acc = acc + last[::7].sum(axis=0) * (k + 1)
            buf = (ctypes.c_uint * 128)()
            assert L.mxg_debug_osc_pace(bank.stream, buf) == 0
            return acc, last, bank.phase.numpy().copy(), bank.output.numpy().copy(), list(buf)[8 * 14: 8 * 14 + 8]
        finally:
            L.mxg_tune(b"osc_pace", prev)
    ref = run(1)
    got = run(0)
    for a, b, what in zip(ref[:4], got[:4], ("fingerprint of every 16th block", "last block", "phase", "output member")):
        assert_bits_equal(b, a, "trial against the free-running kernel: %s" % what)
    t = got[4]
    assert t[0] == 1 or 40 <= t[0] <= 70, "the trial's words after %d more launches: %r" % (K, t)
    sel = np.arange(0, V, 997)
    eo, eph, _ = port.osc(8, freq[sel], K * N)
    assert_bits_equal(ref[1][:, sel], eo[-N:], "last block against the oracle")
    assert_bits_equal(ref[2][sel], eph, "phase against the oracle")


def test_paced_launches_are_graph_capturable(mx):
    """A launch on the paced schedule (saw at 131 072 voices: csrc/mxg_pace.h) takes its controller's words from per-stream scratch.  Captured
    into a hipGraph on a stream that has never launched it, it must not allocate inside the capture -- the launch is then simply not
    paced; captured after an eager launch it carries the words, and the replays keep updating them.  Same bits either way."""
    import torch
    L = mx.lib()
    V, B, K = 131072, 64, 4
    dev = torch.device("cuda", 0)
    freq = torch.from_numpy(20 + (np.arange(V) % 4096) * 4.8828125).to(dev)

    def fresh():
        return (torch.zeros(V, dtype=torch.float64, device=dev), torch.zeros(V, dtype=torch.float64, device=dev),
                torch.empty((B, V), dtype=torch.float64, device=dev))

    def block(st, phase, hold, out):
        assert L.mxg_osc_render(3, V, B, freq.data_ptr(), 0, None, None, phase.data_ptr(), hold.data_ptr(), out.data_ptr(), st) == 0

    ref = fresh()
    s0 = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s0):
        for _ in range(K):
            block(s0.cuda_stream, *ref)
    s0.synchronize()
    for warm in (False, True):
        s = torch.cuda.Stream(device=dev)      # a stream the library has no scratch for
        st = s.cuda_stream
        if warm:
            w = fresh()
            with torch.cuda.stream(s):
                block(st, *w)
            s.synchronize()
        g_state = fresh()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            block(st, *g_state)
        for _ in range(K):
            graph.replay()
        torch.cuda.synchronize()
        for a, b, what in zip(ref, g_state, ("phase", "output member", "block")):
            assert torch.equal(a, b), "graph replays (%s eager launch first): %s" % ("an" if warm else "no", what)


def test_time_part_timeout_is_reported_and_state_kept(mx, port):
    """A time-split launch whose writer part does not get its sibling parts' signals (forced: knob part_fault makes it wait for
    one signal more than will ever come, part_spin_limit bounds the wait) must fail LOUDLY: the next synchronising call returns
    MXG_ERR_HIP with a message, the per-voice state of that launch is NOT stored (so no later launch starts from a state a late
    part may not have seen), and the following launches work again with zeroed counters and give the oracle's bits."""
    L = mx.lib()
    V, N = 4096, 256
    freq = 20.0 + np.arange(V) * 0.3
    bank = mx.maxiOscBank(V)
    L.mxg_tune(b"osc_split", 2)
    first = bank.render(8, freq, N).numpy()            # a clean split launch: sinebuf in two time parts
    ph1 = bank.phase.numpy()
    e1, eph1, _ = port.osc(8, freq, N)
    assert_bits_equal(first, e1)
    assert_bits_equal(ph1, eph1)
    L.mxg_tune(b"part_fault", 1)
    L.mxg_tune(b"part_spin_limit", 50)
    try:
        bank.render(8, freq, N)                        # enqueues; the kernel times out on the device
        rc = L.mxg_stream_sync(bank.stream)
        assert rc < 0, "the time-out was not reported at the synchronising call"
        assert b"timed out" in L.mxg_last_error()
        assert L.mxg_last_async_error() == 0           # reported once, then cleared
    finally:
        L.mxg_tune(b"part_fault", 0)
        L.mxg_tune(b"part_spin_limit", 1 << 20)
    assert_bits_equal(bank.phase.numpy(), ph1, "state must not be stored by a launch that timed out")
    second = bank.render(8, freq, N).numpy()           # the block again, from the kept state, counters re-zeroed
    e2, eph2, _ = port.osc(8, freq, N, phase=eph1)
    assert_bits_equal(second, e2)
    third = bank.render(8, freq, N).numpy()
    e3, _, _ = port.osc(8, freq, N, phase=eph2)
    assert_bits_equal(third, e3)
    L.mxg_tune(b"osc_split", 0)


@pytest.mark.parametrize("V", [64, 65536 + 128])
def test_pulse_and_triangle_on_their_corners(mx, port, V):
    """pulse holds its old output when the phase EQUALS the duty (C:308-309: neither `<` nor `>`) and with a NaN duty; the device tick takes
    the sign of phase - duty and handles exactly those cases behind a wave-level test.  Frequencies sr / 8, sr / 16 make the phase walk
    over exact eighths, duties on and off that grid, NaN among them; triangle turns at 0.5 exactly.  Two carried blocks, every launch shape
    of the size (one / two voices per lane)."""
    rng = np.random.default_rng(V)
    freq = np.where(np.arange(V) % 3 == 0, 44100.0 / 8, np.where(np.arange(V) % 3 == 1, 44100.0 / 16, rng.uniform(20, 9000, V)))
    duty = rng.choice([0.0, 0.125, 0.25, 0.5, 0.3, 0.75, 1.0, 1.5, -0.2, np.nan], V)
    N = 96
    for name, p1 in (("pulse", duty), ("triangle", None)):
        bank = mx.maxiOscBank(V)
        o = np.concatenate([bank.render(name, freq, N, p1=p1).numpy(), bank.render(name, freq, N, p1=p1).numpy()])
        eo, eph, ehold = port.osc(OSC.index(name), freq, 2 * N, p1=p1)
        assert_bits_equal(o, eo, name)
        assert_bits_equal(bank.phase.numpy(), eph, name + " phase")
        assert_bits_equal(bank.output.numpy(), ehold, name + " output member")


def test_pulse_width_and_frequency_per_sample(mx, port):
    """mxg_osc_render fps = 2: frequency AND p1 (the pulse width; phasorBetween's start phase) per sample, [N][V] each -- what the
    per-sample engine renders when a patch writes `sound.pulse(f, mod.phasor(1))` (16.Replicant).  Against the oracle called one
    sample at a time with that sample's arguments and the carried state: the same bits."""
    rng = np.random.default_rng(77)
    V, N = 24, 300
    L = mx.lib()
    chk = mx._lib.check
    for name in ("pulse", "phasorBetween"):
        wf = mx.OSC_WAVEFORMS[name]
        freq = rng.uniform(20, 3000, (N, V))
        p1 = rng.uniform(0.05, 0.95, (N, V))
        p2 = np.full(V, 1.5)
        ph, hd = np.zeros(V), np.zeros(V)
        exp = np.empty((N, V))
        for n in range(N):
            o, ph, hd = port.osc(wf, freq[n], 1, phase=ph, hold=hd, p1=p1[n], p2=p2)
            exp[n] = o[0]
        DB = mx.DeviceBuffer
        d_f, d_p1, d_p2 = DB.from_numpy(freq), DB.from_numpy(p1), DB.from_numpy(p2)
        d_ph, d_hd = DB(V), DB(V)
        d_o = DB((N, V), np.float64, zero=False)
        chk(L.mxg_osc_render(wf, V, N, d_f.ptr, 2, d_p1.ptr, d_p2.ptr, d_ph.ptr, d_hd.ptr, d_o.ptr, None), "mxg_osc_render fps=2")
        assert_bits_equal(d_o.numpy(), exp, "fps = 2, waveform %d" % wf)
        assert_bits_equal(d_ph.numpy(), ph, "phase")


@pytest.mark.parametrize("wf", [8, 3, 0, 10, 6])
@pytest.mark.parametrize("V,N,pad", [(4096, 301, 32), (4098, 64, 2), (4097, 33, 1), (70000, 40, 512), (2 * 98304 + 4098, 96, 34), (512, 2, 6)])
def test_row_pitch_same_bits(mx, wf, V, N, pad):
    """mxg_osc_render_pitch: the block written with a row pitch of V + pad doubles holds, in its first V columns, the bits of the
    unpadded render (pair-row, two-voices-per-lane and 8-byte store paths, odd pitches, the launch plan of large banks), the padding
    columns are never written, and the state afterwards is the same."""
    freq = 20.0 + np.arange(V) * (15000.0 / V)
    p1 = np.full(V, 0.3) if wf == 6 else None
    a = mx.maxiOscBank(V)
    ref = np.concatenate([a.render(wf, freq, N, p1=p1).numpy(), a.render(wf, freq, N, p1=p1).numpy()])
    b = mx.maxiOscBank(V)
    buf = mx.DeviceBuffer.from_numpy(np.full((N, V + pad), -7.0))
    got = []
    for _ in range(2):
        b.render(wf, freq, N, p1=p1, out=buf, pitch=V + pad)
        g = buf.numpy()
        assert (g[:, V:] == -7.0).all(), "padding columns touched"
        got.append(g[:, :V].copy())
    assert_bits_equal(np.concatenate(got), ref, "padded rows vs natural rows")
    assert_bits_equal(b.phase.numpy(), a.phase.numpy(), "phase")
    assert_bits_equal(b.output.numpy(), a.output.numpy(), "output")
    L = mx.lib()
    assert L.mxg_osc_render_pitch(wf, V, N, 1, 0, None, None, 1, 1, 1, (V - 1) * 8, None) < 0   # pitch below the bank
    assert L.mxg_osc_render_pitch(wf, V, N, 1, 0, None, None, 1, 1, 1, V * 8 + 4, None) < 0     # not a multiple of 8
