"""GPU parity (-m gpu): maxiDelayline and maxiSample banks through the C-ABI vs oracle + golden."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

SMP = ["play", "playOnce", "playLoop", "playUntil", "playAtSpeed", "playOnceAtSpeed",
       "playUntilAtSpeed", "play4", "playAtSpeedBetweenPoints"]


def test_delay_golden(mx, golden):
    g = golden("delay.npz")
    cap, V = int(g["cap"]), g["x"].shape[1]
    for mode, name in enumerate(["dl", "dlFromPosition"]):
        bank = mx.maxiDelaylineBank(V, cap)
        x1, x2 = mx.DeviceBuffer.from_numpy(g["x"][:250]), mx.DeviceBuffer.from_numpy(g["x"][250:])
        if mode == 0:
            o = np.concatenate([bank.dl(x1, g["size"], g["fb"]).numpy(), bank.dl(x2, g["size"], g["fb"]).numpy()])
        else:
            o = np.concatenate([bank.dlFromPosition(x1, g["size"], g["fb"], g["pos"]).numpy(),
                                bank.dlFromPosition(x2, g["size"], g["fb"], g["pos"]).numpy()])
        assert_bits_equal(o, g["out_" + name], name)
        assert_bits_equal(bank.memory.numpy(), g["mem_" + name], name + " mem")
        assert np.array_equal(bank.phase.numpy(), g["phase_" + name])


def test_delay_vs_oracle_large(mx, port):
    rng = np.random.default_rng(21)
    V, N, cap = 4096, 700, 300   # N > size: the ring wraps inside one launch
    x = rng.uniform(-1, 1, (N, V))
    size = np.where(np.arange(V) % 3 == 0, 257, rng.integers(1, cap + 1, V)).astype(np.int32)
    fb = rng.uniform(0, 0.9, V)
    bank = mx.maxiDelaylineBank(V, cap)
    dx = mx.DeviceBuffer.from_numpy(x)
    o = bank.dl(dx, size, fb).numpy()
    e, emem, eph = port.delay(0, x, size, fb, cap)
    assert_bits_equal(o, e, "dl")
    assert_bits_equal(bank.memory.numpy(), emem, "mem")
    assert np.array_equal(bank.phase.numpy(), eph)
    L = mx.lib()
    assert L.mxg_delay_render(2, V, N, dx.ptr, None, None, None, bank.memory.ptr, cap, bank.phase.ptr, dx.ptr, None) == -1


@pytest.mark.parametrize("mode", range(9))
def test_sample_golden(mx, golden, mode):
    g = golden("sample.npz")
    name = SMP[mode]
    N = int(g["N"])
    V = g["pos0_" + name].size
    bank = mx.maxiSampleBank(V)
    bank.setSample(g["samples"])
    assert bank.getLength() == g["samples"].size and bank.mySampleRate == 44100
    assert np.array_equal(bank.position.numpy(), np.full(V, g["samples"].size - 1.0))  # H:677
    bank.position.upload(g["pos0_" + name])
    kw = dict(a=g["a_" + name], start=g["start_" + name], end=g["end_" + name])
    o = np.concatenate([bank.render(mode, N // 2, **kw).numpy(), bank.render(mode, N - N // 2, **kw).numpy()])
    assert_bits_equal(o, g["out_" + name], name)
    assert_bits_equal(bank.position.numpy(), g["pos_" + name], name + " position")


def test_sample_speed_mod_sr96k(mx, golden):
    g = golden("sample.npz")
    V = g["speed_mod"].shape[1]
    mx.maxiSettings.setup(96000, 2, 1024)
    try:
        bank = mx.maxiSampleBank(V)
        bank.setSample(g["samples"])
        bank.trigger()
        o = bank.playAtSpeed(g["speed_mod"], int(g["N"]), per_sample=True).numpy()
        p = bank.position.numpy()
    finally:
        mx.maxiSettings.setup(44100, 2, 1024)
    assert_bits_equal(o, g["out_speed_mod_sr96k"])
    assert_bits_equal(p, g["pos_speed_mod_sr96k"])


def test_sample_vs_oracle_large(mx, port):
    rng = np.random.default_rng(31)
    V, N, Ls = 3000, 512, 100000
    smp = rng.uniform(-1, 1, Ls)
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    bank.setPosition(np.arange(V) / V)
    speed = 0.25 + 1.5 * (np.arange(V) % 97) / 96
    pos0 = bank.position.numpy()
    o = bank.playAtSpeed(speed, N).numpy()
    e, ep = port.sample(4, smp, N, pos0, a=speed)
    assert_bits_equal(o, e, "playAtSpeed")
    assert_bits_equal(bank.position.numpy(), ep)
    # integer-index play(): bit-exact and wraps
    bank.trigger()
    o = bank.play(N).numpy()
    e, _ = port.sample(0, smp, N, np.zeros(V))
    assert_bits_equal(o, e, "play")


@pytest.mark.parametrize("Ls,N", [(200000, 512), (5000, 512), (70000, 203)])
def test_play_with_heads_between_elements(mx, port, Ls, N):
    """maxiSample::play() (C:740-747) from heads that sit BETWEEN two elements -- what playAtSpeed leaves behind: the index is (long)pos
    and the head grows by 1.0 per sample, so a full wavefront of such heads that does not wrap inside the block takes the 16-byte row
    loads of the integer-head path (round 6), with the head after the block from the exact multi-step sum.  Heads just below powers of
    two (the fraction loses a bit at every binade crossing), fractions within 2^-20 of the next integer (must NOT take the path: an
    addition may round up to it), heads that wrap, three carried blocks."""
    rng = np.random.default_rng(Ls + N)
    V = 256
    smp = rng.uniform(-1, 1, Ls)
    span = max(Ls - 3 * N - 8, 16)
    pos0 = rng.uniform(0, span, V)
    pos0[0:64] = np.floor(pos0[0:64]) + rng.uniform(0.001, 0.999, 64)            # wave 0: plain fractional heads
    pw = 2.0 ** np.arange(1, 12)
    pos0[64:64 + pw.size] = pw - rng.uniform(0.01, 0.9, pw.size)                  # wave 1: binade crossings inside the block
    pos0[100] = 1000.0 + (1.0 - 2.0 ** -30)                                       # wave 1: a fraction that may round up: general path
    pos0[128:192] = np.floor(pos0[128:192])                                       # wave 2: integer heads (the old straight path)
    pos0[192:256] = Ls - rng.uniform(1, 2 * N, 64)                                # wave 3: wraps inside the blocks
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    bank.position.upload(pos0)
    o = np.concatenate([bank.render(0, N).numpy() for _ in range(3)])
    e, ep = port.sample(0, smp, 3 * N, pos0)
    assert_bits_equal(o, e, "play, heads between elements")
    assert_bits_equal(bank.position.numpy(), ep, "head after three blocks")


@pytest.mark.parametrize("V,N", [(65536, 100), (49152 + 70, 64), (131072, 40)])
def test_play_paced_schedule_same_bits(mx, port, V, N):
    """maxiSample::play() at the store-bound bank sizes runs its whole-chunk path on the paced schedule (csrc/mxg_pace.h: eight samples
    every P ticks of the 100 MHz counter, P from a controller in device scratch that the kernel updates): five carried blocks against
    the free-running kernel (knob smp_pace 1) and a fixed period, bit for bit; one wavefront of heads that wrap (general path beside
    paced ones), a subsample of voices against the oracle."""
    L = mx.lib()
    rng = np.random.default_rng(V)
    Ls = 30000
    smp = rng.uniform(-1, 1, Ls)
    pos0 = np.floor(rng.uniform(0, Ls - 5 * N - 16, V))
    pos0[64:128] = Ls - 1 - rng.integers(0, 3 * N, 64)       # a wavefront whose heads wrap inside the blocks

    def run(pace):
        prev = L.mxg_tune(b"smp_pace", pace)
        try:
            bank = mx.maxiSampleBank(V)
            bank.setSample(smp)
            bank.position.upload(pos0)
            o = np.concatenate([bank.render(0, N).numpy() for _ in range(5)])
            return o, bank.position.numpy().copy()
        finally:
            L.mxg_tune(b"smp_pace", prev)
    ref = run(1)
    for pace in (0, 40):
        got = run(pace)
        assert_bits_equal(got[0], ref[0], "play, smp_pace=%d" % pace)
        assert_bits_equal(got[1], ref[1], "heads, smp_pace=%d" % pace)
    sel = np.unique(np.concatenate([np.arange(0, V, 1009), np.arange(60, 132)]))
    e, ep = port.sample(0, smp, 5 * N, pos0[sel])
    assert_bits_equal(ref[0][:, sel], e, "play against the oracle")
    assert_bits_equal(ref[1][sel], ep, "heads against the oracle")


@pytest.mark.parametrize("mode", range(9))
@pytest.mark.parametrize("N", [1, 7, 8, 9, 16, 21, 203])
def test_sample_ragged_blocks(mx, golden, port, mode, N):
    """Block lengths around the pipelined chunk size (8, or 4 for play4): full chunks, odd chunk
    counts, the speculative run-ahead and the per-sample tail must all leave the same head."""
    g = golden("sample.npz")
    name = SMP[mode]
    V = g["pos0_" + name].size
    bank = mx.maxiSampleBank(V)
    bank.setSample(g["samples"])
    bank.position.upload(g["pos0_" + name])
    kw = dict(a=g["a_" + name], start=g["start_" + name], end=g["end_" + name])
    o = bank.render(mode, N, **kw).numpy()
    e, ep = port.sample(mode, g["samples"], N, g["pos0_" + name], **kw)
    assert_bits_equal(o, e, f"{name} N={N}")
    assert_bits_equal(bank.position.numpy(), ep, f"{name} N={N} position")


@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("N", [5, 12, 24, 203])
def test_sample_per_sample_speed_ragged(mx, golden, port, mode, N):
    g = golden("sample.npz")
    name = SMP[mode]
    V = g["pos0_" + name].size
    rng = np.random.default_rng(100 + mode)
    a = g["a_" + name][None, :] * rng.uniform(0.5, 1.5, (N, V))
    if mode >= 7:
        a *= np.where(rng.uniform(size=(N, V)) < 0.2, -1.0, 1.0)   # direction flips
    bank = mx.maxiSampleBank(V)
    bank.setSample(g["samples"])
    bank.position.upload(g["pos0_" + name])
    kw = dict(start=g["start_" + name], end=g["end_" + name])
    o = bank.render(mode, N, a=a, per_sample=True, **kw).numpy()
    e, ep = port.sample(mode, g["samples"], N, g["pos0_" + name], a=a, aps=True, **kw)
    assert_bits_equal(o, e, f"{name} N={N}")
    assert_bits_equal(bank.position.numpy(), ep, f"{name} N={N} position")


@pytest.mark.parametrize("N", [3, 8, 24, 700, 1001])
@pytest.mark.parametrize("sizes", ["uniform", "ge16", "mixed"])
@pytest.mark.parametrize("rw,V", [(0, 1024), (2, 1024), (3, 1000), (4, 130), (3, 1023)])
def test_delay_pipelined_paths(mx, port, N, sizes, rw, V):
    """dl(): wavefronts whose lines all have size >= 16 take the pipelined path, the rest the plain
    loop; both must give the reference's ring, phase and output, across two carried blocks.  rw: knob rw_store -- the whole chunks
    of the pipelined path with 16-byte pair-row input / output streams (even banks; an odd bank keeps the 8-byte streams)."""
    rng = np.random.default_rng(N)
    cap = 300
    L = mx.lib()
    prev = L.mxg_tune(b"rw_store", rw)
    try:
        _delay_pipelined_case(mx, port, rng, N, sizes, V, cap)
    finally:
        L.mxg_tune(b"rw_store", prev)


def _delay_pipelined_case(mx, port, rng, N, sizes, V, cap):
    if sizes == "uniform":
        size = np.full(V, 257, np.int32)
    elif sizes == "ge16":
        size = rng.integers(16, cap + 1, V).astype(np.int32)
    else:
        size = rng.integers(1, cap + 1, V).astype(np.int32)
        size[:min(256, V)] = rng.integers(16, 40, min(256, V))      # whole wavefronts on the pipelined path
    x = rng.uniform(-1, 1, (2 * N, V))
    fb = rng.uniform(0, 0.9, V)
    bank = mx.maxiDelaylineBank(V, cap)
    o = np.concatenate([bank.dl(mx.DeviceBuffer.from_numpy(x[:N]), size, fb).numpy(),
                        bank.dl(mx.DeviceBuffer.from_numpy(x[N:]), size, fb).numpy()])
    e, emem, eph = port.delay(0, x, size, fb, cap)
    assert_bits_equal(o, e, "dl")
    assert_bits_equal(bank.memory.numpy(), emem, "mem")
    assert np.array_equal(bank.phase.numpy(), eph)


ZX = ["playOnZX", "playOnZXAtSpeed", "playOnZXAtSpeedFromOffset", "playOnZXAtSpeedBetweenPoints", "loopSetPosOnZX"]


def _zx_case(V, N, seed):
    rng = np.random.default_rng(seed)
    smp = rng.uniform(-1, 1, 1500)
    trig = np.sin(np.arange(N)[:, None] * rng.uniform(0.02, 0.3, V)[None, :] + rng.uniform(0, 6, V))
    trig[:, ::5] = np.where(rng.uniform(size=(N, (V + 4) // 5)) < 0.05, 1.0, 0.0)   # sparse impulses, exact zeros
    return dict(smp=smp, trig=trig, a=rng.uniform(0.3, 2.5, V), p0=rng.uniform(0, 0.6, V),
                p1=rng.uniform(0.1, 0.5, V), pos0=rng.uniform(0, 1499, V))


@pytest.mark.parametrize("mode", range(5))
@pytest.mark.parametrize("N", [5, 64, 203])
def test_sample_on_zx(mx, port, mode, N):
    """playOnZX* / loopSetPosOnZX (C:1006-1042): two carried blocks; output, position and the
    maxiTrigger state must match the reference bit for bit."""
    V = 200
    c = _zx_case(V, 2 * N, 50 + mode)
    bank = mx.maxiSampleBank(V)
    bank.setSample(c["smp"])
    bank.position.upload(c["pos0"])
    name = ZX[mode]
    kw = dict(a=c["a"] if mode in (1, 2, 3) else None, p0=c["p0"] if mode >= 2 else None,
              p1=c["p1"] if mode == 3 else None)
    o = np.concatenate([bank.render_trig(name, c["trig"][:N], **kw).numpy(),
                        bank.render_trig(name, c["trig"][N:], **kw).numpy()])
    e, ep, ezp, ezf = port.sample_zx(mode, c["smp"], c["trig"], c["pos0"], a=c["a"], p0=c["p0"], p1=c["p1"])
    assert_bits_equal(o, e, name)
    assert_bits_equal(bank.position.numpy(), ep, name + " position")
    assert_bits_equal(bank.zx_prev.numpy(), ezp, name + " previousValue")
    assert np.array_equal(bank.zx_first.numpy(), ezf)
    assert np.abs(e).max() > 0.1


def test_sample_on_zx_per_sample_speed(mx, port):
    V, N = 130, 100
    c = _zx_case(V, N, 77)
    sp = c["a"][None, :] * np.random.default_rng(3).uniform(0.5, 1.5, (N, V))
    bank = mx.maxiSampleBank(V)
    bank.setSample(c["smp"])
    bank.position.upload(c["pos0"])
    o = bank.render_trig("playOnZXAtSpeedBetweenPoints", c["trig"], a=sp, p0=c["p0"], p1=c["p1"], per_sample=True).numpy()
    e, ep, _, _ = port.sample_zx(3, c["smp"], c["trig"], c["pos0"], a=sp, aps=True, p0=c["p0"], p1=c["p1"])
    assert_bits_equal(o, e)
    assert_bits_equal(bank.position.numpy(), ep)


@pytest.mark.parametrize("N", [3, 64, 301])
def test_sample_play_with_phasor(mx, port, N):
    """playWithPhasor (C:753-816): forward and backward ramps, stalls (pos1 == pos2), out-of-range
    phasors (clamped), the size_t wrap of pos1-- at 0; two carried blocks."""
    V = 150
    rng = np.random.default_rng(N)
    smp = rng.uniform(-1, 1, 1500)
    n = np.arange(2 * N)[:, None]
    pha = (n * rng.uniform(0.0003, 0.01, V)[None, :] + rng.uniform(0, 1, V)) % 1.0
    pha[:, 1::4] = 1.0 - pha[:, 1::4]                 # backward
    pha[:, 2::4] = np.round(pha[:, 2::4] * 8) / 8     # staircase: repeated values
    pha[N // 2: N // 2 + 3, :] = rng.uniform(-0.3, 1.3, (min(3, 2 * N - N // 2), V))[: pha[N // 2: N // 2 + 3].shape[0]]
    pha[:, 3] = 0.0                                   # stuck at 0: pos >= prev -> pos2++
    pha[1:, 7] = np.linspace(0.001, 0.0, 2 * N - 1)   # creeping down to 0: pos1-- path
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    o = np.concatenate([bank.playWithPhasor(pha[:N]).numpy(), bank.playWithPhasor(pha[N:]).numpy()])
    e, epp, epf = port.sample_phasor(smp, pha)
    assert_bits_equal(o, e, "playWithPhasor")
    assert_bits_equal(bank.phasor_prev.numpy(), epp)
    assert np.array_equal(bank.phasor_first.numpy(), epf)


def test_players_outside_the_defined_range_stay_inside_the_buffer(mx):
    """play4 with a step much longer than its loop, playLoop with end > 1, heads uploaded far outside the sample: the
    reference indexes outside its vector there (undefined).  The device's gathers are clamped to the uploaded buffer and
    its guard elements, so such a patch reads zeros instead of memory beyond the allocation: finite, bounded output."""
    rng = np.random.default_rng(8)
    V, N = 4096, 300
    for Ls in (509, 4093):                    # (len + 3) * 8 bytes ends right on a 4 KiB page
        smp = rng.uniform(-1, 1, Ls)
        bank = mx.maxiSampleBank(V)
        bank.setSample(smp)
        bank.position.upload(rng.uniform(-50.0 * Ls, 50.0 * Ls, V))
        freq = rng.uniform(500, 20000, V) * np.where(np.arange(V) % 2, -1.0, 1.0)
        o = bank.render(7, N, a=freq, start=np.full(V, 2.0), end=np.full(V, Ls - 1.0)).numpy()
        assert np.isfinite(o).all() and np.abs(o).max() < 8.0
        bank.position.upload(rng.uniform(0, Ls, V))
        o = bank.render(2, N, start=np.zeros(V), end=np.full(V, 40.0)).numpy()
        assert np.isfinite(o).all() and np.abs(o).max() <= 1.0
        o = bank.render(4, N, a=rng.uniform(-3000, 3000, V)).numpy()
        assert np.isfinite(o).all() and np.abs(o).max() <= 3.0


def test_delay_sizes_and_phases_outside_the_ring(mx, port):
    """size <= 0 is defined in the reference (`phase >= size` resets the phase on every sample: slot 0 only) and must
    match the oracle.  A size beyond the bank's capacity, or an uploaded negative / huge phase or position, would index
    outside the ring: the kernel holds size to the capacity and restarts such a phase at slot 0 -- outputs stay finite,
    phases inside the ring, and the voices next to them are untouched."""
    rng = np.random.default_rng(33)
    V, N, cap = 512, 200, 64
    x = rng.uniform(-1, 1, (N, V))
    fb = rng.uniform(0, 0.9, V)
    size = rng.integers(1, cap + 1, V).astype(np.int32)
    size[::8] = 0
    size[1::8] = -5
    bank = mx.maxiDelaylineBank(V, cap)
    o = bank.dl(mx.DeviceBuffer.from_numpy(x), size, fb).numpy()
    # `phase >= size` (C:421) is always true for size <= 0, so a negative size behaves like 0 (the port takes 0 .. cap)
    e, emem, eph = port.delay(0, x, np.maximum(size, 0), fb, cap)
    assert_bits_equal(o, e, "dl with size <= 0")
    assert_bits_equal(bank.memory.numpy(), emem, "mem")
    # undefined in the reference: must stay inside the ring
    wild = rng.integers(1, cap + 1, V).astype(np.int32)
    wild[::4] = cap + 1000
    wild[1::4] = 2 ** 31 - 1
    bank = mx.maxiDelaylineBank(V, cap)
    ph = np.zeros(V, np.int32)
    ph[::3] = -7
    ph[1::3] = 2 ** 30
    bank.phase.upload(ph)
    pos = rng.integers(-(2 ** 31), 2 ** 31 - 1, V).astype(np.int32)
    for mode_call in (lambda: bank.dl(mx.DeviceBuffer.from_numpy(x), wild, fb),
                      lambda: bank.dlFromPosition(mx.DeviceBuffer.from_numpy(x), wild, fb, pos)):
        o = mode_call().numpy()
        assert np.isfinite(o).all()
        p = bank.phase.numpy()
        assert (p >= 0).all() and (p <= cap).all()
    sane = (np.arange(V) % 4 >= 2) & (np.arange(V) % 3 == 2)           # voices with an in-range size and phase 0
    e, _, _ = port.delay(0, np.ascontiguousarray(x[:, sane]), wild[sane], fb[sane], cap)
    bank2 = mx.maxiDelaylineBank(V, cap); bank2.phase.upload(ph)
    o = bank2.dl(mx.DeviceBuffer.from_numpy(x), wild, fb).numpy()
    assert_bits_equal(o[:, sane], e, "in-range voices beside the wild ones")


def test_play_at_speed_between_points_from_pos(mx, port):
    """maxiSample::playAtSpeedBetweenPointsFromPos (C:826-880) with a caller-supplied position signal [N][V]: forward
    and backward frequencies, positions before `start`, beyond `end` and on the last samples of the buffer."""
    rng = np.random.default_rng(828)
    Ls, V, N = 3000, 37, 50
    smp = rng.uniform(-1, 1, Ls)
    freq = np.where(rng.uniform(size=V) < 0.5, 1, -1) * rng.uniform(0.5, 40, V)
    start, end = rng.uniform(0, 1000, V), rng.uniform(1500, Ls + 50, V)
    pos = rng.uniform(-10, Ls + 5, (N, V))
    pos[0, :5] = [0, Ls - 1, Ls - 2, Ls, 1]
    sb = mx.maxiSampleBank(V)
    sb.setSample(smp)
    L = mx.lib()
    d = [mx.DeviceBuffer.from_numpy(a) for a in (freq, start, end, pos)]
    out = mx.DeviceBuffer((N, V), zero=False)
    assert L.mxg_sample_render_frompos(V, N, sb.d_samples, Ls, d[0].ptr, 0, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr, None) == 0
    exp = np.stack([port.sample(8, smp, 1, pos[n], a=freq, start=start, end=end)[0][0] for n in range(N)])
    assert_bits_equal(out.numpy(), exp, "playAtSpeedBetweenPointsFromPos")
    # per-sample frequencies
    fm = freq[None, :] * rng.uniform(0.5, 2.0, (N, V))
    dfm = mx.DeviceBuffer.from_numpy(fm)
    assert L.mxg_sample_render_frompos(V, N, sb.d_samples, Ls, dfm.ptr, 1, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr, None) == 0
    exp = np.stack([port.sample(8, smp, 1, pos[n], a=fm[n], start=start, end=end)[0][0] for n in range(N)])
    assert_bits_equal(out.numpy(), exp, "playAtSpeedBetweenPointsFromPos, per-sample frequency")
    assert L.mxg_sample_render_frompos(V, N, sb.d_samples, Ls, None, 0, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr, None) == -1


def _mixed_speeds(V, rng, reverse):
    """Waves 0-1 and 3: slow voices; wave 2: a few fast ones; with `reverse` every third voice runs backwards."""
    sp = rng.uniform(0.2, 1.7, V)
    sp[128:192:7] = rng.uniform(2.0, 3.5, sp[128:192:7].size)
    if reverse:
        sp[1::3] *= -1.0
    return sp


@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("Ls,split", [(20000, 0), (700, 0), (700, 1), (700, 3), (700, 2), (20000, 8), (20000, 5), (700, -3), (20000, -6)])
def test_sample_speed_players_full_waves(mx, port, mode, Ls, split):
    """The interpolating players over full wavefronts, three carried blocks of ragged lengths, long and short (many wraps)
    buffers -- and, for playAtSpeed / playOnceAtSpeed / playUntilAtSpeed, every setting of the time-part knob: a part skips to
    its first sample with the exact multi-step head advance, so the bits (and the head left behind) cannot depend on it.
    (A negative split: that many parts without the software pipeline, smp_pipe 0.)"""
    pipe = 0 if split < 0 else 1
    split = abs(split)
    rng = np.random.default_rng(900 + mode + Ls)
    V = 256
    smp = rng.uniform(-1, 1, Ls)
    speed = _mixed_speeds(V, rng, reverse=mode >= 7)   # head step per output sample
    if mode >= 7:
        # play4 / playAtSpeedBetweenPoints(frequency, start, end): start / end in samples, step = (end-start)*f/sr (C:826-940)
        start = np.floor(rng.uniform(2, Ls * 0.3, V)); end = np.floor(start + rng.uniform(Ls * 0.2, Ls * 0.6, V))
        start[192:] = np.floor(rng.uniform(2, Ls - 90, 64)); end[192:] = start[192:] + np.floor(rng.uniform(20, 60, 64))
        a = speed * 44100.0 / (end - start)
        pos0 = start + np.floor(rng.uniform(0, 1, V) * (end - start - 1))
    else:
        a = speed
        start = np.zeros(V)
        end = rng.uniform(0.3, 1.0, V) if mode == 6 else np.ones(V)     # playUntilAtSpeed: end as a fraction
        pos0 = rng.uniform(0, Ls - 2, V)
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    bank.position.upload(pos0)
    kw = dict(a=a, start=start, end=end)
    blocks = [203, 64, 9]
    prev = mx.lib().mxg_tune(b"smp_split", split)
    prev_pipe = mx.lib().mxg_tune(b"smp_pipe", pipe)
    try:
        o = np.concatenate([bank.render(mode, n, **kw).numpy() for n in blocks])
    finally:
        mx.lib().mxg_tune(b"smp_split", prev)
        mx.lib().mxg_tune(b"smp_pipe", prev_pipe)
    e, ep = port.sample(mode, smp, sum(blocks), pos0, **kw)
    assert_bits_equal(o, e, "mode %d" % mode)
    assert_bits_equal(bank.position.numpy(), ep, "position")
    assert np.abs(e).max() > 0.1


@pytest.mark.parametrize("mode", [4, 5, 6])
@pytest.mark.parametrize("Ls,split,slow", [(20000, 0, False), (700, 3, False), (20000, 5, True), (700, -3, True), (300000, 2, False)])
def test_sample_speed_players_ring_rows(mx, port, mode, Ls, split, slow):
    """Round 6: the time-part kernel's window rows used as RINGS (knob smp_ring 2; automatic for samples beyond 1 GiB) -- a smooth chunk
    fetches only the 16-byte pieces behind the last one its voice holds.  Same bits as the oracle over three carried blocks: mixed speeds
    (a wavefront with fast voices falls back to windows or gathers chunk by chunk and forgets its ring), very slow heads (chunks that need
    nothing new), short samples (wraps inside a block), every part count, with and without the software pipeline."""
    pipe = 0 if split < 0 else 1
    split = abs(split)
    rng = np.random.default_rng(1900 + mode + Ls)
    V = 256
    smp = rng.uniform(-1, 1, Ls)
    a = _mixed_speeds(V, rng, reverse=False)
    if slow:
        a[:64] = rng.uniform(0.001, 0.05, 64)      # wave 0: heads that stay inside one piece for many chunks
        a[64:128] = rng.uniform(0.9, 1.1, 64)
    end = rng.uniform(0.3, 1.0, V) if mode == 6 else np.ones(V)
    pos0 = rng.uniform(0, Ls - 2, V)
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    bank.position.upload(pos0)
    kw = dict(a=a, start=np.zeros(V), end=end)
    blocks = [512, 203, 64]
    L = mx.lib()
    prev = [L.mxg_tune(b"smp_split", split), L.mxg_tune(b"smp_pipe", pipe), L.mxg_tune(b"smp_ring", 2)]
    try:
        o = np.concatenate([bank.render(mode, n, **kw).numpy() for n in blocks])
    finally:
        L.mxg_tune(b"smp_split", prev[0]); L.mxg_tune(b"smp_pipe", prev[1]); L.mxg_tune(b"smp_ring", prev[2])
    e, ep = port.sample(mode, smp, sum(blocks), pos0, **kw)
    assert_bits_equal(o, e, "mode %d, rows as rings" % mode)
    assert_bits_equal(bank.position.numpy(), ep, "position")


@pytest.mark.parametrize("mode", [4, 5, 6])
def test_sample_time_parts_corner_heads(mx, port, mode):
    """Heads and speeds the skip must not touch or must get exactly right: negative and zero speeds and negative heads in the
    wavefront (part 0 renders it whole), a step below half an ulp of the head (the head does not move), a head sitting
    exactly on len, steps longer than the buffer, a huge playOnce head."""
    Ls, N = 1000, 512
    rng = np.random.default_rng(77 + mode)
    smp = rng.uniform(-1, 1, Ls)
    V = 192
    a = rng.uniform(0.3, 1.5, V); pos0 = rng.uniform(0, Ls - 1, V)
    a[64:128:5] = [-0.7, 0.0, -2.5, 1e-300, -1e-3, 0.9, 1.1, -0.4, 0.0, 3.0, -1.0, 0.5, -0.2][:len(a[64:128:5])]  # wave 1: mixed signs
    if mode == 6:
        a = np.abs(a)                        # playUntilAtSpeed only tests the upper bound: a head below 0 is the reference's UB
    else:
        pos0[70] = -3.5
    a[130] = 1e-14; pos0[130] = 800.0        # below half an ulp of 800: stuck
    pos0[131] = float(Ls)                    # exactly on the wrap limit
    a[132] = 2500.0                          # longer than the buffer
    a[133] = 999.5; pos0[133] = 0.25
    if mode == 5:
        pos0[134] = 1e15; a[134] = 0.75       # far beyond the sample: silence, the head keeps adding
    end = rng.uniform(0.4, 1.0, V)
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    bank.position.upload(pos0)
    kw = dict(a=a, start=np.zeros(V), end=end)
    o = np.concatenate([bank.render(mode, N, **kw).numpy(), bank.render(mode, 100, **kw).numpy()])
    e, ep = port.sample(mode, smp, N + 100, pos0, **kw)
    assert_bits_equal(o, e, "mode %d" % mode)
    assert_bits_equal(bank.position.numpy(), ep, "position")


@pytest.mark.parametrize("mode", [4, 5])
def test_sample_time_parts_many_short_launches(mx, port, mode):
    """The head is read by every time part and stored by one: with 40 samples per part the storing part is done in no time,
    and nothing but the kernel's own part_signal / part_wait keeps another part from starting on the new head.  300 launches of
    a bank wide enough that its workgroups do not all start together."""
    Ls, V, N, reps = 3000, 8192, 160, 300
    rng = np.random.default_rng(5 + mode)
    smp = rng.uniform(-1, 1, Ls)
    a = rng.uniform(0.3, 1.6, V)
    pos0 = rng.uniform(0, Ls - 2, V) if mode == 4 else rng.uniform(0, 8, V)
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    bank.position.upload(pos0)
    da = mx.DeviceBuffer.from_numpy(a)
    ring = [mx.DeviceBuffer((N, V), zero=False) for _ in range(3)]
    prev = mx.lib().mxg_tune(b"smp_split", 4)
    try:
        for r in range(reps):                                     # enqueued back to back, nothing in between
            bank.render(mode, N, a=da, out=ring[r % 3])
        o = np.concatenate([ring[r % 3].numpy() for r in range(reps - 3, reps)])
    finally:
        mx.lib().mxg_tune(b"smp_split", prev)
    e, ep = port.sample(mode, smp, N * reps, pos0, a=a)
    assert_bits_equal(o, e[-3 * N:], "mode %d" % mode)
    assert_bits_equal(bank.position.numpy(), ep, "position")


def test_sample_zx_and_phasor_slow_full_waves(mx, port):
    """Trigger-driven players and playWithPhasor on full wavefronts of slow voices, retriggers included."""
    V, N = 192, 300
    c = _zx_case(V, N, 333)
    c["a"] = np.random.default_rng(4).uniform(0.3, 1.7, V)
    for mode in (1, 2, 3):
        bank = mx.maxiSampleBank(V)
        bank.setSample(c["smp"])
        bank.position.upload(c["pos0"])
        kw = dict(a=c["a"], p0=c["p0"] if mode >= 2 else None, p1=c["p1"] if mode == 3 else None)
        o = bank.render_trig(ZX[mode], c["trig"], **kw).numpy()
        e, ep, _, _ = port.sample_zx(mode, c["smp"], c["trig"], c["pos0"], a=c["a"], p0=c["p0"], p1=c["p1"])
        assert_bits_equal(o, e, ZX[mode])
        assert_bits_equal(bank.position.numpy(), ep, ZX[mode] + " position")
    rng = np.random.default_rng(8)
    smp = rng.uniform(-1, 1, 4000)
    pha = (np.arange(N)[:, None] * rng.uniform(0.00005, 0.0004, V)[None, :] + rng.uniform(0, 1, V)) % 1.0
    pha[:, 1::2] = 1.0 - pha[:, 1::2]
    bank = mx.maxiSampleBank(V)
    bank.setSample(smp)
    o = bank.playWithPhasor(pha).numpy()
    e, _, _ = port.sample_phasor(smp, pha)
    assert_bits_equal(o, e, "playWithPhasor, slow ramps")
