"""GPU parity (-m gpu): maxiTimeStretch / maxiStretch banks through the C-ABI vs oracle + golden."""
import numpy as np
import pytest

from conftest import assert_bits_equal, mix_tol

pytestmark = pytest.mark.gpu

CASES = {"ts_hann": (0, 0, 0.05, 4, True), "ts_hamming_norand": (0, 1, 0.03, 3, False),
         "st_hann": (1, 0, 0.05, 2, True), "st_gauss": (1, 8, 0.021, 5, True)}


def _sample_bank(mx, samples):
    sb = mx.maxiSampleBank(1)
    sb.setSample(samples)
    return sb


def make_bank(mx, mode, window, samples, S):
    sb = mx.maxiSampleBank(1)
    sb.setSample(samples)
    bank = (mx.maxiTimeStretchBank if mode == 0 else mx.maxiStretchBank)(S, sb, window)
    return bank


@pytest.fixture(params=[(1, 1), (1, 0), (0, 1)], ids=["tiles", "walk", "serial"])
def chunked(mx, request):
    """Run with the scheduler pre-pass + tile renders (K8a + K8c / K8d, the default), with the (stream, chunk) walk K8b in
    place of K8d, and with the serial K8."""
    prev = mx.lib().mxg_tune(b"grain_chunked", request.param[0])
    prev2 = mx.lib().mxg_tune(b"grain_line", request.param[1])
    yield request.param[0]
    mx.lib().mxg_tune(b"grain_chunked", prev)
    mx.lib().mxg_tune(b"grain_line", prev2)


@pytest.mark.parametrize("name", list(CASES))
def test_granular_golden(mx, golden, name, chunked):
    g = golden("grains.npz")
    mode, w, gl, ov, use_rnd = CASES[name]
    T = int(g["T"])
    S = g["speed"].size
    bank = make_bank(mx, mode, w, g["samples"], S)
    bank.setPosition(np.arange(S) / S)
    assert_bits_equal(bank.state.numpy(), g["st0"], "setPosition")
    h = T // 2
    rnd = g["rnd"] if use_rnd else None
    if mode == 0:
        o1 = bank.play(g["speed"], gl, ov, h, rnd=rnd).numpy()
        o2 = bank.play(g["speed"], gl, ov, T - h, rnd=rnd).numpy()
    else:
        o1 = bank.play(g["speed"], g["timestretch"], gl, ov, h, rnd=rnd).numpy()
        o2 = bank.play(g["speed"], g["timestretch"], gl, ov, T - h, rnd=rnd).numpy()
    assert_bits_equal(np.concatenate([o1, o2]), g["out_" + name], name)
    assert_bits_equal(bank.state.numpy(), g["st_" + name], name + " state")
    assert_bits_equal(bank.grains.numpy(), g["gst_" + name], name + " grains")


def test_granular_vs_oracle_many_streams(mx, port, chunked):
    """Config-5-shaped bank at reduced size: 1000 streams, 0.05 s hann grains, 4 overlaps."""
    rng = np.random.default_rng(51)
    Ls = 441000
    n = np.arange(Ls)
    smp = 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) \
        + 0.05 * rng.uniform(-1, 1, Ls)
    S, T = 1000, 3000
    speed = 0.25 + 1.5 * (np.arange(S) % 97) / 96
    bank = make_bank(mx, 0, "hann", smp, S)
    bank.setPosition(np.arange(S) / S)
    st0 = bank.state.numpy()
    o = bank.play(speed, 0.05, 4, T).numpy()
    e, est, egst, rc = port.granular(0, 0, smp, T, speed, grainLength=0.05, overlaps=4, st=st0)
    assert rc == 0
    assert_bits_equal(o, e, "timestretch bank")
    assert_bits_equal(bank.state.numpy(), est)
    assert_bits_equal(bank.grains.numpy(), egst)
    # stereo mixdown of the streams (pan x_s = s/(S-1)), within the mix tolerance
    pan = np.arange(S) / (S - 1.0)
    m = mx.maxiMixBank(S).stereo(mx.DeviceBuffer.from_numpy(o), pan).numpy()
    em = port.mix_stereo(e, pan)
    assert np.abs(m - em).max() <= mix_tol(S, np.abs(e).max())


def test_granular_errors(mx, chunked):
    rng = np.random.default_rng(3)
    smp = rng.uniform(-1, 1, 5000)
    bank = make_bank(mx, 0, "hann", smp, 4)
    # errors a kernel detects are deferred (mxg_last_async_error, include/maxigpu.h): the render call only enqueues; the
    # failure is the status of the next synchronising call -- here the download of the block
    with pytest.raises(mx.MaxiGpuError, match="more than 8 grains"):
        bank.play(1.0, 0.2, 12, 8000).numpy()       # > 8 live grains
    assert mx.lib().mxg_last_async_error() == 0     # reported once
    bank = make_bank(mx, 0, "hann", smp, 4)
    with pytest.raises(mx.MaxiGpuError, match="d_rnd exhausted"):
        bank.play(1.0, 0.05, 4, 4000, rnd=np.zeros((4, 2), np.int32)).numpy()  # rand queue exhausted
    prev = mx.lib().mxg_tune(b"grain_sync", 1)      # the synchronous form returns it from the call itself
    try:
        bank = make_bank(mx, 0, "hann", smp, 4)
        with pytest.raises(mx.MaxiGpuError, match="more than 8 grains"):
            bank.play(1.0, 0.2, 12, 8000)
    finally:
        mx.lib().mxg_tune(b"grain_sync", prev)
    with pytest.raises(ValueError):
        make_bank(mx, 0, "hann", smp, 4).play(1.0, 0.7, 4, 10)         # grain > 500 ms


@pytest.mark.parametrize("mode", [0, 1])
def test_fast_scheduler_random_rates(mx, port, mode):
    """The event-driven scheduler (exact multi-step additions) against the oracle's step-by-step
    loop for rates with full 53-bit mantissas, tiny/huge rates, negative rates (sequential
    fallback), several sample lengths (many wraps) and random jitter -- bit-exact state."""
    rng = np.random.default_rng(77 + mode)
    S, T = 192, 6000
    for Ls in (3000, 44100):
        smp = rng.uniform(-1, 1, Ls)
        a = rng.uniform(0.01, 3.0, S)
        a[:8] = [1.0, 0.265625, 2.0 ** -20, 7.3, -0.8, 1e-9, 0.1, 0.30000000000000004]
        b = rng.uniform(0.05, 4.0, S)
        b[:6] = [1.0, 0.5, 2.0 ** -30, 13.7, -1.5, 1.0 / 3.0]
        rnd = rng.integers(0, 10, (S, 64)).astype(np.int32)
        pm = rng.uniform(-0.05, 0.05, S)
        bank = make_bank(mx, mode, "hann", smp, S)
        bank.setPosition(rng.uniform(0, 1, S))
        st0 = bank.state.numpy()
        if mode == 0:
            o = bank.play(a, 0.05, 4, T, posMod=pm, rnd=rnd).numpy()
        else:
            o = bank.play(np.abs(a) + 0.05, b, 0.05, 3, T, posMod=pm, rnd=rnd).numpy()
        aa = a if mode == 0 else np.abs(a) + 0.05
        e, est, egst, rc = port.granular(mode, 0, smp, T, aa, b=b, posMod=pm, rnd=rnd, grainLength=0.05,
                                         overlaps=4 if mode == 0 else 3, st=st0)
        assert rc == 0
        assert_bits_equal(bank.state.numpy(), est, "scheduler state (Ls=%d)" % Ls)
        assert_bits_equal(bank.grains.numpy(), egst, "grains")
        assert_bits_equal(o, e, "output")


@pytest.fixture(params=[1, 0], ids=["eventdriven", "stepwise"])
def fast_sched(mx, request):
    prev = mx.lib().mxg_tune(b"grain_fast_sched", request.param)
    yield request.param
    mx.lib().mxg_tune(b"grain_fast_sched", prev)


@pytest.mark.parametrize("overlaps,window", [(2, "hann"), (4, "cosine"), (3, "gaussian")])
def test_play_at_position(mx, port, chunked, fast_sched, overlaps, window):
    """maxiTimeStretch::playAtPosition (L/maxiGrains.h:359-367): caller-driven per-sample position,
    spawn when floor(fmod(looper, cycle)) == 0; two carried blocks; out, scheduler and grains bit-exact."""
    rng = np.random.default_rng(60 + overlaps)
    L, S, T = 30000, 70, 4000
    smp = rng.uniform(-1, 1, L)
    pos = ((np.arange(2 * T)[:, None] * rng.uniform(0.2, 2.0, S)[None, :] / L) + rng.uniform(0, 1, S)) % 1.0
    pos[:, 3] = rng.uniform(-0.2, 1.2, 2 * T)          # clamped to [0, 1]
    pos[:, 5] = 1.0                                    # grains born at the very end: endPos clamp + wrap
    bank = make_bank(mx, 0, window, smp, S)
    o = np.concatenate([bank.playAtPosition(pos[:T], 0.05, overlaps).numpy(),
                        bank.playAtPosition(pos[T:], 0.05, overlaps).numpy()])
    w = mx.WINDOWS[window]
    e1, st, gst, rc = port.granular(2, w, smp, T, pos[:T], grainLength=0.05, overlaps=overlaps)
    e2, st, gst, rc2 = port.granular(2, w, smp, T, pos[T:], grainLength=0.05, overlaps=overlaps, st=st, gst=gst)
    assert rc == 0 and rc2 == 0
    assert_bits_equal(o, np.concatenate([e1, e2]), "playAtPosition")
    assert_bits_equal(bank.state.numpy(), st, "state")
    assert_bits_equal(bank.grains.numpy(), gst, "grains")
    assert np.abs(e2).max() > 0.3


@pytest.mark.parametrize("overlaps,gl", [(2, 0.05), (4, 0.05), (3, 0.031)])
def test_pitch_shift(mx, port, chunked, fast_sched, overlaps, gl):
    """maxiPitchShift::play (L/maxiGrains.h:412-430): grains with arbitrary (also negative, zero-ish)
    increments, speed - (cycleMod/cycleLength)*0.1 per grain; two carried blocks, bit-exact."""
    rng = np.random.default_rng(70 + overlaps)
    L, S, T = 30000, 90, 4000
    smp = rng.uniform(-1, 1, L)
    speed = rng.uniform(-2.0, 2.5, S)
    speed[:4] = [1.0, 0.5, -1.0, 2.0]
    pm = rng.uniform(-0.1, 0.1, S)
    sb = mx.maxiSampleBank(1)
    sb.setSample(smp)
    bank = mx.maxiPitchShiftBank(S, sb, "hann")
    st0 = np.zeros((4, S))
    st0[0] = rng.uniform(0, L, S)
    st0[0, 7] = L - 3.0                                # position wraps to 0 inside the block (:415)
    bank.state.upload(st0)
    o = np.concatenate([bank.play(speed, gl, overlaps, T, posMod=pm).numpy(),
                        bank.play(speed, gl, overlaps, T, posMod=pm).numpy()])
    e1, st, gst, rc = port.granular(3, 0, smp, T, speed, posMod=pm, grainLength=gl, overlaps=overlaps, st=st0)
    e2, st, gst, rc2 = port.granular(3, 0, smp, T, speed, posMod=pm, grainLength=gl, overlaps=overlaps, st=st, gst=gst)
    assert rc == 0 and rc2 == 0
    assert_bits_equal(o, np.concatenate([e1, e2]), "pitchshift")
    assert_bits_equal(bank.state.numpy(), st, "state")
    assert_bits_equal(bank.grains.numpy(), gst, "grains")
    assert np.abs(e2).max() > 0.3


@pytest.mark.parametrize("gl,overlaps", [(0.05, 4), (0.0123, 3), (0.05, 7), (0.002, 2)])
def test_play_at_position_event_driven_cycles(mx, port, gl, overlaps):
    """The birth-to-birth scheduler of playAtPosition for cycle lengths that are integers (0.05 s / 7 is not,
    0.05*44100/4 = 551.25 is not, 0.002*44100/2 = 44.1 ...), long runs and a looper that starts far from 0."""
    rng = np.random.default_rng(int(gl * 1e5) + overlaps)
    L, S, T = 60000, 40, 12000
    smp = rng.uniform(-1, 1, L)
    pos = rng.uniform(0, 1, (T, S))
    sb = mx.maxiSampleBank(1)
    sb.setSample(smp)
    bank = mx.maxiTimeStretchBank(S, sb, "hann")
    st0 = np.zeros((4, S))
    st0[1] = rng.integers(0, 100000, S).astype(np.float64)        # looper
    st0[1, 3] = 7.5                                                # not an integer: the stepwise path must take over
    bank.state.upload(st0)
    o = bank.playAtPosition(pos, gl, overlaps).numpy()
    e, st, gst, rc = port.granular(2, 0, smp, T, pos, grainLength=gl, overlaps=overlaps, st=st0)
    if rc == -3:                                                   # > 8 grains alive: both sides must refuse
        pytest.skip("capacity")
    assert rc == 0
    assert_bits_equal(o, e, "playAtPosition")
    assert_bits_equal(bank.state.numpy(), st, "state")
    assert_bits_equal(bank.grains.numpy(), gst, "grains")


def test_pitch_shift_event_driven_long_run(mx, port):
    """maxiPitchShift over a run long enough for `position` to pass the end of the sample (reset to 0, :415) several
    times, fractional start positions and a cycles counter that starts far from 0."""
    rng = np.random.default_rng(314)
    L, S, T = 9000, 48, 30000
    smp = rng.uniform(-1, 1, L)
    speed = rng.uniform(-1.5, 2.0, S)
    pm = rng.uniform(-0.05, 0.05, S)
    sb = mx.maxiSampleBank(1)
    sb.setSample(smp)
    bank = mx.maxiPitchShiftBank(S, sb, "hann")
    st0 = np.zeros((4, S))
    st0[0] = rng.uniform(0, L, S)
    st0[0, :4] = [0.0, L, L - 0.5, 123.25]
    st0[1] = rng.integers(0, 50000, S).astype(np.float64)
    bank.state.upload(st0)                                        # (`cycles` is a long in the reference: integers only)
    o = bank.play(speed, 0.02, 3, T, posMod=pm).numpy()
    e, st, gst, rc = port.granular(3, 0, smp, T, speed, posMod=pm, grainLength=0.02, overlaps=3, st=st0)
    assert rc == 0
    assert_bits_equal(o, e, "pitchshift")
    assert_bits_equal(bank.state.numpy(), st, "state")
    assert_bits_equal(bank.grains.numpy(), gst, "grains")


@pytest.mark.parametrize("knobs", [{b"grain_unit": 0}, {b"grain_line": 0}, {b"grain_lanes_k": 16}, {b"grain_lanes_k": 4096},
                                   {b"grain_streamed": 0}, {b"grain_streamed": 0, b"grain_slices": 1},
                                   {b"grain_streamed": 0, b"grain_slices": 2}, {b"grain_streamed": 0, b"grain_slices": 16},
                                   {b"grain_streamed": 1, b"grain_fast_sched": 0}],
                         ids=lambda k: ",".join("%s=%d" % (a.decode(), b) for a, b in k.items()))
def test_granular_launch_knobs_same_bits(mx, knobs):
    """The coalesced unit-increment render vs the general (stream, chunk) render, the chunking granularity of the latter, the
    unit path as ONE launch (scheduler lanes beside the tile renders, the default) vs time slices on the auxiliary streams, and
    the one-launch form with the sample-by-sample scheduler (rows published at tile boundaries): same output, scheduler state
    and live grains."""
    L = mx.lib()
    rng = np.random.default_rng(9)
    Ls, S, T = 50000, 100, 5000
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(-1.5, 2.0, S)

    def run(mode):
        bank = make_bank(mx, mode, "hann", smp, S)
        bank.setPosition(np.arange(S) / S)
        o = bank.play(speed, 0.05, 4, T).numpy() if mode == 0 else bank.play(speed, np.abs(speed) + 0.1, 0.05, 4, T).numpy()
        return o, bank.state.numpy(), bank.grains.numpy()
    ref = run(0) + run(1)
    prev = {k: L.mxg_tune(k, v) for k, v in knobs.items()}
    try:
        got = run(0) + run(1)
    finally:
        for k, v in prev.items():
            L.mxg_tune(k, v)
    for r, g in zip(ref, got):
        assert_bits_equal(g, r, str(knobs))


def test_granular_streamed_renders_wait_for_slow_schedulers(mx, port):
    """The one-launch unit path with streams whose scheduler walks sample by sample (negative speeds) next to event-driven ones,
    over enough tiles that renders are dispatched long before their rows exist: they wait on the progress words, and the bits
    are those of the sliced form and of the oracle."""
    L = mx.lib()
    rng = np.random.default_rng(77)
    Ls, S, T = 30000, 200, 64 * 600 + 17
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.2, 2.0, S) * np.where(np.arange(S) % 7 == 3, -1.0, 1.0)
    pos0 = rng.uniform(0, 1, S)

    def run():
        bank = make_bank(mx, 0, "hann", smp, S)
        bank.setPosition(pos0)
        st0 = bank.state.numpy()
        o = bank.play(speed, 0.05, 4, T).numpy()
        return o, bank.state.numpy(), bank.grains.numpy(), st0
    got = run()
    prev = L.mxg_tune(b"grain_streamed", 0)
    try:
        ref = run()
    finally:
        L.mxg_tune(b"grain_streamed", prev)
    for g, r, what in zip(got[:3], ref[:3], ("output", "state", "grains")):
        assert_bits_equal(g, r, what)
    sel = np.arange(0, S, 13)
    e, est, egst, rc = port.granular(0, 0, smp, T, speed[sel], grainLength=0.05, overlaps=4, st=got[3][:, sel])
    assert rc == 0
    assert_bits_equal(got[0][:, sel], e, "oracle")


@pytest.mark.parametrize("mode", [0, 1])
def test_streamed_renders_that_give_up_are_rendered_again(mx, port, mode):
    """a tile render of the one-launch form that sees no scheduler progress for its polling budget must not leave
    silence.  Forced here with the budget lowered to one poll (knob grain_spin_limit): most renders give up at once, the retry
    kernel that follows the launch renders every tile again from the completed lists, and output, scheduler state, live grains and
    the fused mixdown are the sliced form's bits = the oracle's; no asynchronous error is raised; mxg_granular_retries() counts the
    calls that took the detour.  mode 0 = K8c (unit increments), mode 1 = K8d (maxiStretch)."""
    L = mx.lib()
    rng = np.random.default_rng(606 + mode)
    Ls, S, T = 30000, 200, 64 * 300 + 17
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.2, 2.0, S) * np.where(np.arange(S) % 7 == 3, -1.0, 1.0)
    stretch = rng.uniform(0.5, 1.5, S)
    pan = rng.uniform(0, 1, S)
    pos0 = rng.uniform(0, 1, S)

    def run():
        bank = make_bank(mx, mode, "hann", smp, S)
        bank.setPosition(pos0)
        st0 = bank.state.numpy()
        out = mx.DeviceBuffer((T, S), zero=False)
        mix = mx.DeviceBuffer((T, 2))
        a, b, dp = mx.DeviceBuffer.from_numpy(speed), mx.DeviceBuffer.from_numpy(stretch), mx.DeviceBuffer.from_numpy(pan)
        assert L.mxg_granular_render_mix(bank._plan(0.05), mode, S, T, bank.sample.d_samples, Ls, 4, a.ptr, b.ptr if mode == 1 else None,
                                         None, None, 0, bank.state.ptr, bank.grains.ptr, out.ptr, dp.ptr, mix.ptr, None) == 0
        assert L.mxg_sync() == 0, L.mxg_last_error()
        return out.numpy(), bank.state.numpy(), bank.grains.numpy(), mix.numpy(), st0
    prev = L.mxg_tune(b"grain_streamed", 0)
    try:
        ref = run()
    finally:
        L.mxg_tune(b"grain_streamed", prev)
    before = L.mxg_granular_retries()
    calm = run()
    assert L.mxg_granular_retries() == before, "a healthy launch takes no detour"
    prev = L.mxg_tune(b"grain_spin_limit", 1)
    try:
        got = run()
    finally:
        L.mxg_tune(b"grain_spin_limit", prev)
    assert L.mxg_granular_retries() > before, "the lowered polling budget must have forced at least one retry"
    assert L.mxg_last_async_error() == 0
    for g, c, r, what in zip(got[:4], calm[:4], ref[:4], ("output", "state", "grains", "mix")):
        assert_bits_equal(g, r, what + " (forced retry vs sliced form)")
        assert_bits_equal(c, r, what + " (one launch vs sliced form)")
    sel = np.arange(0, S, 13)
    e, est, egst, rc = port.granular(mode, 0, smp, T, speed[sel], grainLength=0.05, overlaps=4, st=got[4][:, sel],
                                     **({"b": stretch[sel]} if mode == 1 else {}))
    assert rc == 0
    assert_bits_equal(got[0][:, sel], e, "oracle")


@pytest.mark.parametrize("streamed", [1, 0])
def test_time_stretch_with_off_grid_carried_grains_picks_the_general_renderer_on_the_device(mx, port, streamed):
    """maxiTimeStretch with unit increments takes the closed-form tile render K8c -- unless a carried-in grain sits between two
    buffer elements.  That is device state: both renderers are enqueued, a word written by the call's first kernel picks one, and
    every workgroup of the other returns at once (in the one-launch form: its scheduler lanes too).  Live grains uploaded at
    x.5 positions in every third stream: the call must equal the port, and so must the next call (which finds grains K8d left)."""
    L = mx.lib()
    rng = np.random.default_rng(515)
    Ls, S, T = 20000, 150, 64 * 40 + 9
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.3, 1.7, S)
    prev = L.mxg_tune(b"grain_streamed", streamed)
    try:
        bank = make_bank(mx, 0, "hann", smp, S)
        bank.setPosition(rng.uniform(0, 1, S))
        dur = int(0.05 * 44100)
        gst = np.zeros((4, 8, S))
        for s in range(0, S, 3):
            gst[0, 0, s] = 100.5 + 7 * s
            gst[1, 0, s] = 1.0 if s % 2 else -1.0
            gst[2, 0, s] = 10 + s
            gst[3, 0, s] = dur
        bank.grains.upload(gst)
        st = bank.state.numpy()
        for n in (T, 1000):
            o = bank.play(speed, 0.05, 4, n).numpy()
            e, st, gst, rc = port.granular(0, 0, smp, n, speed, grainLength=0.05, overlaps=4, st=st, gst=gst)
            assert rc == 0
            assert_bits_equal(o, e, "output (%d samples)" % n)
            assert_bits_equal(bank.state.numpy(), st, "scheduler state")
            assert_bits_equal(bank.grains.numpy(), gst, "live grains")
    finally:
        L.mxg_tune(b"grain_streamed", prev)


@pytest.mark.parametrize("unit", [1, 0])
def test_granular_signed_zeros_and_non_finite_samples(mx, port, unit):
    """`(1-remainder)*buffer[a] + remainder*buffer[b]` with remainder == 0 still depends on buffer[b]: 0*Inf and 0*NaN
    are NaN, and -0.0 + 0*b is +0.0 or -0.0 with the sign of b.  A sample buffer sprinkled with -0.0, +-Inf and NaN,
    forward and backward grains, wraps, a ragged last tile and a partial last wavefront: the unit-increment render (whose
    interior pairs take buffer[a+1] from the neighbouring lane) and the general render both reproduce the oracle."""
    rng = np.random.default_rng(4242)
    Ls, S, T = 9000, 70, 3000 + 37
    smp = rng.uniform(-1, 1, Ls)
    idx = rng.choice(Ls, 120, replace=False)
    smp[idx[:40]] = -0.0
    smp[idx[40:60]] = 0.0
    smp[idx[60:80]] = np.inf
    smp[idx[80:100]] = -np.inf
    smp[idx[100:]] = np.nan
    smp[[0, Ls - 1]] = [-0.0, np.inf]                       # the wrap pair (b = buffer[0] after a = buffer[len-1])
    speed = rng.uniform(0.3, 1.8, S) * np.where(np.arange(S) % 3 == 0, -1.0, 1.0)
    prev = mx.lib().mxg_tune(b"grain_unit", unit)
    try:
        bank = make_bank(mx, 0, "hann", smp, S)
        bank.setPosition(rng.uniform(0, 1, S))
        st0 = bank.state.numpy()
        o = bank.play(speed, 0.05, 4, T).numpy()
    finally:
        mx.lib().mxg_tune(b"grain_unit", prev)
    e, est, egst, rc = port.granular(0, 0, smp, T, speed, grainLength=0.05, overlaps=4, st=st0)
    assert rc == 0
    assert np.isnan(e).any() and np.isfinite(e).any()
    assert_bits_equal(o, e, "output")
    assert_bits_equal(bank.state.numpy(), est, "state")
    assert_bits_equal(bank.grains.numpy(), egst, "grains")


def test_scheduler_position_stuck_on_the_wrap_limit(mx, port):
    """maxiStretch (`position >= len` wraps, L/maxiGrains.h:516) with a rate below half an ulp of the position: a head that
    sits exactly on len does not move when the rate is added, but the test still fires on the next sample and wraps
    it; a head just below len stays there for ever.  (Found by the host fuzz of mxg_advance.h.)"""
    rng = np.random.default_rng(5150)
    Ls, S, T = 262144, 64, 4000
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.5, 1.5, S)
    rate = np.full(S, 2.0 ** -40)
    rate[1::4] = 2.0 ** -36 * 1.37
    rate[2::4] = 0.75
    st0 = np.zeros((4, S))
    st0[0] = float(Ls)                       # exactly on the limit
    st0[0, 3::4] = np.nextafter(float(Ls), 0.0)
    bank = make_bank(mx, 1, "hann", smp, S)
    bank.state.upload(st0)
    o = bank.play(speed, rate, 0.05, 4, T).numpy()
    e, est, egst, rc = port.granular(1, 0, smp, T, speed, b=rate, grainLength=0.05, overlaps=4, st=st0)
    assert rc == 0
    assert_bits_equal(bank.state.numpy(), est, "scheduler state")
    assert_bits_equal(bank.grains.numpy(), egst, "grains")
    assert_bits_equal(o, e, "output")


@pytest.mark.parametrize("mode", [0, 1])
def test_foreign_or_corrupt_live_grains_are_refused(mx, mode):
    """d_gst is caller-visible state.  A live grain made by a plan of another grain length (its window index would run
    past this plan's window table), a negative window index, a position outside the sample or a NaN step are refused
    with MXG_ERR_INVALID by every render path instead of being used as array indices."""
    rng = np.random.default_rng(12)
    Ls, S, T = 20000, 96, 700
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.5, 1.5, S)
    dur = int(0.05 * 44100)
    good = np.zeros((4, 8, S))
    good[:, 0, :] = np.array([100.0, 1.0, 10.0, dur])[:, None]        # one live grain per stream, this plan's
    for what, row, value in (("another grain length", 3, 4410.0), ("negative window index", 2, -3.0),
                             ("window index past the end", 2, dur + 5.0), ("position outside the sample", 0, 3.0 * Ls),
                             ("NaN step", 1, np.nan)):
        for unit in (1, 0):
            prev = mx.lib().mxg_tune(b"grain_unit", unit)
            try:
                bank = make_bank(mx, mode, "hann", smp, S)
                g = good.copy()
                g[row, 0, 17] = value
                bank.grains.upload(g)
                with pytest.raises(mx.MaxiGpuError, match="could not have made"):
                    if mode == 0:
                        bank.play(speed, 0.05, 4, T).numpy()
                    else:
                        bank.play(speed, speed, 0.05, 4, T).numpy()
                bank.grains.upload(good)                               # the same call with a clean state works
                o = (bank.play(speed, 0.05, 4, T) if mode == 0 else bank.play(speed, speed, 0.05, 4, T)).numpy()
                assert np.isfinite(o).all(), what
            finally:
                mx.lib().mxg_tune(b"grain_unit", prev)


@pytest.mark.parametrize("mode,S,T", [(0, 200, 700), (0, 64, 64), (0, 2100, 1000), (1, 130, 300)])
def test_granular_render_with_fused_mixdown(mx, port, mode, S, T):
    """mxg_granular_render_mix: the per-stream block and the carried state are bit-identical to mxg_granular_render, the
    mix equals the tree-ordered K3 mixdown of that block within the mix tolerance and the oracle's sequential sum likewise.
    Mode 0 at 44.1 kHz takes the unit-increment path (mix fused into the render kernel: stream counts that are not a multiple
    of the 64-stream tile, a ragged last sample tile, more than one stream tile); mode 1 mixes with K3 after the render."""
    rng = np.random.default_rng(S + T)
    Ls = 30000
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.3, 1.7, S)
    stretch = rng.uniform(0.5, 1.5, S)
    pan = rng.uniform(-0.1, 1.1, S)
    L = mx.lib()

    def run(fused):
        bank = make_bank(mx, mode, "hann", smp, S)
        bank.setPosition(np.arange(S) / S)
        out = mx.DeviceBuffer((T, S), zero=False)
        mix = mx.DeviceBuffer((T, 2))
        a, b = mx.DeviceBuffer.from_numpy(speed), mx.DeviceBuffer.from_numpy(stretch)
        dp = mx.DeviceBuffer.from_numpy(pan)
        plan = bank._plan(0.05)
        if fused:
            assert L.mxg_granular_render_mix(plan, mode, S, T, bank.sample.d_samples, Ls, 4, a.ptr, b.ptr if mode == 1 else None,
                                             None, None, 0, bank.state.ptr, bank.grains.ptr, out.ptr, dp.ptr, mix.ptr, None) == 0
        else:
            assert L.mxg_granular_render(plan, mode, S, T, bank.sample.d_samples, Ls, 4, a.ptr, b.ptr if mode == 1 else None,
                                         None, None, 0, bank.state.ptr, bank.grains.ptr, out.ptr, None) == 0
            assert L.mxg_mix_stereo(S, T, out.ptr, dp.ptr, mix.ptr, None) == 0
        return out.numpy(), mix.numpy(), bank.state.numpy(), bank.grains.numpy()
    o1, m1, s1, g1 = run(True)
    o0, m0, s0, g0 = run(False)
    assert_bits_equal(o1, o0, "per-stream block")
    assert_bits_equal(s1, s0, "scheduler state")
    assert_bits_equal(g1, g0, "live grains")
    em = port.mix_stereo(o0, pan)
    scale = max(1.0, np.abs(o0).max())
    err = max(np.abs(m1 - em).max(), np.abs(m0 - em).max())
    print("granular mixdown S=%d: max |mix - sequential sum| = %.3e (%.2e x S x peak)" % (S, err, err / (S * scale)))
    assert err <= mix_tol(S, scale)


@pytest.mark.parametrize("mode", [1, 3])
@pytest.mark.parametrize("Ls", [300, 70000])
def test_tile_render_lines_and_their_exceptions(mx, port, mode, Ls):
    """K8d evaluates sample k of a grain on a line over the mantissa grid of its binade; the (grain, tile) pairs without one
    walk their steps: grains wrapping around a short sample several times per tile row, positions crossing powers of two,
    increments whose ulp remainder is exactly one half in some binade (ties), increments far below an ulp... plus backward
    grains, three carried launches of ragged lengths (live grains carried in) and a ragged last tile.  Bit-identical to the
    port, scheduler and live-grain state included."""
    rng = np.random.default_rng(60 + mode + Ls)
    S = 160
    smp = rng.uniform(-1, 1, Ls)
    speed = rng.uniform(0.2, 2.2, S)
    speed[::5] = 1.0 + 2.0 ** -rng.integers(18, 46, speed[::5].size)     # short mantissas: exact halves of an ulp somewhere
    speed[1::16] = 2.0 ** -rng.integers(1, 30, speed[1::16].size)           # tiny steps: many samples per buffer element
    speed[2::16] *= -1.0                                                     # backwards
    speed[3::16] = rng.uniform(4.0, 30.0, speed[3::16].size)               # several buffer elements per sample
    if mode == 1:
        bank = mx.maxiStretchBank(S, _sample_bank(mx, smp), "hann")
        play = lambda n: bank.play(speed, np.abs(speed) * 0.7 + 0.05, 0.05, 4, n).numpy()
        ref = lambda n, st, gst: port.granular(1, 0, smp, n, speed, b=np.abs(speed) * 0.7 + 0.05, st=st, gst=gst)
    else:
        bank = mx.maxiPitchShiftBank(S, _sample_bank(mx, smp), "hann")
        play = lambda n: bank.play(speed, 0.05, 4, n).numpy()
        ref = lambda n, st, gst: port.granular(3, 0, smp, n, speed, st=st, gst=gst)
    bank.setPosition(np.arange(S) / S)
    st, gst = bank.state.numpy(), bank.grains.numpy()
    for n in (3000, 1000, 333):
        o = play(n)
        e, st, gst, rc = ref(n, st, gst)
        assert rc == 0
        assert_bits_equal(o, e, "mode %d, %d samples" % (mode, n))
        assert_bits_equal(bank.state.numpy(), st, "scheduler state")
        assert_bits_equal(bank.grains.numpy(), gst, "live grains")



def test_config5_render_is_asynchronous_and_graph_capturable(mx):
    """mxg_granular_render_mix (maxiTimeStretch + fused stereo mixdown: the config-5 step) makes no synchronising HIP call --
    the path choice that depends on device state is taken on the device, errors are deferred (mxg_last_async_error) -- so its
    launch sequence, fork / join onto the library's auxiliary stream included, can be captured into a hipGraph.  K replays
    must leave the outputs, the scheduler state and the live grains of K eager calls."""
    import torch
    L = mx.lib()
    S, T, Ls, K = 256, 4096, 200000, 3
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    smp = rng.uniform(-1, 1, Ls)
    sb = mx.maxiSampleBank(1)
    sb.setSample(smp)
    speed = torch.from_numpy(0.25 + 1.5 * (np.arange(S) % 97) / 96).to(dev)
    pan = torch.from_numpy(np.arange(S) / (S - 1.0)).to(dev)
    plan = L.mxg_grain_plan_create(0, 0.05, 44100)
    assert plan

    def fresh():
        st0 = np.zeros((4, S))
        st0[0] = np.arange(S) / S * Ls
        return (torch.from_numpy(st0).to(dev), torch.zeros((4, 8, S), dtype=torch.float64, device=dev),
                torch.empty((T, S), dtype=torch.float64, device=dev), torch.zeros((T, 2), dtype=torch.float64, device=dev))

    def call(st, state, grains, out, mix):
        assert L.mxg_granular_render_mix(plan, 0, S, T, sb.d_samples, Ls, 4, speed.data_ptr(), None, None, None, 0,
                                         state.data_ptr(), grains.data_ptr(), out.data_ptr(), pan.data_ptr(), mix.data_ptr(), st) == 0

    s = torch.cuda.Stream(device=dev)
    st = s.cuda_stream
    e = fresh()
    with torch.cuda.stream(s):
        for _ in range(K):
            call(st, *e)
    s.synchronize()
    assert L.mxg_last_async_error() == 0
    g_state, warm = fresh(), fresh()
    with torch.cuda.stream(s):
        call(st, *warm)                      # allocates the library's per-stream scratch and its auxiliary stream
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        call(st, *g_state)
    for _ in range(K):
        graph.replay()
    torch.cuda.synchronize()
    assert L.mxg_last_async_error() == 0
    for i, what in enumerate(("scheduler state", "live grains", "last block", "last mix")):
        assert torch.equal(g_state[i], e[i]), what
    L.mxg_grain_plan_destroy(plan)


def test_one_launch_granular_calls_on_two_streams_at_once(mx):
    """Two banks rendered on two streams with nothing ordering them: their one-launch calls (scheduler workgroups + tile renders that
    poll them) share the device, and each render only ever waits for scheduler workgroups of ITS OWN launch, which were dispatched
    before it.  Three calls per stream, interleaved enqueue, a large unrelated kernel in between; outputs, scheduler state and live
    grains equal the same calls made one stream after the other."""
    import torch
    L = mx.lib()
    S, T, Ls, K = 512, 64 * 300, 150000, 3
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(15)
    smp = rng.uniform(-1, 1, Ls)
    sb = mx.maxiSampleBank(1)
    sb.setSample(smp)
    plan = L.mxg_grain_plan_create(0, 0.05, 44100)
    assert plan
    speeds = [torch.from_numpy(0.3 + 1.4 * ((np.arange(S) * (3 + 2 * b)) % 89) / 88).to(dev) for b in range(2)]
    pan = torch.from_numpy(np.arange(S) / (S - 1.0)).to(dev)

    def fresh(b):
        st0 = np.zeros((4, S))
        st0[0] = ((np.arange(S) * (b + 1)) % S) / S * Ls
        return (torch.from_numpy(st0).to(dev), torch.zeros((4, 8, S), dtype=torch.float64, device=dev),
                torch.empty((T, S), dtype=torch.float64, device=dev), torch.zeros((T, 2), dtype=torch.float64, device=dev))

    def call(st, b, state, grains, out, mix):
        assert L.mxg_granular_render_mix(plan, 0, S, T, sb.d_samples, Ls, 4, speeds[b].data_ptr(), None, None, None, 0,
                                         state.data_ptr(), grains.data_ptr(), out.data_ptr(), pan.data_ptr(), mix.data_ptr(), st) == 0

    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    ref = [fresh(b) for b in range(2)]
    for b in range(2):                       # one stream after the other
        with torch.cuda.stream(streams[b]):
            for _ in range(K):
                call(streams[b].cuda_stream, b, *ref[b])
        streams[b].synchronize()
    got = [fresh(b) for b in range(2)]
    big = torch.empty((4096, 4096), dtype=torch.float32, device=dev).normal_()
    torch.cuda.synchronize()
    for _ in range(K):                       # both at once
        for b in range(2):
            with torch.cuda.stream(streams[b]):
                call(streams[b].cuda_stream, b, *got[b])
        big = big @ big * 1e-4               # (the default stream: a third party on the device)
    torch.cuda.synchronize()
    assert L.mxg_last_async_error() == 0
    for b in range(2):
        for i, what in enumerate(("scheduler state", "live grains", "last block", "last mix")):
            assert torch.equal(got[b][i], ref[b][i]), "stream %d: %s" % (b, what)
    L.mxg_grain_plan_destroy(plan)
