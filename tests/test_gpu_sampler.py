"""GPU parity (-m gpu): maxiSampler banks (L/maxiSynths.h:137-187) through the C-ABI vs the oracle: per-slot
outputs, the in-order mix, slot state (position, trigger, envelope) -- all bit-exact -- and the host-side
control mirror (trigger(), midiNoteOn/Off round robin)."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def _case(voices, NS, seed):
    rng = np.random.default_rng(seed)
    V = NS * voices
    return dict(
        smp=rng.uniform(-1, 1, 3000), pitch=rng.integers(-24, 25, V).astype(np.float64), gain=rng.uniform(0.2, 1.0, V),
        par=np.stack([rng.uniform(0.001, 0.2, V), rng.uniform(0.99, 0.9999, V), rng.uniform(0.3, 1.0, V),
                      rng.uniform(0.99, 0.9999, V)]),
        hold=rng.integers(1, 50, V), trig=(rng.uniform(size=V) < 0.6).astype(np.int32))


@pytest.mark.parametrize("voices", [32, 8, 1, 3, 6, 21, 31])
@pytest.mark.parametrize("sustain", [1, 0])
def test_sampler_render_vs_oracle(mx, port, voices, sustain):
    NS, N = 7, 1500
    c = _case(voices, NS, 100 + voices)
    V = NS * voices
    L = mx.lib()
    d = lambda a: mx.DeviceBuffer.from_numpy(np.ascontiguousarray(a))
    sb = mx.maxiSampleBank(1)
    sb.setSample(c["smp"])
    freq = np.zeros(V)
    assert L.mxg_sampler_freq_host(V, c["pitch"].ctypes.data, c["smp"].size, freq.ctypes.data) == 0
    _set_restype(port)
    assert_bits_equal(freq, np.array([port.L.mxo_sampler_frequency(p, c["smp"].size) for p in c["pitch"]]), "freq")
    dpos, dtrig, dout = d(np.zeros(V)), d(c["trig"]), d(np.zeros(V))
    ddst, dist = d(np.zeros((2, V))), d(np.zeros((6, V), np.int64))
    dfreq, dgain, dpar, dhold = d(freq), d(c["gain"]), d(c["par"]), d(c["hold"].astype(np.int64))
    outs, mixes = [], []
    e_state = None
    for blk in range(2):
        mix, outputs = mx.DeviceBuffer((N, NS)), mx.DeviceBuffer((N, V))
        if blk == 1:    # control events between renders: note-offs and re-triggers
            t = dtrig.numpy(); t[::3] = 0; t[1::7] = 1; dtrig.upload(t)
        assert L.mxg_sampler_render(V, N, voices, sustain, sb.d_samples, c["smp"].size, dfreq.ptr, dgain.ptr, dpar.ptr,
                                    dhold.ptr, dpos.ptr, dtrig.ptr, dout.ptr, ddst.ptr, dist.ptr, mix.ptr, outputs.ptr,
                                    None) == 0
        if blk == 0:
            e = port.sampler(voices, c["smp"], N, c["pitch"], c["gain"], c["par"], c["hold"], np.zeros(V), c["trig"], sustain)
        else:
            t2 = e[3].copy(); t2[::3] = 0; t2[1::7] = 1
            e = port.sampler(voices, c["smp"], N, c["pitch"], c["gain"], c["par"], c["hold"], e[2], t2, sustain, e[4], e[5], e[6])
        assert_bits_equal(outputs.numpy(), e[1], "outputs[i], block %d" % blk)
        assert_bits_equal(mix.numpy(), e[0], "play(), block %d" % blk)
        assert_bits_equal(dpos.numpy(), e[2], "position")
        assert np.array_equal(dtrig.numpy(), e[3])
        assert_bits_equal(dout.numpy(), e[4], "outputs hold")
        assert_bits_equal(ddst.numpy(), e[5], "env amplitude/output")
        assert np.array_equal(dist.numpy(), e[6])
    assert np.abs(e[0]).max() > 0.01
    assert L.mxg_sampler_render(V, N, 33, sustain, sb.d_samples, c["smp"].size, dfreq.ptr, dgain.ptr, dpar.ptr, dhold.ptr,
                                dpos.ptr, dtrig.ptr, dout.ptr, ddst.ptr, dist.ptr, mix.ptr, None, None) == -1


def _set_restype(port):
    import ctypes
    port.L.mxo_sampler_frequency.restype = ctypes.c_double
    port.L.mxo_sampler_frequency.argtypes = [ctypes.c_double, ctypes.c_size_t]
    return True


def test_sampler_bank_control_mirror(mx, port):
    """maxiSamplerBank: trigger()/midiNoteOn round robin on the host, play() on the device -- against the
    oracle driven with the same slot edits."""
    voices, NS, N = 8, 3, 800
    rng = np.random.default_rng(5)
    smp = rng.uniform(-1, 1, 4000)
    bank = mx.maxiSamplerBank(NS, voices)
    bank.setSample(smp)
    V = NS * voices
    trig = np.zeros(V, np.int32)
    pos = np.full(V, smp.size - 1.0)
    pitch, gain = np.zeros(V), np.zeros(V)
    cur = 0
    state = None
    for step, note in enumerate([0, 7, -5, 12]):
        bank.midiNoteOn(note, 100)
        bank.trigger()
        sl = np.arange(NS) * voices + cur
        pitch[sl] = note; gain[sl] = 100 / 128; trig[sl] = 1; pos[sl] = 0.0
        cur = (cur + 1) % voices
        if step == 2:
            bank.midiNoteOff(7)
            trig[pitch == 7] = 0
        mix = bank.play(N, want_outputs=True).numpy()
        args = (voices, smp, N, pitch, gain, bank.env.par, bank.env.holdtime, pos, trig, True)
        e = port.sampler(*args) if state is None else port.sampler(*args, state[0], state[1], state[2])
        assert_bits_equal(mix, e[0], "play() after note %d" % note)
        assert_bits_equal(bank.outputs.numpy(), e[1])
        pos, trig, state = e[2], e[3], (e[4], e[5], e[6])
    assert np.abs(mix).max() > 0.01
