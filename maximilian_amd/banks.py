"""Host-side mirror of the reference's operator API, one *bank* of V instances per object.

Names follow the reference (src/maximilian.h): `maxiOscBank.sinebuf(freq, N)` renders what N
consecutive calls of `maxiOsc::sinebuf(freq[v])` on each of the V oscillators would return,
as an [N, V] sample-major block that stays on the device.  State (phase, filter memories,
envelope flags ...) lives in device SoA arrays owned by the bank and carries from block to
block, exactly like consecutive `play()` callbacks.  Everything here is plumbing over the
C-ABI (maximilian_amd/_lib.py); no arithmetic on signals happens in Python.
"""
import ctypes
import os

import numpy as np

from ._lib import check, lib

OSC_WAVEFORMS = {
    "sinewave": 0, "coswave": 1, "phasor": 2, "saw": 3, "triangle": 4, "square": 5, "pulse": 6,
    "impulse": 7, "sinebuf": 8, "sinebuf4": 9, "sawn": 10, "phasorBetween": 11,
}
FILTER_KINDS = {"lores": 0, "hires": 1, "bandpass": 2, "lopass": 3, "hipass": 4}


class DeviceBuffer:
    """A typed device allocation (hipMalloc via mxg_malloc) with numpy upload/download.

    Stream contract: banks launch on their own `stream`, while allocation-time zeroing and the
    copies here go through the library's default stream.  So the zeroing is complete before the
    constructor returns, upload() is synchronous (mxg_memcpy_h2d waits), and numpy() waits for ALL
    outstanding device work (mxg_sync) before it copies -- a block a kernel on any stream is still
    writing is never read half-finished, and a first render never races the zero fill."""

    def __init__(self, shape, dtype=np.float64, zero=True):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = lib().mxg_malloc(max(self.nbytes, 8))
        if not self.ptr:
            raise MemoryError("mxg_malloc(%d): %s" % (self.nbytes, lib().mxg_last_error().decode()))
        if zero and self.nbytes:
            check(lib().mxg_memset(self.ptr, 0, self.nbytes, None), "mxg_memset")
            check(lib().mxg_stream_sync(None), "mxg_stream_sync")

    @classmethod
    def from_numpy(cls, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype if dtype is not None else a.dtype)
        b = cls(a.shape, a.dtype, zero=False)
        b.upload(a)
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        if self.nbytes:
            check(lib().mxg_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes, None), "mxg_memcpy_h2d")
        return self

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(lib().mxg_sync(), "mxg_sync")  # the producer may have run on any stream
            check(lib().mxg_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None), "mxg_memcpy_d2h")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            lib().mxg_free(self.ptr)
            self.ptr = None

    def __del__(self):  # best effort
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    """Device pointer of a DeviceBuffer / torch tensor / raw int / None."""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


def _as_dev(x, V, dtype=np.float64):
    """Per-voice parameter: scalar / array-like -> DeviceBuffer [V]; device objects pass through."""
    if isinstance(x, DeviceBuffer) or hasattr(x, "data_ptr"):
        return x
    a = np.asarray(x, dtype=dtype)
    if a.ndim == 0:
        a = np.full(V, a, dtype=dtype)
    return DeviceBuffer.from_numpy(a)


class maxiSettings:
    """maxiSettings (H:117-163): global sampleRate / channels / bufferSize."""
    sampleRate, channels, bufferSize = 44100, 2, 1024

    @classmethod
    def setup(cls, sampleRate, channels, bufferSize):
        check(lib().mxg_settings(sampleRate, channels, bufferSize), "mxg_settings")
        cls.sampleRate, cls.channels, cls.bufferSize = sampleRate, channels, bufferSize


class _Bank:
    def __init__(self, voices, stream=None):
        check(lib().mxg_init(-1), "mxg_init")
        self.V = int(voices)
        self.stream = stream

    def _out(self, N, out):
        return out if out is not None else DeviceBuffer((N, self.V), np.float64, zero=False)

    def sync(self):
        check(lib().mxg_stream_sync(self.stream), "mxg_stream_sync")


class maxiOscBank(_Bank):
    """V x maxiOsc (H:169-215).  Methods: one per reference waveform, `(freq, N)` -> [N, V]."""

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.phase = DeviceBuffer(self.V)   # maxiOsc::phase  (ctor sets 0, C:209-212)
        self.output = DeviceBuffer(self.V)  # maxiOsc::output (held by square/pulse)

    def phaseReset(self, phaseIn):
        """maxiOsc::phaseReset (C:222-226), per voice."""
        self.phase.upload(np.broadcast_to(np.asarray(phaseIn, np.float64), (self.V,)))

    def render(self, waveform, freq, N, p1=None, p2=None, out=None, per_sample=False, pitch=None):
        """`pitch`: row pitch of the output in DOUBLES (mxg_osc_render_pitch): the block is then a [N, pitch] buffer whose first V
        columns are the voices (the rest is padding nobody writes); None = V."""
        wf = OSC_WAVEFORMS[waveform] if isinstance(waveform, str) else int(waveform)
        if per_sample:
            f = freq if (isinstance(freq, DeviceBuffer) or hasattr(freq, "data_ptr")) else \
                DeviceBuffer.from_numpy(np.asarray(freq, np.float64).reshape(N, self.V))
        else:
            f = _as_dev(freq, self.V)
        a = None if p1 is None else _as_dev(p1, self.V)
        b = None if p2 is None else _as_dev(p2, self.V)
        if pitch is not None and int(pitch) != self.V:
            out = out if out is not None else DeviceBuffer((N, int(pitch)), np.float64)
            check(lib().mxg_osc_render_pitch(wf, self.V, N, _ptr(f), 1 if per_sample else 0, _ptr(a), _ptr(b), self.phase.ptr,
                                             self.output.ptr, _ptr(out), int(pitch) * 8, self.stream), "mxg_osc_render_pitch")
            self._keep = (f, a, b)
            return out
        out = self._out(N, out)
        check(lib().mxg_osc_render(wf, self.V, N, _ptr(f), 1 if per_sample else 0, _ptr(a), _ptr(b),
                                   self.phase.ptr, self.output.ptr, _ptr(out), self.stream),
              "mxg_osc_render")
        self._keep = (f, a, b)  # keep parameter buffers alive until the next call
        return out

    def render_mix(self, waveform, freq, pan, N, p1=None, p2=None, out=None, store=True, mix=None, rows=None):
        """Render + fused maxiMix::stereo mixdown in one pass (mxg_osc_render_mix).
        Returns (out [N,V] or None, mix [N,2]).  rows: a device pointer / buffer [mxg_osc_mix_groups(V)][N][2] (e.g. a
        grouped mix queue's slot) -- then the per-workgroup rows are left there (mxg_osc_render_mix_rows) and no mix is
        formed here; returns (out, None)."""
        wf = OSC_WAVEFORMS[waveform] if isinstance(waveform, str) else int(waveform)
        f, pn = _as_dev(freq, self.V), _as_dev(pan, self.V)
        a = None if p1 is None else _as_dev(p1, self.V)
        b = None if p2 is None else _as_dev(p2, self.V)
        out = (self._out(N, out) if store else None)
        if rows is not None:
            check(lib().mxg_osc_render_mix_rows(wf, self.V, N, _ptr(f), _ptr(a), _ptr(b), self.phase.ptr,
                                                self.output.ptr, _ptr(out), _ptr(pn), _ptr(rows), self.stream),
                  "mxg_osc_render_mix_rows")
            self._keep = (f, a, b, pn)
            return out, None
        mix = mix if mix is not None else DeviceBuffer((N, 2), np.float64, zero=False)
        check(lib().mxg_osc_render_mix(wf, self.V, N, _ptr(f), _ptr(a), _ptr(b), self.phase.ptr,
                                       self.output.ptr, _ptr(out), _ptr(pn), _ptr(mix), self.stream),
              "mxg_osc_render_mix")
        self._keep = (f, a, b, pn)
        return out, mix

    def sinebuf_tables(self, freq, tables, N, pan=None, store=True, out=None, ahead=False, fused_sum=False):
        """EXTENSION: sinebuf with a 514-entry table per voice (mxg_osc_render_tables).  tables: [V][514] (numpy or device).
        Returns (out [N,V] or None, mix [N,2] or None): the mix is formed from the kernel's partial rows with mxg_mix_rows_sum, or
        (fused_sum) inside the render kernel.  ahead: the pipelined form (MXG_TABLES_AHEAD): freq / pan must then be the SAME device
        buffers from call to call (pass DeviceBuffers) and self.phase untouched in between."""
        if ahead or fused_sum:
            f = _as_dev(freq, self.V)
            tb = tables if isinstance(tables, DeviceBuffer) or hasattr(tables, "data_ptr") else DeviceBuffer.from_numpy(
                np.ascontiguousarray(tables, np.float64))
            out = self._out(N, out) if store else None
            pn = rows = mix = None
            if pan is not None:
                pn = _as_dev(pan, self.V)
                G = lib().mxg_osc_tables_groups(self.V)
                rows = DeviceBuffer((G, N, 2), np.float64)
                mix = DeviceBuffer((N, 2), np.float64, zero=False)
            check(lib().mxg_osc_render_tables_ex(self.V, N, _ptr(f), _ptr(tb), self.phase.ptr, self.output.ptr, _ptr(out), _ptr(pn), _ptr(rows),
                                                 _ptr(mix) if (fused_sum and mix is not None) else None, 1 if ahead else 0, self.stream),
                  "mxg_osc_render_tables_ex")
            if pan is not None and not fused_sum:
                check(lib().mxg_mix_rows_sum(G, N * 2, rows.ptr, mix.ptr, self.stream), "mxg_mix_rows_sum")
            self._keep = (f, tb, pn, rows)
            self.last_rows = rows
            return out, mix
        f = _as_dev(freq, self.V)
        tb = tables if isinstance(tables, DeviceBuffer) or hasattr(tables, "data_ptr") else DeviceBuffer.from_numpy(
            np.ascontiguousarray(tables, np.float64))
        out = self._out(N, out) if store else None
        pn = rows = mix = None
        if pan is not None:
            pn = _as_dev(pan, self.V)
            G = lib().mxg_osc_tables_groups(self.V)
            rows = DeviceBuffer((G, N, 2), np.float64)
        check(lib().mxg_osc_render_tables(self.V, N, _ptr(f), _ptr(tb), self.phase.ptr, self.output.ptr, _ptr(out), _ptr(pn), _ptr(rows),
                                          self.stream), "mxg_osc_render_tables")
        if pan is not None:
            mix = DeviceBuffer((N, 2), np.float64, zero=False)
            check(lib().mxg_mix_rows_sum(G, N * 2, rows.ptr, mix.ptr, self.stream), "mxg_mix_rows_sum")
        self._keep = (f, tb, pn, rows)
        return out, mix

    def noise(self, rand, out=None):
        """maxiOsc::noise (C:214-220) from caller-supplied rand() draws, int32 [N][V] (draw n*V+v is the
        one a voice-inner per-sample loop hands to voice v at sample n)."""
        if not (isinstance(rand, DeviceBuffer) or hasattr(rand, "data_ptr")):
            rand = DeviceBuffer.from_numpy(np.ascontiguousarray(rand, np.int32))
        N = rand.shape[0]
        out = self._out(N, out)
        check(lib().mxg_osc_noise(self.V, N, _ptr(rand), self.output.ptr, _ptr(out), self.stream),
              "mxg_osc_noise")
        self._keep = rand
        return out

    def sinewave(self, freq, N, **kw): return self.render("sinewave", freq, N, **kw)
    def coswave(self, freq, N, **kw): return self.render("coswave", freq, N, **kw)
    def phasor(self, freq, N, **kw): return self.render("phasor", freq, N, **kw)
    def saw(self, freq, N, **kw): return self.render("saw", freq, N, **kw)
    def triangle(self, freq, N, **kw): return self.render("triangle", freq, N, **kw)
    def square(self, freq, N, **kw): return self.render("square", freq, N, **kw)
    def pulse(self, freq, duty, N, **kw): return self.render("pulse", freq, N, p1=duty, **kw)
    def impulse(self, freq, N, **kw): return self.render("impulse", freq, N, **kw)
    def sinebuf(self, freq, N, **kw): return self.render("sinebuf", freq, N, **kw)
    def sinebuf4(self, freq, N, **kw): return self.render("sinebuf4", freq, N, **kw)
    def sawn(self, freq, N, **kw): return self.render("sawn", freq, N, **kw)

    def phasorBetween(self, freq, startphase, endphase, N, **kw):
        return self.render("phasorBetween", freq, N, p1=startphase, p2=endphase, **kw)


def filter_coeffs(kind, cutoff, res):
    """Host-libm coefficients [3][V] of maxiFilter::lores/hires (c, r; C:456-461) or bandpass
    (inputs[0..2]; C:489-495), per voice."""
    k = FILTER_KINDS[kind] if isinstance(kind, str) else int(kind)
    cutoff = np.ascontiguousarray(cutoff, np.float64)
    res = np.ascontiguousarray(np.broadcast_to(np.asarray(res, np.float64), cutoff.shape))
    coef = np.zeros((3, cutoff.size))
    check(lib().mxg_filter_coeffs_host(k, cutoff.size, cutoff.ctypes.data, res.ctypes.data,
                                       coef.ctypes.data), "mxg_filter_coeffs_host")
    return coef


class maxiFilterBank(_Bank):
    """V x maxiFilter (H:289-366)."""

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.state = DeviceBuffer((5, self.V))  # x, y, outputs[0..2]  (ctor zeros, C:1517)

    def render(self, kind, x, cutoff, resonance=None, out=None, cutoff_per_sample=False,
               res_per_sample=False):
        k = FILTER_KINDS[kind] if isinstance(kind, str) else int(kind)
        N = x.shape[0]
        coef = None
        if k in (0, 1, 2) and not (cutoff_per_sample or res_per_sample):
            cu = np.broadcast_to(np.asarray(cutoff, np.float64), (self.V,))
            coef = DeviceBuffer.from_numpy(filter_coeffs(k, cu, resonance))
        cut = cutoff if cutoff_per_sample and not isinstance(cutoff, np.ndarray) else (
            DeviceBuffer.from_numpy(np.asarray(cutoff, np.float64).reshape(N, self.V))
            if cutoff_per_sample else _as_dev(cutoff, self.V))
        rs = None
        if resonance is not None:
            rs = (DeviceBuffer.from_numpy(np.asarray(resonance, np.float64).reshape(N, self.V))
                  if res_per_sample and isinstance(resonance, np.ndarray)
                  else (resonance if res_per_sample else _as_dev(resonance, self.V)))
        out = self._out(N, out)
        check(lib().mxg_filter_render(k, self.V, N, _ptr(x), _ptr(cut), int(cutoff_per_sample),
                                      _ptr(rs), int(res_per_sample), _ptr(coef), self.state.ptr,
                                      _ptr(out), self.stream), "mxg_filter_render")
        self._keep = (cut, rs, coef)
        return out

    def lores(self, x, cutoff, resonance, **kw): return self.render("lores", x, cutoff, resonance, **kw)
    def hires(self, x, cutoff, resonance, **kw): return self.render("hires", x, cutoff, resonance, **kw)
    def bandpass(self, x, cutoff, resonance, **kw): return self.render("bandpass", x, cutoff, resonance, **kw)
    def lopass(self, x, cutoff, **kw): return self.render("lopass", x, cutoff, None, **kw)
    def hipass(self, x, cutoff, **kw): return self.render("hipass", x, cutoff, None, **kw)


class maxiEnvGenBank(_Bank):
    """V x maxiEnvGen (H:2268-2547) sharing one envelope shape; per-voice (or shared) trigger signals."""
    HOLD = -46692.0

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.stages = None
        self.loop = self.retrigger = False
        self._arm()

    def _arm(self):
        d = np.zeros((5, self.V))
        d[2:5] = 1.0                                   # maxiTrigger::previousValue (H:593)
        i = np.zeros((7, self.V), np.int64)
        i[4:7] = 1                                     # firstTrigger (H:594); state WAITING (resetAndArm)
        self.dstate, self.istate = DeviceBuffer.from_numpy(d), DeviceBuffer.from_numpy(i)

    def setup(self, levels, times, curves, looping, allowRetrigger=False):
        levels, times, curves = (np.ascontiguousarray(a, np.float64) for a in (levels, times, curves))
        if not (levels.size == times.size + 1 and levels.size == curves.size + 1):
            return False                               # H:2395-2398
        st = np.zeros((times.size, 6))
        rc = lib().mxg_envgen_stages_host(levels.size, levels.ctypes.data, times.ctypes.data, curves.ctypes.data,
                                          st.ctypes.data)
        if rc < 0:
            return False
        self.host_stages, self.stages = st, DeviceBuffer.from_numpy(st)
        self.loop, self.retrigger = bool(looping), bool(allowRetrigger)
        self._arm()
        return True

    def setupAR(self, attack, release): return self.setup([0, 1, 0], [attack, release], [1, 1], False, False)

    def setupASR(self, attack, release):
        return self.setup([0, 1, 1, 0], [attack, self.HOLD, release], [1, 1, 1], False, False)

    def setupADSR(self, attack, decay, sustain, release):
        return self.setup([0, 1, sustain, sustain, 0], [attack, decay, self.HOLD, release], [1, 1, 1, 1], False, False)

    def setRetrigger(self, val): self.retrigger = bool(val)
    def setLoop(self, val): self.loop = bool(val)

    def play(self, trigger, out=None):
        """trigger: [N][V] per voice, or [N] shared."""
        if not (isinstance(trigger, DeviceBuffer) or hasattr(trigger, "data_ptr")):
            trigger = DeviceBuffer.from_numpy(np.ascontiguousarray(trigger, np.float64))
        tpv = len(trigger.shape) == 2
        N = trigger.shape[0]
        out = self._out(N, out)
        check(lib().mxg_envgen_render(self.V, N, _ptr(trigger), int(tpv), self.stages.ptr, self.host_stages.shape[0],
                                      int(self.loop), int(self.retrigger), self.dstate.ptr, self.istate.ptr, _ptr(out),
                                      self.stream), "mxg_envgen_render")
        self._keep = trigger
        return out


class maxiSamplerBank(_Bank):
    """NS x maxiSampler (L/maxiSynths.h:137-187) of `voices` slots each, all playing one sample.  The control
    methods mirror the reference (they edit host copies of the slot state between renders); play(N) renders N
    calls of maxiSampler::play() for every sampler."""

    def __init__(self, samplers, voices=32, stream=None):
        super().__init__(samplers * voices, stream)
        self.NS, self.voices = int(samplers), int(voices)
        self.sample = maxiSampleBank(1, stream)
        V = self.V
        self.sustain = True
        self.currentVoice = np.zeros(self.NS, np.int64)
        self.pitch = np.zeros(V)                      # ctor, maxiSynths.cpp:262-283
        self.gain = np.zeros(V)                       # envOutGain (uninitialised in the reference: 0 here)
        self.env = maxiEnvBank(V, stream)
        self.env.setAttack(0); self.env.setDecay(1); self.env.setSustain(1.0); self.env.setRelease(2000)
        self.position = np.zeros(V)
        self.trigger_state = np.zeros(V, np.int32)
        self.outhold = np.zeros(V)
        self._state_dirty = True

    # -- control side (host), maxiSynths.cpp:303-491 ------------------------------------------------------
    def load(self, fileName):
        ok = self.sample.load(fileName)
        self.position[:] = float(self.sample.getLength()) if ok else self.position
        self._state_dirty = True
        return ok

    def setSample(self, samples):
        self.sample.setSample(samples)
        self.position[:] = self.sample.getLength() - 1.0      # maxiSample::setSample (H:677)
        self._state_dirty = True

    def _slots(self, setall):
        if setall:
            return np.arange(self.V)
        return np.arange(self.NS) * self.voices + self.currentVoice

    def _pull(self):
        if not self._state_dirty:
            self.position, self.trigger_state, self.outhold = self._dpos.numpy(), self._dtrig.numpy(), self._dout.numpy()

    def setPitch(self, pitch, setall=False):
        self.pitch[self._slots(setall)] = pitch

    def midiNoteOn(self, pitch, velocity, setall=False):
        sl = self._slots(setall)
        self.pitch[sl] = pitch
        if not setall:
            self.gain[sl] = velocity / 128

    def midiNoteOff(self, pitch, velocity=0):
        self._pull()
        self.trigger_state[self.pitch == pitch] = 0
        self._state_dirty = True

    def trigger(self):
        self._pull()
        sl = self._slots(False)
        self.trigger_state[sl] = 1
        self.position[sl] = 0.0
        self.currentVoice = (self.currentVoice + 1) % self.voices
        self._state_dirty = True

    def play(self, N, want_outputs=False):
        V = self.V
        if self._state_dirty:
            self._dpos = DeviceBuffer.from_numpy(self.position)
            self._dtrig = DeviceBuffer.from_numpy(self.trigger_state)
            self._dout = DeviceBuffer.from_numpy(self.outhold)
            self._state_dirty = False
        freq = np.zeros(V)
        check(lib().mxg_sampler_freq_host(V, self.pitch.ctypes.data, self.sample.getLength(), freq.ctypes.data),
              "mxg_sampler_freq_host")
        dfreq, dgain = DeviceBuffer.from_numpy(freq), DeviceBuffer.from_numpy(self.gain)
        dpar, dhold = self.env._params()
        mix = DeviceBuffer((N, self.NS), zero=False)
        self.outputs = DeviceBuffer((N, V), zero=False) if want_outputs else None
        check(lib().mxg_sampler_render(V, N, self.voices, int(self.sustain), self.sample.d_samples, self.sample.getLength(),
                                       dfreq.ptr, dgain.ptr, dpar.ptr, dhold.ptr, self._dpos.ptr, self._dtrig.ptr,
                                       self._dout.ptr, self.env.dstate.ptr, self.env.istate.ptr, mix.ptr,
                                       _ptr(self.outputs), self.stream), "mxg_sampler_render")
        self._keep = (dfreq, dgain)
        return mix


class _Filter2Bank(_Bank):
    KIND = 0

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.state = DeviceBuffer((3, self.V))
        self.coef = None

    def _play(self, x, out):
        N = x.shape[0]
        out = self._out(N, out)
        check(lib().mxg_filter2_render(self.KIND, self.V, N, _ptr(x), self.coef.ptr, self.state.ptr, _ptr(out),
                                       self.stream), "mxg_filter2_render")
        return out


class maxiDCBlockerBank(_Filter2Bank):
    """V x maxiDCBlocker (H:1255-1267)."""
    KIND = 0

    def play(self, x, R, out=None):
        self.coef = _as_dev(R, self.V)
        return self._play(x, out)


class maxiSVFBank(_Filter2Bank):
    """V x maxiSVF (H:1281-1338); the ctor's setParams(1000, 1) (H:1284)."""
    KIND = 1

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.freq = np.full(self.V, 1000.0)
        self.res = np.full(self.V, 1.0)
        self._host = None

    def setCutoff(self, cutoff):
        self.freq = np.ascontiguousarray(np.broadcast_to(np.asarray(cutoff, np.float64), (self.V,)))
        self._host = None

    def setResonance(self, q):
        self.res = np.ascontiguousarray(np.broadcast_to(np.asarray(q, np.float64), (self.V,)))
        self._host = None

    def coefficients(self):
        if self._host is None:
            c = np.zeros((5, self.V))
            check(lib().mxg_svf_coeffs_host(self.V, self.freq.ctypes.data, self.res.ctypes.data, c.ctypes.data),
                  "mxg_svf_coeffs_host")
            self._host = c
        return self._host

    def play(self, w, lpmix, bpmix, hpmix, notchmix, out=None):
        mix = np.stack([np.broadcast_to(np.asarray(m, np.float64), (self.V,)) for m in (lpmix, bpmix, hpmix, notchmix)])
        self.coef = DeviceBuffer.from_numpy(np.concatenate([self.coefficients(), mix]))
        return self._play(w, out)


class maxiBiquadBank(_Filter2Bank):
    """V x maxiBiquad (H:1343-1486)."""
    KIND = 2
    LOWPASS, HIGHPASS, BANDPASS, NOTCH, PEAK, LOWSHELF, HIGHSHELF = range(7)

    def set(self, filtType, cutoff, Q, peakGain):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(filtType, np.int32), (self.V,)))
        cu, q, g = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), (self.V,))) for a in (cutoff, Q, peakGain))
        c = np.zeros((5, self.V))
        check(lib().mxg_biquad_coeffs_host(self.V, t.ctypes.data, cu.ctypes.data, q.ctypes.data, g.ctypes.data,
                                           c.ctypes.data), "mxg_biquad_coeffs_host")
        self.host_coef = c
        self.coef = DeviceBuffer.from_numpy(c)

    def play(self, x, out=None):
        return self._play(x, out)


class maxiEnvBank(_Bank):
    """V x maxiEnv (H:888-932).  No constructor in the reference: all state starts at zero
    (static-storage objects), holdtime defaults to 1 (H:915)."""

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.par = np.zeros((4, self.V))  # attack, decay, sustain, release
        self.holdtime = np.ones(self.V, np.int64)
        self.dstate = DeviceBuffer((2, self.V))            # amplitude, output
        self.istate = DeviceBuffer((6, self.V), np.int64)  # holdcount + 5 phase flags
        self._dirty = True

    def _coeff(self, which, ms):
        ms = np.broadcast_to(np.asarray(ms, np.float64), (self.V,))
        f = lib().mxg_env_coeff_host
        cache = {}
        return np.array([cache.setdefault(m, f(which, m)) for m in ms.tolist()])

    def setAttack(self, ms): self.par[0] = self._coeff(0, ms); self._dirty = True      # C:1480-1482
    def setAttackMS(self, ms): self.par[0] = self._coeff(3, ms); self._dirty = True    # C:1486-1488
    def setDecay(self, ms): self.par[1] = self._coeff(1, ms); self._dirty = True       # C:1475-1477
    def setSustain(self, level): self.par[2] = np.asarray(level, np.float64); self._dirty = True
    def setRelease(self, ms): self.par[3] = self._coeff(2, ms); self._dirty = True     # C:1470-1472

    def _params(self):
        if self._dirty:
            self._dpar = DeviceBuffer.from_numpy(self.par)
            self._dhold = DeviceBuffer.from_numpy(self.holdtime)
            self._dirty = False
        return self._dpar, self._dhold

    def render(self, mode, x, trigger, N, out=None):
        dpar, dhold = self._params()
        trig = trigger
        tpv = 0
        if not (isinstance(trigger, DeviceBuffer) or hasattr(trigger, "data_ptr")):
            t = np.ascontiguousarray(trigger, np.int32)
            tpv = 1 if t.ndim == 2 else 0
            trig = DeviceBuffer.from_numpy(t)
        else:
            tpv = 1 if len(trigger.shape) == 2 else 0
        out = self._out(N, out)
        check(lib().mxg_env_render(mode, self.V, N, _ptr(x), _ptr(trig), tpv, dpar.ptr, dhold.ptr,
                                   self.dstate.ptr, self.istate.ptr, _ptr(out), self.stream),
              "mxg_env_render")
        self._keep = trig
        return out

    def adsr(self, x, trigger, N, **kw): return self.render(0, x, trigger, N, **kw)
    def ar(self, x, trigger, N, **kw): return self.render(1, x, trigger, N, **kw)


class maxiVoiceBank(_Bank):
    """V fused subtractive voices: maxiOsc::saw -> maxiFilter::lores -> maxiEnv::adsr
    (config 3; the per-voice body of 14.monosynth / 15.polysynth), one kernel, one store/sample."""

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.env = maxiEnvBank(voices, stream)
        self.osc_state = DeviceBuffer((2, self.V))
        self.flt_state = DeviceBuffer((5, self.V))

    def render(self, mode, freq, cutoff, resonance, trigger, N, out=None):
        f = _as_dev(freq, self.V)
        cu = np.broadcast_to(np.asarray(cutoff, np.float64), (self.V,))
        rs = np.broadcast_to(np.asarray(resonance, np.float64), (self.V,))
        coef = None
        if mode == 0:
            coef = DeviceBuffer.from_numpy(filter_coeffs(0, cu, rs))
        dcu, drs = DeviceBuffer.from_numpy(cu), DeviceBuffer.from_numpy(rs)
        dpar, dhold = self.env._params()
        trig = trigger
        if not (isinstance(trigger, DeviceBuffer) or hasattr(trigger, "data_ptr")):
            trig = DeviceBuffer.from_numpy(np.ascontiguousarray(trigger, np.int32))
        tpv = 1 if len(trig.shape) == 2 else 0
        out = self._out(N, out)
        check(lib().mxg_voice_render(mode, self.V, N, _ptr(f), dcu.ptr, drs.ptr, _ptr(coef), _ptr(trig),
                                     tpv, dpar.ptr, dhold.ptr, self.osc_state.ptr, self.flt_state.ptr,
                                     self.env.dstate.ptr, self.env.istate.ptr, _ptr(out), self.stream),
              "mxg_voice_render")
        self._keep = (f, dcu, drs, coef, trig)
        return out

    def render_mix(self, mode, freq, cutoff, resonance, trigger, pan, N, out=None, store=True, mix=None, rows=None):
        """render() + the fused maxiMix::stereo mixdown of the bank in the same kernel (mxg_voice_render_mix).
        Returns (out [N,V] or None, mix [N,2]).  rows: a device pointer / buffer [mxg_osc_mix_groups(V)][N][2] (e.g. a grouped
        mix queue's slot) -- then the per-workgroup rows are left there (mxg_voice_render_mix_rows); returns (out, None)."""
        f, pn = _as_dev(freq, self.V), _as_dev(pan, self.V)
        cu = np.broadcast_to(np.asarray(cutoff, np.float64), (self.V,))
        rs = np.broadcast_to(np.asarray(resonance, np.float64), (self.V,))
        coef = DeviceBuffer.from_numpy(filter_coeffs(0, cu, rs)) if mode == 0 else None
        dcu, drs = DeviceBuffer.from_numpy(cu), DeviceBuffer.from_numpy(rs)
        dpar, dhold = self.env._params()
        trig = trigger
        if not (isinstance(trigger, DeviceBuffer) or hasattr(trigger, "data_ptr")):
            trig = DeviceBuffer.from_numpy(np.ascontiguousarray(trigger, np.int32))
        tpv = 1 if len(trig.shape) == 2 else 0
        out = self._out(N, out) if store else None
        args = (mode, self.V, N, _ptr(f), dcu.ptr, drs.ptr, _ptr(coef), _ptr(trig), tpv, dpar.ptr, dhold.ptr, self.osc_state.ptr,
                self.flt_state.ptr, self.env.dstate.ptr, self.env.istate.ptr, _ptr(out), _ptr(pn))
        self._keep = (f, dcu, drs, coef, trig, pn)
        if rows is not None:
            check(lib().mxg_voice_render_mix_rows(*args, _ptr(rows), self.stream), "mxg_voice_render_mix_rows")
            return out, None
        mix = mix if mix is not None else DeviceBuffer((N, 2), np.float64, zero=False)
        check(lib().mxg_voice_render_mix(*args, _ptr(mix), self.stream), "mxg_voice_render_mix")
        return out, mix


class maxiMixBank(_Bank):
    """maxiMix::stereo over a bank + the mixdown over voices (C:503-509)."""

    def stereo(self, x, pan, out=None):
        N = x.shape[0]
        p = _as_dev(pan, self.V)
        out = out if out is not None else DeviceBuffer((N, 2), np.float64, zero=False)
        check(lib().mxg_mix_stereo(self.V, N, _ptr(x), _ptr(p), _ptr(out), self.stream),
              "mxg_mix_stereo")
        self._keep = p
        return out

    def bus(self, channels, x, px, py=None, pz=None, out=None, bus=None):
        """stereo (2) / quad (4, C:512-522) / ambisonic (8, C:525-541): mix [N][channels]; `bus`, if a
        DeviceBuffer [N][channels][V], also receives the per-voice two/four/eight signals."""
        N = x.shape[0]
        dx = _as_dev(px, self.V)
        dy = None if py is None else _as_dev(py, self.V)
        dz = None if pz is None else _as_dev(pz, self.V)
        out = out if out is not None else DeviceBuffer((N, channels), np.float64, zero=False)
        check(lib().mxg_mix_bus(channels, self.V, N, _ptr(x), _ptr(dx), _ptr(dy), _ptr(dz), _ptr(bus),
                                _ptr(out), self.stream), "mxg_mix_bus")
        self._keep = (dx, dy, dz)
        return out

    def quad(self, x, px, py, **kw): return self.bus(4, x, px, py, **kw)
    def ambisonic(self, x, px, py, pz, **kw): return self.bus(8, x, px, py, pz, **kw)


class maxiDelaylineBank(_Bank):
    """V x maxiDelayline (H:266-284).  The ring is `cap` slots per voice (slot-major on device)
    instead of the reference's fixed 88200*8; `size` <= cap."""

    def __init__(self, voices, cap, stream=None):
        super().__init__(voices, stream)
        self.cap = int(cap)
        self.memory = DeviceBuffer((self.cap, self.V))    # ctor memset 0, C:415-417
        self.phase = DeviceBuffer(self.V, np.int32)       # static-storage objects start at 0

    def _render(self, mode, x, size, feedback, position, out):
        N = x.shape[0]
        sz = _as_dev(size, self.V, np.int32)
        fb = _as_dev(feedback, self.V)
        ps = None if position is None else _as_dev(position, self.V, np.int32)
        out = self._out(N, out)
        check(lib().mxg_delay_render(mode, self.V, N, _ptr(x), _ptr(sz), _ptr(fb), _ptr(ps),
                                     self.memory.ptr, self.cap, self.phase.ptr, _ptr(out), self.stream),
              "mxg_delay_render")
        self._keep = (sz, fb, ps)
        return out

    def dl(self, x, size, feedback, out=None):
        return self._render(0, x, size, feedback, None, out)

    def dlFromPosition(self, x, size, feedback, position, out=None):
        return self._render(1, x, size, feedback, position, out)


SAMPLE_MODES = {"play": 0, "playOnce": 1, "playLoop": 2, "playUntil": 3, "playAtSpeed": 4,
                "playOnceAtSpeed": 5, "playUntilAtSpeed": 6, "play4": 7, "playAtSpeedBetweenPoints": 8,
                # trigger-driven (mxg_sample_render_trig)
                "playOnZX": 9, "playOnZXAtSpeed": 10, "playOnZXAtSpeedFromOffset": 11,
                "playOnZXAtSpeedBetweenPoints": 12, "loopSetPosOnZX": 13, "playWithPhasor": 14}


class maxiSampleBank(_Bank):
    """V play heads over one maxiSample (H:602-783): the bank shares the sample data the way
    maxiGrains alias one maxiSample (L/maxiGrains.h:162)."""

    def __init__(self, voices, stream=None):
        super().__init__(voices, stream)
        self.d_samples = None
        self.length = 0
        self.mySampleRate = int(maxiSettings.sampleRate)  # ctor, C:546
        self.position = DeviceBuffer(self.V)
        # maxiTrigger zxTrig (H:593-594: previousValue = 1, firstTrigger = 1), phasorPrev/First (H:731-732)
        self.zx_prev = DeviceBuffer.from_numpy(np.ones(self.V))
        self.zx_first = DeviceBuffer.from_numpy(np.ones(self.V, np.int32))
        self.phasor_prev = DeviceBuffer(self.V)
        self.phasor_first = DeviceBuffer.from_numpy(np.ones(self.V, np.int32))

    def setSample(self, samples):
        """maxiSample::setSample (H:670-678): copies the data, mySampleRate=44100, position=len-1."""
        a = np.ascontiguousarray(samples, np.float64)
        self.clear()
        p = lib().mxg_sample_upload(a.ctypes.data, a.size)
        if not p:
            raise MemoryError(lib().mxg_last_error().decode())
        self.d_samples, self.length = p, a.size
        self.mySampleRate = 44100
        self.position.upload(np.full(self.V, a.size - 1.0))

    def load(self, fileName, channel=0):
        """maxiSample::load (C:605-692): 16-bit PCM WAV -> device amplitudes (de-interleave + /32767.0 on
        the device).  Returns False if the file cannot be read (the reference's bool).  position = size."""
        n = ctypes.c_size_t(0)
        hdr = np.zeros(8, np.int32)
        p = lib().mxg_sample_load_wav(os.fsencode(fileName), int(channel), ctypes.byref(n), hdr.ctypes.data)
        if not p:
            return False
        self.clear()
        self.d_samples, self.length = p, n.value
        self.wav_header = hdr
        self.mySampleRate = int(hdr[4])
        self.position.upload(np.full(self.V, float(n.value)))   # C:681
        return True

    def save(self, fileName):
        """maxiSample::save (C:698-725) with the header fields of the loaded file (or a mono 16-bit default)."""
        hdr = getattr(self, "wav_header", None)
        if hdr is None:
            hdr = np.array([36 + 2 * self.length, 16, 1, 1, self.mySampleRate, 2 * self.mySampleRate, 2, 16], np.int32)
        check(lib().mxg_sample_save_wav(os.fsencode(fileName), self.d_samples, self.length, hdr.ctypes.data,
                                        self.stream), "mxg_sample_save_wav")
        return True

    def amplitudes(self):
        """The device sample buffer as a numpy array (for tests)."""
        out = np.empty(self.length)
        check(lib().mxg_memcpy_d2h(out.ctypes.data, self.d_samples, out.nbytes, self.stream), "mxg_memcpy_d2h")
        return out

    def setSampleAndRate(self, samples, sampleRate):
        self.setSample(samples)
        self.mySampleRate = int(sampleRate)

    def trigger(self):
        """maxiSample::trigger (C:597-600)."""
        self.position.upload(np.zeros(self.V))

    def setPosition(self, newPos):
        """maxiSample::setPosition (C:749-751): clamp(newPos,0,1)*length."""
        p = np.clip(np.broadcast_to(np.asarray(newPos, np.float64), (self.V,)), 0.0, 1.0) * self.length
        self.position.upload(p)

    def getLength(self):
        return self.length

    def clear(self):
        if self.d_samples:
            lib().mxg_sample_free(self.d_samples)
            self.d_samples = None

    def render(self, mode, N, a=None, start=None, end=None, out=None, per_sample=False):
        m = SAMPLE_MODES[mode] if isinstance(mode, str) else int(mode)
        if per_sample and not (isinstance(a, DeviceBuffer) or hasattr(a, "data_ptr")):
            a = DeviceBuffer.from_numpy(np.asarray(a, np.float64).reshape(N, self.V))
        da = None if a is None else (a if per_sample else _as_dev(a, self.V))
        ds = None if start is None else _as_dev(start, self.V)
        de = None if end is None else _as_dev(end, self.V)
        out = self._out(N, out)
        check(lib().mxg_sample_render(m, self.V, N, self.d_samples, self.length, self.mySampleRate,
                                      _ptr(da), int(per_sample), _ptr(ds), _ptr(de), self.position.ptr,
                                      _ptr(out), self.stream), "mxg_sample_render")
        self._keep = (da, ds, de)
        return out

    def render_trig(self, mode, trig, a=None, p0=None, p1=None, out=None, per_sample=False):
        """Modes 9-14: `trig` is the per-sample [N][V] trigger (or phasor) signal."""
        m = SAMPLE_MODES[mode] if isinstance(mode, str) else int(mode)
        if not (isinstance(trig, DeviceBuffer) or hasattr(trig, "data_ptr")):
            trig = DeviceBuffer.from_numpy(np.ascontiguousarray(trig, np.float64))
        N = trig.shape[0]
        if per_sample and not (isinstance(a, DeviceBuffer) or hasattr(a, "data_ptr")):
            a = DeviceBuffer.from_numpy(np.asarray(a, np.float64).reshape(N, self.V))
        da = None if a is None else (a if per_sample else _as_dev(a, self.V))
        d0 = None if p0 is None else _as_dev(p0, self.V)
        d1 = None if p1 is None else _as_dev(p1, self.V)
        tprev, tfirst = (self.phasor_prev, self.phasor_first) if m == 14 else (self.zx_prev, self.zx_first)
        out = self._out(N, out)
        check(lib().mxg_sample_render_trig(m, self.V, N, self.d_samples, self.length, self.mySampleRate,
                                           _ptr(trig), _ptr(da), int(per_sample), _ptr(d0), _ptr(d1),
                                           self.position.ptr, tprev.ptr, tfirst.ptr, _ptr(out), self.stream),
              "mxg_sample_render_trig")
        self._keep = (trig, da, d0, d1)
        return out

    def playOnZX(self, trig, **kw): return self.render_trig("playOnZX", trig, **kw)
    def playOnZXAtSpeed(self, trig, speed, **kw): return self.render_trig("playOnZXAtSpeed", trig, a=speed, **kw)

    def playOnZXAtSpeedFromOffset(self, trig, speed, offset, **kw):
        return self.render_trig("playOnZXAtSpeedFromOffset", trig, a=speed, p0=offset, **kw)

    def playOnZXAtSpeedBetweenPoints(self, trig, speed, offset, length, **kw):
        return self.render_trig("playOnZXAtSpeedBetweenPoints", trig, a=speed, p0=offset, p1=length, **kw)

    def loopSetPosOnZX(self, trig, pos, **kw): return self.render_trig("loopSetPosOnZX", trig, p0=pos, **kw)
    def playWithPhasor(self, pha, **kw): return self.render_trig("playWithPhasor", pha, **kw)

    def play(self, N, **kw): return self.render("play", N, **kw)
    def playOnce(self, N, **kw): return self.render("playOnce", N, **kw)
    def playLoop(self, start, end, N, **kw): return self.render("playLoop", N, start=start, end=end, **kw)
    def playUntil(self, end, N, **kw): return self.render("playUntil", N, end=end, **kw)
    def playAtSpeed(self, speed, N, **kw): return self.render("playAtSpeed", N, a=speed, **kw)
    def playOnceAtSpeed(self, speed, N, **kw): return self.render("playOnceAtSpeed", N, a=speed, **kw)

    def playUntilAtSpeed(self, end, speed, N, **kw):
        return self.render("playUntilAtSpeed", N, a=speed, end=end, **kw)

    def play4(self, frequency, start, end, N, **kw):
        return self.render("play4", N, a=frequency, start=start, end=end, **kw)

    def playAtSpeedBetweenPoints(self, frequency, start, end, N, **kw):
        return self.render("playAtSpeedBetweenPoints", N, a=frequency, start=start, end=end, **kw)

    def __del__(self):
        try:
            self.clear()
        except Exception:
            pass
