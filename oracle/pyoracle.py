"""oracle/pyoracle.py -- ctypes driver for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(never by maximilian_amd/).  Both libraries export the same mxo_* symbols:

    port()       oracle/liboracle.so        plain-C restatement (oracle/maxi_oracle.c)
    reference()  oracle/_ref/libmaxiref.so  the unmodified reference + oracle/ref_harness.cpp
                                            (prebuilt in the build container; may be absent)
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_size_t, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libmaxiref.so")

_SIGS = {
    "mxo_kind": (c_char_p, []),
    "mxo_settings": (None, [c_size_t, c_size_t, c_size_t]),
    "mxo_sine_table_guard": (c_double, []),
    "mxo_transition_guard": (c_double, []),
    "mxo_osc": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_void_p]),
    "mxo_filter": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_int,
                           c_void_p, c_void_p]),
    "mxo_filter_coeffs": (None, [c_int, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mxo_env": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                        c_void_p, c_void_p, c_void_p]),
    "mxo_env_coeff": (c_double, [c_int, c_double]),
    "mxo_voice": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxo_mix_stereo": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mxo_time_osc": (c_double, [c_int, c_size_t, c_size_t, c_void_p, c_int, c_void_p]),
    "mxo_osc_tables": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # (port only: an extension)
}


def _p(a):
    return None if a is None else a.ctypes.data


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, np.float64)
    return a if shape is None else np.ascontiguousarray(np.broadcast_to(a, shape))


class Oracle:
    """Stateless-call wrapper: state arrays are passed in, updated copies are returned."""

    def __init__(self, path):
        self.path = path
        self.L = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            if not hasattr(self.L, name):
                continue  # later-added families are checked where used
            fn = getattr(self.L, name)
            fn.restype = res
            fn.argtypes = args
        self.kind = self.L.mxo_kind().decode()

    def settings(self, sr=44100, ch=2, buf=1024):
        self.L.mxo_settings(sr, ch, buf)

    # -- maxiOsc ------------------------------------------------------------------------
    def osc(self, wf, freq, N, phase=None, hold=None, p1=None, p2=None, per_sample=False):
        freq = np.ascontiguousarray(freq, np.float64)
        V = freq.shape[-1]
        phase = np.zeros(V) if phase is None else _f64(phase, (V,)).copy()
        hold = np.zeros(V) if hold is None else _f64(hold, (V,)).copy()
        p1 = np.zeros(V) if p1 is None else _f64(p1, (V,))
        p2 = np.zeros(V) if p2 is None else _f64(p2, (V,))
        out = np.empty((N, V))
        rc = self.L.mxo_osc(wf, V, N, _p(freq), int(per_sample), _p(p1), _p(p2), _p(phase), _p(hold),
                            _p(out))
        assert rc == 0
        return out, phase, hold

    def osc_tables(self, freq, tables, N, phase=None, hold=None):
        """EXTENSION (port only): sinebuf with a 514-entry table per voice, tables [V][514]."""
        freq = np.ascontiguousarray(freq, np.float64)
        V = freq.shape[-1]
        tables = _f64(tables, (V, 514))
        phase = np.zeros(V) if phase is None else _f64(phase, (V,)).copy()
        hold = np.zeros(V) if hold is None else _f64(hold, (V,)).copy()
        out = np.empty((N, V))
        rc = self.L.mxo_osc_tables(V, N, _p(freq), _p(tables), _p(phase), _p(hold), _p(out))
        assert rc == 0
        return out, phase, hold

    def sine_table(self):
        """sineBuffer[0..513] (C:63) as the oracle holds it."""
        self.L.mxo_sine_table.restype = ctypes.POINTER(c_double)
        p = self.L.mxo_sine_table()
        return np.array([p[i] for i in range(514)])

    # -- maxiFilter -----------------------------------------------------------------------
    def filter(self, kind, x, cutoff, res=None, state=None, cps=False, rps=False):
        x = _f64(x)
        N, V = x.shape
        cutoff = _f64(cutoff, (N, V) if cps else (V,))
        r = None if res is None else _f64(res, (N, V) if rps else (V,))
        st = np.zeros((5, V)) if state is None else _f64(state).copy()
        out = np.empty((N, V))
        rc = self.L.mxo_filter(kind, V, N, _p(x), _p(cutoff), int(cps), _p(r), int(rps), _p(st), _p(out))
        assert rc == 0
        return out, st

    def filter_coeffs(self, kind, cutoff, res):
        cutoff = _f64(cutoff)
        res = _f64(res, cutoff.shape)
        coef = np.zeros((3, cutoff.size))
        self.L.mxo_filter_coeffs(kind, cutoff.size, _p(cutoff), _p(res), _p(coef))
        return coef

    # -- maxiEnv ----------------------------------------------------------------------------
    def env_coeff(self, which, ms):
        return self.L.mxo_env_coeff(which, float(ms))

    def env(self, mode, x, trig, par, holdtime, dstate=None, istate=None, N=None):
        par = _f64(par)
        V = par.shape[1]
        trig = np.ascontiguousarray(trig, np.int32)
        N = trig.shape[0] if N is None else N
        x = None if x is None else _f64(x, (N, V))
        holdtime = np.ascontiguousarray(np.broadcast_to(np.asarray(holdtime, np.int64), (V,)))
        dst = np.zeros((2, V)) if dstate is None else _f64(dstate).copy()
        ist = np.zeros((6, V), np.int64) if istate is None else np.ascontiguousarray(istate, np.int64).copy()
        out = np.empty((N, V))
        rc = self.L.mxo_env(mode, V, N, _p(x), _p(trig), int(trig.ndim == 2), _p(par), _p(holdtime),
                            _p(dst), _p(ist), _p(out))
        assert rc == 0
        return out, dst, ist

    # -- fused subtractive voice ---------------------------------------------------------------
    def voice(self, mode, freq, cutoff, res, trig, par, holdtime, ost=None, fst=None, dstate=None,
              istate=None):
        freq = _f64(freq)
        V = freq.size
        cutoff, res, par = _f64(cutoff, (V,)), _f64(res, (V,)), _f64(par)
        trig = np.ascontiguousarray(trig, np.int32)
        N = trig.shape[0]
        holdtime = np.ascontiguousarray(np.broadcast_to(np.asarray(holdtime, np.int64), (V,)))
        ost = np.zeros((2, V)) if ost is None else _f64(ost).copy()
        fst = np.zeros((5, V)) if fst is None else _f64(fst).copy()
        dst = np.zeros((2, V)) if dstate is None else _f64(dstate).copy()
        ist = np.zeros((6, V), np.int64) if istate is None else np.ascontiguousarray(istate, np.int64).copy()
        out = np.empty((N, V))
        rc = self.L.mxo_voice(mode, V, N, _p(freq), _p(cutoff), _p(res), _p(trig), int(trig.ndim == 2),
                              _p(par), _p(holdtime), _p(ost), _p(fst), _p(dst), _p(ist), _p(out))
        assert rc == 0
        return out, ost, fst, dst, ist

    # -- maxiMix ------------------------------------------------------------------------------
    def mix_stereo(self, x, pan):
        x = _f64(x)
        N, V = x.shape
        pan = _f64(pan, (V,))
        mix = np.empty((N, 2))
        rc = self.L.mxo_mix_stereo(V, N, _p(x), _p(pan), _p(mix))
        assert rc == 0
        return mix

    def mix_bus(self, channels, x, px, py=None, pz=None, want_bus=False):
        """maxiMix::stereo/quad/ambisonic (C:503-541): returns (mix [N][C], bus [N][C][V] or None)."""
        x = _f64(x)
        N, V = x.shape
        px = _f64(px, (V,))
        py = None if py is None else _f64(py, (V,))
        pz = None if pz is None else _f64(pz, (V,))
        mix = np.empty((N, channels))
        bus = np.empty((N, channels, V)) if want_bus else None
        fn = self.L.mxo_mix_bus
        fn.restype = c_int
        fn.argtypes = [c_int, c_size_t, c_size_t] + [c_void_p] * 6
        rc = fn(channels, V, N, _p(x), _p(px), _p(py), _p(pz), _p(bus), _p(mix))
        assert rc == 0, rc
        return mix, bus

    def noise(self, seed, V, N):
        """maxiOsc::noise (C:214-220): (rand() draws int32 [N][V], out [N][V]) for srand(seed)."""
        rnd = np.empty((N, V), np.int32)
        out = np.empty((N, V))
        fn = self.L.mxo_noise
        fn.restype = c_int
        fn.argtypes = [ctypes.c_uint, c_size_t, c_size_t, c_void_p, c_void_p]
        rc = fn(seed, V, N, _p(rnd), _p(out))
        assert rc == 0, rc
        return rnd, out

    def filter2(self, kind, x, par, st=None):
        """maxiDCBlocker (0; par [1][V] R), maxiSVF (1; par [6][V] cutoff,res,lp,bp,hp,notch), maxiBiquad (2; par [4][V]
        type,cutoff,Q,peakGain).  Returns (out, st [3][V], coef [5][V])."""
        x = _f64(x)
        N, V = x.shape
        par = _f64(par).reshape(-1, V)
        st = np.zeros((3, V)) if st is None else _f64(st).copy()
        coef = np.zeros((5, V))
        out = np.empty((N, V))
        fn = self.L.mxo_filter2
        fn.restype = c_int
        fn.argtypes = [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        rc = fn(kind, V, N, _p(x), _p(par), _p(st), _p(coef), _p(out))
        assert rc == 0, rc
        return out, st, coef

    ENVGEN_HOLD = -46692.0

    @staticmethod
    def envgen_fresh(V):
        """State of V freshly set-up maxiEnvGen objects: dst [5][V], ist [7][V] (WAITING, detectors 1.0 / first)."""
        dst = np.zeros((5, V))
        dst[2:5] = 1.0
        ist = np.zeros((7, V), np.int64)
        ist[4:7] = 1
        return dst, ist

    def envgen(self, trig, levels, times, curves, loop=False, retrigger=False, dst=None, ist=None, V=None):
        """maxiEnvGen (H:2268-2547).  trig [N][V] (per voice) or [N] with V given.  Returns (out, dst, ist, stages)."""
        trig = _f64(trig)
        tpv = trig.ndim == 2
        N = trig.shape[0]
        V = trig.shape[1] if tpv else int(V)
        levels, times, curves = _f64(levels), _f64(times), _f64(curves)
        d0, i0 = self.envgen_fresh(V)
        dst = d0 if dst is None else _f64(dst).copy()
        ist = i0 if ist is None else np.ascontiguousarray(ist, np.int64).copy()
        stages = np.zeros((levels.size - 1, 6))
        out = np.empty((N, V))
        fn = self.L.mxo_envgen
        fn.restype = c_int
        fn.argtypes = [c_size_t, c_size_t, c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_int,
                       c_void_p, c_void_p, c_void_p, c_void_p]
        rc = fn(V, N, _p(trig), int(tpv), levels.size, _p(levels), _p(times), _p(curves), int(loop), int(retrigger),
                _p(dst), _p(ist), _p(stages), _p(out))
        assert rc == 0, rc
        return out, dst, ist, stages

    # -- maxiDelayline ---------------------------------------------------------------------------
    def delay(self, mode, x, size, feedback, cap, position=None, mem=None, phase=None):
        x = _f64(x)
        N, V = x.shape
        size = np.ascontiguousarray(np.broadcast_to(np.asarray(size, np.int32), (V,)))
        feedback = _f64(feedback, (V,))
        position = np.zeros(V, np.int32) if position is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(position, np.int32), (V,)))
        mem = np.zeros((cap, V)) if mem is None else _f64(mem).copy()
        phase = np.zeros(V, np.int32) if phase is None else np.ascontiguousarray(phase, np.int32).copy()
        out = np.empty((N, V))
        fn = self.L.mxo_delay
        fn.restype = c_int
        fn.argtypes = [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_size_t, c_void_p, c_void_p]
        rc = fn(mode, V, N, _p(x), _p(size), _p(feedback), _p(position), _p(mem), cap, _p(phase), _p(out))
        assert rc == 0, rc
        return out, mem, phase

    # -- maxiSample play family ------------------------------------------------------------------
    @staticmethod
    def guarded(samples):
        """[0, samples..., 0, 0] and the offset-1 view the mxo_sample/mxg_sample contract wants."""
        g = np.zeros(len(samples) + 3)
        g[1:-2] = samples
        return g

    def sample(self, mode, samples, N, position, a=None, start=None, end=None, aps=False,
               mySampleRate=44100):
        g = self.guarded(np.asarray(samples, np.float64))
        position = _f64(position).copy()
        V = position.size
        a = None if a is None else _f64(a, (N, V) if aps else (V,))
        start = None if start is None else _f64(start, (V,))
        end = None if end is None else _f64(end, (V,))
        out = np.empty((N, V))
        fn = self.L.mxo_sample
        fn.restype = c_int
        fn.argtypes = [c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_int, c_void_p,
                       c_void_p, c_void_p, c_void_p]
        rc = fn(mode, V, N, g.ctypes.data + 8, len(samples), mySampleRate, _p(a), int(aps), _p(start),
                _p(end), _p(position), _p(out))
        assert rc == 0, rc
        return out, position

    def sample_zx(self, mode, samples, trig, position, a=None, aps=False, p0=None, p1=None,
                  zx_prev=None, zx_first=None, mySampleRate=44100):
        """playOnZX (0), playOnZXAtSpeed (1), ...FromOffset (2), ...BetweenPoints (3), loopSetPosOnZX (4)
        C:1006-1042.  Returns (out, position, zx_prev, zx_first)."""
        g = self.guarded(np.asarray(samples, np.float64))
        trig = _f64(trig)
        N, V = trig.shape
        position = _f64(position, (V,)).copy()
        a = None if a is None else _f64(a, (N, V) if aps else (V,))
        p0 = None if p0 is None else _f64(p0, (V,))
        p1 = None if p1 is None else _f64(p1, (V,))
        zx_prev = np.ones(V) if zx_prev is None else _f64(zx_prev, (V,)).copy()
        zx_first = np.ones(V, np.int32) if zx_first is None else np.ascontiguousarray(zx_first, np.int32).copy()
        out = np.empty((N, V))
        fn = self.L.mxo_sample_zx
        fn.restype = c_int
        fn.argtypes = [c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_int,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        rc = fn(mode, V, N, g.ctypes.data + 8, len(samples), mySampleRate, _p(trig), _p(a), int(aps), _p(p0),
                _p(p1), _p(position), _p(zx_prev), _p(zx_first), _p(out))
        assert rc == 0, rc
        return out, position, zx_prev, zx_first

    def sample_phasor(self, samples, pha, phasor_prev=None, phasor_first=None):
        """maxiSample::playWithPhasor C:753-816.  Returns (out, phasorPrev, phasorFirst)."""
        g = self.guarded(np.asarray(samples, np.float64))
        pha = _f64(pha)
        N, V = pha.shape
        phasor_prev = np.zeros(V) if phasor_prev is None else _f64(phasor_prev, (V,)).copy()
        phasor_first = np.ones(V, np.int32) if phasor_first is None else \
            np.ascontiguousarray(phasor_first, np.int32).copy()
        out = np.empty((N, V))
        fn = self.L.mxo_sample_phasor
        fn.restype = c_int
        fn.argtypes = [c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]
        rc = fn(V, N, g.ctypes.data + 8, len(samples), _p(pha), _p(phasor_prev), _p(phasor_first), _p(out))
        assert rc == 0, rc
        return out, phasor_prev, phasor_first

    # -- maxiFFT streamed over a signal -------------------------------------------------------------
    def fft_stream(self, signal, fftSize=1024, hopSize=512, windowSize=0, want=("real", "imag", "mags", "phases")):
        signal = np.ascontiguousarray(signal, np.float32)
        win = max(windowSize, fftSize)
        max_frames = max(0, (signal.size + (win - hopSize) - win) // hopSize + 1) if signal.size else 0
        bins = fftSize // 2
        bufs = {k: (np.zeros((max_frames, bins), np.float32) if k in want else None)
                for k in ("real", "imag", "mags", "phases")}
        fn = self.L.mxo_fft_stream
        fn.restype = ctypes.c_long
        fn.argtypes = [c_void_p, c_size_t, c_int, c_int, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]
        n = fn(_p(signal), signal.size, fftSize, hopSize, windowSize, max_frames, _p(bufs["real"]),
               _p(bufs["imag"]), _p(bufs["mags"]), _p(bufs["phases"]))
        assert n >= 0, n
        return {k: (v[:n] if v is not None else None) for k, v in bufs.items()}

    def fft_to_db(self, mags):
        mags = np.ascontiguousarray(mags, np.float32)
        out = np.empty_like(mags)
        fn = self.L.mxo_fft_to_db
        fn.restype = None
        fn.argtypes = [c_void_p, c_void_p, c_size_t]
        fn(_p(mags), _p(out), mags.size)
        return out

    def fft_features(self, mags, fftSize):
        """maxiFFT::spectralFlatness / spectralCentroid (L/maxiFFT.cpp:113-132) per frame of [bins] magnitudes."""
        mags = np.ascontiguousarray(mags, np.float32)
        nframes = mags.shape[0]
        flat = np.empty(nframes, np.float32)
        cen = np.empty(nframes, np.float32)
        fn = self.L.mxo_fft_features
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_size_t, c_int, c_void_p, c_void_p]
        rc = fn(_p(mags), nframes, fftSize, _p(flat), _p(cen))
        assert rc == 0, rc
        return flat, cen

    def ifft_stream(self, mags, phases, fftSize=1024, hopSize=512, windowSize=0, buffer=None):
        """maxiIFFT (L/maxiFFT.cpp:140-192, SPECTRUM mode).  Returns (signal [nframes*hop], ifftOut [nframes][fftSize],
        buffer [fftSize])."""
        mags = np.ascontiguousarray(mags, np.float32)
        phases = np.ascontiguousarray(phases, np.float32)
        nframes = mags.shape[0]
        out = np.zeros(nframes * hopSize, np.float32)
        io = np.zeros((nframes, fftSize), np.float32)
        buf = np.zeros(fftSize, np.float32) if buffer is None else np.ascontiguousarray(buffer, np.float32).copy()
        fn = self.L.mxo_ifft_stream
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
        rc = fn(_p(mags), _p(phases), nframes, fftSize, hopSize, windowSize, _p(out), _p(io), _p(buf))
        assert rc == 0, rc
        return out, io, buf

    # -- maxiMFCC ---------------------------------------------------------------------------------------
    def mfcc_tables(self, numBins=512, numFilters=42, numCoeffs=13, minFreq=20.0, maxFreq=20000.0):
        W = np.zeros(numFilters * numBins)
        D = np.zeros(numCoeffs * numFilters)
        fn = self.L.mxo_mfcc_tables
        fn.restype = c_int
        fn.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, c_double, c_double, c_void_p, c_void_p]
        fn(numBins, numFilters, numCoeffs, minFreq, maxFreq, _p(W), _p(D))
        return W, D

    def mfcc(self, mags, numFilters=42, numCoeffs=13, minFreq=20.0, maxFreq=20000.0):
        mags = np.ascontiguousarray(mags, np.float32)
        n, numBins = mags.shape
        mel = np.zeros((n, numFilters))
        out = np.zeros((n, numCoeffs))
        fn = self.L.mxo_mfcc
        fn.restype = c_int
        fn.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, c_double, c_double, c_void_p, c_size_t,
                       c_size_t, c_void_p, c_void_p]
        rc = fn(numBins, numFilters, numCoeffs, minFreq, maxFreq, _p(mags), numBins, n, _p(mel), _p(out))
        assert rc == 0
        return mel, out

    # -- maxiGrains ---------------------------------------------------------------------------------------
    def grain_window(self, kind, length):
        out = np.zeros(length)
        fn = self.L.mxo_grain_window
        fn.restype = c_int
        fn.argtypes = [c_int, ctypes.c_uint, c_void_p]
        rc = fn(kind, length, _p(out))
        assert rc == 0, rc
        return out

    def granular(self, mode, window_kind, samples, T, a, b=None, posMod=None, rnd=None, grainLength=0.05,
                 overlaps=4, mySampleRate=44100, st=None, gst=None):
        """maxiTimeStretch::play (mode 0) / maxiStretch::play (1) / maxiTimeStretch::playAtPosition (2, a =
        per-sample pos [T,S]) / maxiPitchShift::play (3) bank.  Returns (out [T,S], st, gst, rc)."""
        samples = np.asarray(samples, np.float64)
        g = np.concatenate([samples, [0.0]])  # guard element amp[len]
        a = _f64(a)
        S = a.shape[1] if mode == 2 else a.size
        b = None if b is None else _f64(b, (S,))
        posMod = None if posMod is None else _f64(posMod, (S,))
        R = 0
        if rnd is not None:
            rnd = np.ascontiguousarray(rnd, np.int32).reshape(S, -1)
            R = rnd.shape[1]
        st = np.zeros((4, S)) if st is None else _f64(st).copy()
        gst = np.zeros((4, 8, S)) if gst is None else _f64(gst).copy()
        out = np.zeros((T, S))
        fn = self.L.mxo_granular
        fn.restype = c_int
        fn.argtypes = [c_int, c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_double, c_int, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]
        rc = fn(mode, window_kind, S, T, _p(g), samples.size, mySampleRate, grainLength, overlaps, _p(a), _p(b),
                _p(posMod), _p(rnd), R, _p(st), _p(gst), _p(out))
        return out, st, gst, rc

    # -- maxiSample::load / save (16-bit PCM WAV) ---------------------------------------------------
    def wav_load(self, path, channel=0, cap=1 << 22):
        """Returns (amplitudes, hdr[8] int32, position) or None if the file cannot be opened."""
        out = np.zeros(cap)
        hdr = np.zeros(8, np.int32)
        pos = ctypes.c_double(0)
        fn = self.L.mxo_wav_load
        fn.restype = ctypes.c_long
        fn.argtypes = [ctypes.c_char_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]
        n = fn(os.fsencode(path), channel, _p(out), cap, _p(hdr), ctypes.addressof(pos))
        if n < 0:
            return None
        return out[:n].copy(), hdr, pos.value

    def wav_save(self, path, amplitudes, hdr):
        amplitudes = _f64(amplitudes)
        hdr = np.ascontiguousarray(hdr, np.int32)
        fn = self.L.mxo_wav_save
        fn.restype = c_int
        fn.argtypes = [ctypes.c_char_p, c_void_p, c_size_t, c_void_p]
        return fn(os.fsencode(path), _p(amplitudes), amplitudes.size, _p(hdr))

    # -- maxiSampler (L/maxiSynths.h:137-187) -------------------------------------------------------
    def sampler(self, voices, samples, N, pitch, gain, par, holdtime, position, trigger, sustain=True, outhold=None,
                dst=None, ist=None):
        """NS = V/voices samplers.  Returns (mix [N,NS], outputs [N,V], position, trigger, outhold, dst, ist)."""
        g = self.guarded(np.asarray(samples, np.float64))
        pitch = _f64(pitch)
        V = pitch.size
        NS = V // voices
        gain, par = _f64(gain, (V,)), _f64(par)
        holdtime = np.ascontiguousarray(np.broadcast_to(np.asarray(holdtime, np.int64), (V,)))
        position = _f64(position, (V,)).copy()
        trigger = np.ascontiguousarray(np.broadcast_to(np.asarray(trigger, np.int32), (V,))).copy()
        outhold = np.zeros(V) if outhold is None else _f64(outhold).copy()
        dst = np.zeros((2, V)) if dst is None else _f64(dst).copy()
        ist = np.zeros((6, V), np.int64) if ist is None else np.ascontiguousarray(ist, np.int64).copy()
        mix = np.empty((N, NS))
        outputs = np.empty((N, V))
        fn = self.L.mxo_sampler
        fn.restype = c_int
        fn.argtypes = [c_size_t, c_int, c_size_t, c_void_p, c_size_t, c_int] + [c_void_p] * 11
        rc = fn(NS, voices, N, g.ctypes.data + 8, len(samples), int(sustain), _p(pitch), _p(gain), _p(par), _p(holdtime),
                _p(position), _p(trigger), _p(outhold), _p(dst), _p(ist), _p(mix), _p(outputs))
        assert rc == 0, rc
        return mix, outputs, position, trigger, outhold, dst, ist

    # -- CPU baseline timer -----------------------------------------------------------------------
    def time_osc(self, wf, freq, N, threads=1):
        freq = _f64(freq)
        sink = ctypes.c_double(0)
        return self.L.mxo_time_osc(wf, freq.size, N, _p(freq), threads, ctypes.addressof(sink))


    # -- maxiConvolve (L/maxiConvolve.cpp) ------------------------------------------------------------------
    def convolve(self, pcm, x, fftsize=1024, hopsize=256, mode=0):
        """maxiConvolve::setup(16-bit impulse, fftsize, hopsize) then play(x[s]) for every sample.  mode 0 = as the
        reference computes (silence), 1 = sums routed to the inverse transform's inputs.  Returns (out, impReal, impImag)."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        x = np.ascontiguousarray(x, np.float32)
        bins = fftsize // 2
        cap = pcm.size // fftsize + 2
        out = np.zeros(x.size, np.float32)
        ir, ii = np.zeros((cap, bins), np.float32), np.zeros((cap, bins), np.float32)
        fn = self.L.mxo_convolve
        fn.restype = ctypes.c_long
        fn.argtypes = [c_void_p, c_size_t, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p, c_size_t]
        nfr = fn(_p(pcm), pcm.size, fftsize, hopsize, _p(x), x.size, _p(out), mode, _p(ir), _p(ii), cap)
        assert nfr >= 0, nfr
        return out, ir[:nfr], ii[:nfr]

    # -- CPU baselines of BASELINE configs 3/4/5 (bench.py's cpu_baseline leg) ---------------------------
    # With the compiled reference: the threaded timers of oracle/ref_harness.cpp.  The plain-C port has no such
    # entry points; it is timed single-threaded around its bank functions (allocation excluded as far as possible).
    def time_voice(self, mode, freq, cutoff, res, N, threads=1):
        freq, cutoff, res = _f64(freq), _f64(cutoff), _f64(res)
        if hasattr(self.L, "mxo_time_voice"):
            fn = self.L.mxo_time_voice
            fn.restype = c_double
            fn.argtypes = [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
            return fn(mode, freq.size, N, _p(freq), _p(cutoff), _p(res), threads, None)
        import time
        par = np.array([[self.env_coeff(0, 10)], [self.env_coeff(1, 100)], [0.5], [self.env_coeff(2, 500)]]) * np.ones((4, freq.size))
        trig = ((np.arange(N) % 44100) < 22050).astype(np.int32)
        t0 = time.perf_counter()
        self.voice(mode, freq, cutoff, res, trig, par, np.ones(freq.size, np.int64))
        return time.perf_counter() - t0

    def time_spectral(self, signal, threads=1):
        signal = np.ascontiguousarray(signal, np.float32)
        nframes = signal.size // 1024
        if hasattr(self.L, "mxo_time_spectral"):
            fn = self.L.mxo_time_spectral
            fn.restype = c_double
            fn.argtypes = [c_size_t, c_void_p, c_int, c_void_p]
            return fn(nframes, _p(signal), threads, None)
        import time
        t0 = time.perf_counter()
        e = self.fft_stream(signal[:nframes * 1024], 1024, 1024, 1024, want=("mags",))
        self.mfcc(e["mags"])
        return time.perf_counter() - t0

    def time_grains(self, samples, speed, pos01, T, threads=1):
        samples, speed, pos01 = _f64(samples), _f64(speed), _f64(pos01)
        if hasattr(self.L, "mxo_time_grains"):
            fn = self.L.mxo_time_grains
            fn.restype = c_double
            fn.argtypes = [c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p]
            return fn(speed.size, T, _p(samples), samples.size, _p(speed), _p(pos01), threads, None)
        import time
        st = np.zeros((4, speed.size))
        st[0] = np.clip(pos01 * samples.size, 0, samples.size - 1)
        t0 = time.perf_counter()
        self.granular(0, 0, samples, T, speed, st=st)
        return time.perf_counter() - t0


_cache = {}


def port():
    if "port" not in _cache:
        if not os.path.exists(PORT_PATH):
            raise FileNotFoundError("%s missing: run `make -C oracle liboracle.so`" % PORT_PATH)
        _cache["port"] = Oracle(PORT_PATH)
    return _cache["port"]


def have_reference():
    return os.path.exists(REF_PATH)


def reference():
    if "ref" not in _cache:
        _cache["ref"] = Oracle(REF_PATH)
    return _cache["ref"]
