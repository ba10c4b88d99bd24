"""Host-side mirror of maxiFFT / maxiMFCC (src/libs/maxiFFT.h, maxiMFCC.h) in batch form.

`maxiFFT.setup(fftSize, hopSize, windowSize)` keeps the reference's signature; instead of one
`process(sample)` call per audio sample, `process_signal(signal)` analyses every frame the
reference would have produced for that stream (same hop buffer semantics: the buffer starts
with windowSize-hopSize zeros, L/maxiFFT.cpp:56) in one launch.  `maxiMFCC.mfcc(mags)` takes the
[nframes, bins] magnitudes and returns [nframes, numCoeffs].
"""
import numpy as np

from ._lib import check, lib
from .banks import DeviceBuffer, _ptr


def frames_in_stream(nsamples, hopSize, windowSize):
    """How many times maxiFFT::process() reports a new frame over `nsamples` samples."""
    return 0 if nsamples < hopSize else (nsamples - hopSize) // hopSize + 1


def padded_stream(signal, hopSize, windowSize):
    """The signal as the hop buffer sees it: (windowSize-hopSize) zeros, then the samples."""
    signal = np.ascontiguousarray(signal, np.float32)
    return np.concatenate([np.zeros(windowSize - hopSize, np.float32), signal])


class maxiFFT:
    """maxiFFT (L/maxiFFT.h:45-113) over whole signals / frame batches."""
    NO_POLAR_CONVERSION, WITH_POLAR_CONVERSION = 0, 1

    def __init__(self, stream=None):
        self.plan = None
        self.stream = stream

    def setup(self, fftSize=1024, hopSize=512, windowSize=0):
        """L/maxiFFT.cpp:45-60."""
        self.close()
        check(lib().mxg_init(-1), "mxg_init")
        p = lib().mxg_fft_plan_create(fftSize, hopSize, windowSize)
        if not p:
            raise ValueError(lib().mxg_last_error().decode())
        self.plan = p
        self.fftSize, self.hopSize = fftSize, hopSize
        self.windowSize = max(windowSize, fftSize)  # :48
        self.bins = fftSize // 2

    def getNumBins(self): return self.bins
    def getFFTSize(self): return self.fftSize
    def getHopSize(self): return self.hopSize
    def getWindowSize(self): return self.windowSize

    def process_frames(self, d_signal, frame_stride, nframes, mode=1, want_complex=False):
        """Transform nframes frames starting every frame_stride samples in a device signal."""
        real = imag = mags = phases = None
        if mode == self.WITH_POLAR_CONVERSION:
            mags = DeviceBuffer((nframes, self.bins), np.float32, zero=False)
            phases = DeviceBuffer((nframes, self.bins), np.float32, zero=False)
        if want_complex or mode == self.NO_POLAR_CONVERSION:
            real = DeviceBuffer((nframes, self.bins), np.float32, zero=False)
            imag = DeviceBuffer((nframes, self.bins), np.float32, zero=False)
        check(lib().mxg_fft_batch(self.plan, _ptr(d_signal), frame_stride, nframes, _ptr(real), _ptr(imag),
                                  _ptr(mags), _ptr(phases), self.stream), "mxg_fft_batch")
        self.real, self.imag, self.magnitudes, self.phases = real, imag, mags, phases
        return nframes

    def process_signal(self, signal, mode=1, want_complex=False):
        """Every frame process() would report while consuming `signal` sample by sample."""
        n = frames_in_stream(len(signal), self.hopSize, self.windowSize)
        buf = DeviceBuffer.from_numpy(padded_stream(signal, self.hopSize, self.windowSize))
        self._keep = buf
        if n == 0:
            self.real = self.imag = self.magnitudes = self.phases = None
            return 0
        return self.process_frames(buf, self.hopSize, n, mode, want_complex)

    def getMagnitudes(self): return self.magnitudes
    def getPhases(self): return self.phases
    def getReal(self): return self.real
    def getImag(self): return self.imag

    def _features(self, mags, db=False, flatness=False, centroid=False):
        mags = self.magnitudes if mags is None else mags
        n = mags.shape[0]
        d = DeviceBuffer((n, self.bins), np.float32, zero=False) if db else None
        f = DeviceBuffer(n, np.float32, zero=False) if flatness else None
        c = DeviceBuffer(n, np.float32, zero=False) if centroid else None
        check(lib().mxg_fft_features(self.plan, _ptr(mags), n, _ptr(d), _ptr(f), _ptr(c), self.stream),
              "mxg_fft_features")
        return d, f, c

    def magsToDB(self, mags=None):
        """maxiFFT::magsToDB (L/maxiFFT.cpp:101-111) for every frame: [nframes, bins] fp32."""
        return self._features(mags, db=True)[0]

    def spectralFlatness(self, mags=None):
        """maxiFFT::spectralFlatness (L/maxiFFT.cpp:113-123) per frame."""
        return self._features(mags, flatness=True)[1]

    def spectralCentroid(self, mags=None):
        """maxiFFT::spectralCentroid (L/maxiFFT.cpp:125-132) per frame."""
        return self._features(mags, centroid=True)[2]

    def close(self):
        if self.plan:
            lib().mxg_fft_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class maxiMFCC:
    """maxiMFCCAnalyser<double> (L/maxiMFCC.h:41-211) over batches of spectra."""
    EXACT, MFMA = 0, 1

    def __init__(self, stream=None):
        self.plan = None
        self.stream = stream

    def setup(self, numBins, numFilters, numCoeffs, minFreq, maxFreq):
        """L/maxiMFCC.h:56-75."""
        self.close()
        p = lib().mxg_mfcc_plan_create(numBins, numFilters, numCoeffs, minFreq, maxFreq)
        if not p:
            raise ValueError(lib().mxg_last_error().decode())
        self.plan = p
        self.numBins, self.numFilters, self.numCoeffs = numBins, numFilters, numCoeffs

    def tables(self):
        W = np.zeros(self.numFilters * self.numBins)
        D = np.zeros(self.numCoeffs * self.numFilters)
        used = lib().mxg_mfcc_plan_tables(self.plan, W.ctypes.data, D.ctypes.data)
        return W, D, used

    def mfcc(self, mags, nframes=None, mag_stride=None, method=0, want_bands=False):
        nframes = mags.shape[0] if nframes is None else nframes
        mag_stride = self.numBins if mag_stride is None else mag_stride
        out = DeviceBuffer((nframes, self.numCoeffs), np.float64, zero=False)
        raw = bands = None
        if want_bands:
            raw = DeviceBuffer((nframes, self.numFilters), np.float64, zero=False)
            bands = DeviceBuffer((nframes, self.numFilters), np.float64, zero=False)
        check(lib().mxg_mfcc_batch(self.plan, _ptr(mags), mag_stride, nframes, _ptr(raw), _ptr(bands),
                                   _ptr(out), method, self.stream), "mxg_mfcc_batch")
        self.melraw, self.melBands = raw, bands
        return out

    def mfcc_of_frames(self, fft, signal, nframes, frame_stride=None, want_mags=False, want_bands=False):
        """maxiFFT(1024) magnitudes -> mfcc() for `nframes` frames of `signal` in ONE kernel (mxg_fft_mfcc_batch):
        the per-frame body of mfcctest.cpp:21-32.  `fft` is a maxiFFT set up for 1024 points; returns the [nframes,
        numCoeffs] DeviceBuffer; .mags / .melraw / .melBands hold the optional outputs."""
        frame_stride = fft.fftSize if frame_stride is None else frame_stride
        out = DeviceBuffer((nframes, self.numCoeffs), np.float64, zero=False)
        mags = DeviceBuffer((nframes, self.numBins), np.float32, zero=False) if want_mags else None
        raw = bands = None
        if want_bands:
            raw = DeviceBuffer((nframes, self.numFilters), np.float64, zero=False)
            bands = DeviceBuffer((nframes, self.numFilters), np.float64, zero=False)
        check(lib().mxg_fft_mfcc_batch(fft.plan, self.plan, _ptr(signal), frame_stride, nframes, _ptr(mags), _ptr(raw),
                                       _ptr(bands), _ptr(out), self.stream), "mxg_fft_mfcc_batch")
        self.mags, self.melraw, self.melBands = mags, raw, bands
        return out

    def close(self):
        if self.plan:
            lib().mxg_mfcc_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class maxiIFFT:
    """maxiIFFT (L/maxiFFT.h:117-156, SPECTRUM mode) over batches of spectra: `process_frames(mags, phases)`
    returns the nframes*hopSize samples `process()` would return when called hopSize times per spectrum;
    the overlap-add buffer is carried between calls."""
    SPECTRUM, COMPLEX = 0, 1

    def __init__(self, stream=None):
        self.plan = None
        self.stream = stream

    def setup(self, fftSize=1024, hopSize=512, windowSize=0):
        self.close()
        p = lib().mxg_ifft_plan_create(int(fftSize), int(hopSize), int(windowSize))
        if not p:
            raise ValueError(lib().mxg_last_error().decode())
        self.plan = p
        self.fftSize, self.hopSize, self.bins = fftSize, hopSize, fftSize // 2
        self.windowSize = windowSize if windowSize else fftSize   # L/maxiFFT.cpp:143
        self.buffer = DeviceBuffer(fftSize, np.float32)           # member `buffer`, zero-filled by setup()
        self.ifftOut = None

    def getNumBins(self): return self.bins

    def process_frames(self, mags, phases, keep_ifft=False):
        dm = mags if isinstance(mags, DeviceBuffer) or hasattr(mags, "data_ptr") else \
            DeviceBuffer.from_numpy(np.ascontiguousarray(mags, np.float32))
        dp = phases if isinstance(phases, DeviceBuffer) or hasattr(phases, "data_ptr") else \
            DeviceBuffer.from_numpy(np.ascontiguousarray(phases, np.float32))
        n = dm.shape[0]
        out = DeviceBuffer(n * self.hopSize, np.float32, zero=False)
        self.ifftOut = DeviceBuffer((n, self.fftSize), np.float32, zero=False) if keep_ifft else None
        check(lib().mxg_ifft_batch(self.plan, _ptr(dm), _ptr(dp), n, self.buffer.ptr, _ptr(out), _ptr(self.ifftOut),
                                   self.stream), "mxg_ifft_batch")
        self._keep = (dm, dp)
        return out

    def close(self):
        if self.plan:
            lib().mxg_ifft_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class maxiConvolve:
    """maxiConvolve (L/maxiConvolve.cpp): partitioned convolution.  `setup(amplitudes, fftsize, hopsize)` takes the
    impulse as the loaded maxiSample holds it (the reference loads it from a file name); `play(x, mode)` renders what
    play(w) returns for every sample of x (a whole number of fftsize blocks), state carried between calls.
    mode 0 = as the reference computes (its COMPLEX-mode maxiIFFT never sees the sums: silence), 1 = as intended."""

    def __init__(self, stream=None):
        self.h = None
        self.stream = stream

    def setup(self, amplitudes, fftsize=1024, hopsize=256, position0=None):
        self.close()
        a = np.ascontiguousarray(amplitudes, np.float64)
        pos = float(a.size if position0 is None else position0)   # load()/read() leave the play head at size (C:681)
        h = lib().mxg_convolve_create(a.ctypes.data, a.size, pos, fftsize, hopsize)
        if not h:
            raise ValueError(lib().mxg_last_error().decode())
        self.h, self.fftsize, self.hopsize, self.bins = h, fftsize, hopsize, fftsize // 2
        self.frames = lib().mxg_convolve_frames(h)

    def impulse(self):
        r = np.zeros((self.frames, self.bins), np.float32)
        i = np.zeros((self.frames, self.bins), np.float32)
        check(lib().mxg_convolve_impulse(self.h, r.ctypes.data, i.ctypes.data), "mxg_convolve_impulse")
        return r, i

    def play(self, x, mode=0, out=None):
        if not (isinstance(x, DeviceBuffer) or hasattr(x, "data_ptr")):
            x = DeviceBuffer.from_numpy(np.ascontiguousarray(x, np.float32))
        n = int(np.prod(x.shape))
        assert n % self.fftsize == 0, "a whole number of fftsize blocks"
        out = out if out is not None else DeviceBuffer(n, np.float32, zero=False)
        check(lib().mxg_convolve_play(self.h, _ptr(x), n // self.fftsize, _ptr(out), mode, self.stream), "mxg_convolve_play")
        self._keep = x
        return out

    def reset(self):
        check(lib().mxg_convolve_reset(self.h), "mxg_convolve_reset")

    def close(self):
        if getattr(self, "h", None):
            lib().mxg_convolve_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
