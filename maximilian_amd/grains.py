"""Host-side mirror of maxiTimeStretch / maxiStretch (src/libs/maxiGrains.h) as banks of streams."""
import numpy as np

from ._lib import check, lib
from .banks import DeviceBuffer, _Bank, _as_dev, _ptr

WINDOWS = {"hann": 0, "hamming": 1, "cosine": 2, "rect": 3, "triangle": 4, "triangleNZ": 5,
           "blackmanHarris": 6, "blackmanNutall": 7, "gaussian": 8}


class _GranularBank(_Bank):
    MODE = 0

    def __init__(self, streams, sample_bank, window="hann", stream=None):
        """`sample_bank`: a maxiSampleBank holding the shared maxiSample (setSample done)."""
        super().__init__(streams, stream)
        self.sample = sample_bank
        self.window_kind = WINDOWS[window] if isinstance(window, str) else int(window)
        self.state = DeviceBuffer((4, self.V))      # position, looper, randomOffset, rand cursor
        self.grains = DeviceBuffer((4, 8, self.V))  # live grains, creation order
        self._plans = {}

    def setPosition(self, pos):
        """maxiTimeStretch::setPosition (L/maxiGrains.h:335-338): clamp(pos*len, 0, len-1)."""
        st = self.state.numpy()
        n = self.sample.getLength()
        st[0] = np.clip(np.broadcast_to(np.asarray(pos, np.float64), (self.V,)) * n, 0, n - 1)
        self.state.upload(st)

    def getPosition(self):
        return self.state.numpy()[0]

    def getNormalisedPosition(self):
        return self.getPosition() / float(self.sample.getLength())

    def _plan(self, grainLength):
        key = (float(grainLength), self.sample.mySampleRate)
        if key not in self._plans:
            p = lib().mxg_grain_plan_create(self.window_kind, float(grainLength), self.sample.mySampleRate)
            if not p:
                raise ValueError(lib().mxg_last_error().decode())
            self._plans[key] = p
        return self._plans[key]

    def _render(self, a, b, grainLength, overlaps, posMod, N, rnd, out, mode=None):
        mode = self.MODE if mode is None else mode
        if mode == 2:   # per-sample position signal [N][S]
            da = a if (isinstance(a, DeviceBuffer) or hasattr(a, "data_ptr")) else \
                DeviceBuffer.from_numpy(np.ascontiguousarray(a, np.float64).reshape(N, self.V))
        else:
            da = _as_dev(a, self.V)
        db = None if b is None else _as_dev(b, self.V)
        dp = None if posMod is None else _as_dev(posMod, self.V)
        dr, R = None, 0
        if rnd is not None:
            r = np.ascontiguousarray(rnd, np.int32).reshape(self.V, -1)
            dr, R = DeviceBuffer.from_numpy(r), r.shape[1]
        out = self._out(N, out)
        check(lib().mxg_granular_render(self._plan(grainLength), mode, self.V, N, self.sample.d_samples,
                                        self.sample.getLength(), int(overlaps), _ptr(da), _ptr(db), _ptr(dp),
                                        _ptr(dr), R, self.state.ptr, self.grains.ptr, _ptr(out), self.stream),
              "mxg_granular_render")
        return out

    def close(self):
        for p in self._plans.values():
            lib().mxg_grain_plan_destroy(p)
        self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class maxiTimeStretchBank(_GranularBank):
    """S x maxiTimeStretch<F> (L/maxiGrains.h:287-368)."""
    MODE = 0

    def play(self, speed, grainLength, overlaps, N, posMod=None, rnd=None, out=None):
        return self._render(speed, None, grainLength, overlaps, posMod, N, rnd, out)

    def playAtPosition(self, pos, grainLength, overlaps, out=None):
        """maxiTimeStretch::playAtPosition (L/maxiGrains.h:359-367): `pos` is the [N][S] per-sample
        normalised position signal the caller iterates itself."""
        return self._render(pos, None, grainLength, overlaps, None, pos.shape[0], None, out, mode=2)


class maxiPitchShiftBank(_GranularBank):
    """S x maxiPitchShift<F> (L/maxiGrains.h:374-432).  state[1] is the member `cycles`."""
    MODE = 3

    def play(self, speed, grainLength, overlaps, N, posMod=None, out=None):
        return self._render(speed, None, grainLength, overlaps, posMod, N, None, out)


class maxiStretchBank(_GranularBank):
    """S x maxiStretch<F> (L/maxiGrains.h:436-542), default loop points."""
    MODE = 1

    def play(self, pitchstretch, timestretch, grainLength, overlaps, N, posMod=None, rnd=None, out=None):
        return self._render(pitchstretch, timestretch, grainLength, overlaps, posMod, N, rnd, out)
