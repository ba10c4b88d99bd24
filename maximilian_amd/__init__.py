"""maximilian_amd -- MI355X (gfx950) voice-bank renderer behind the Maximilian class API.

The product is the C-ABI library `libmaxigpu.so` (include/maxigpu.h) built from
maximilian_amd/csrc/*.hip, and the C++ host facade include/maximilian_bank.hpp.  This Python
package is the thin host-side mirror used by the tests and bench.py: device buffers, and
`*Bank` classes whose methods carry the reference's names (maxiOsc::sinebuf -> maxiOscBank.sinebuf).
"""
from ._lib import LIB_PATH, MaxiGpuError, calib, lib  # noqa: F401
from .banks import (DeviceBuffer, maxiSettings, maxiOscBank, maxiFilterBank, maxiEnvBank,  # noqa: F401
                    maxiVoiceBank, maxiMixBank, maxiDelaylineBank, maxiSampleBank, maxiDCBlockerBank,
                    maxiSVFBank, maxiBiquadBank, maxiEnvGenBank, maxiSamplerBank, OSC_WAVEFORMS,
                    FILTER_KINDS, SAMPLE_MODES)
from .spectral import maxiConvolve, maxiFFT, maxiIFFT, maxiMFCC, frames_in_stream, padded_stream  # noqa: F401
from .grains import maxiTimeStretchBank, maxiStretchBank, maxiPitchShiftBank, WINDOWS  # noqa: F401
