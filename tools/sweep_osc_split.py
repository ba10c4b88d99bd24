#!/usr/bin/env python3
"""tools/sweep_osc_split.py -- maxiOsc::sinewave / coswave at 65 536 voices x 512 against the number of time parts (knob osc_split),
two passes (HIP events after a warm-up of 200 launches)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
v = np.arange(V)
D = mx.DeviceBuffer.from_numpy
freq = D(20 + v * 0.30517578125)
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
out = mx.DeviceBuffer((B, V), zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=400):
    for _ in range(200): fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
for rep in range(2):
    for split in (1, 2, 3, 4, 5, 6, 8):
        L.mxg_tune(b"osc_split", split)
        row = ["%s %.1f" % (n, timed(lambda: L.mxg_osc_render(mx.OSC_WAVEFORMS[n], V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, out.ptr, None))) for n in ("sinewave", "coswave")]
        print("osc_split", split, " | ".join(row), flush=True)
