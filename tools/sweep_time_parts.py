#!/usr/bin/env python3
"""tools/sweep_time_parts.py -- osc_split / smp_split / smp_roles / smp_nt sweeps at 65 536 voices x 512 (HIP events)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
v = np.arange(V)
D = mx.DeviceBuffer.from_numpy
freq = D(20 + v * 0.30517578125); p1 = D(np.full(V, 0.25)); p2 = D(np.full(V, 0.75))
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
out = mx.DeviceBuffer((B, V), zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=300):
    for _ in range(150): fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
for split in (0, 4):
    L.mxg_tune(b"osc_split", split)
    row = []
    for name in ("sinewave", "coswave", "sinebuf4", "sawn", "sinebuf"):
        wf = mx.OSC_WAVEFORMS[name]
        row.append("%s %.1f" % (name, timed(lambda: L.mxg_osc_render(wf, V, B, freq.ptr, 0, p1.ptr, p2.ptr, phase.ptr, hold.ptr, out.ptr, None))))
    print("osc_split", split, " | ".join(row), flush=True)
L.mxg_tune(b"osc_split", 0)
rng = np.random.default_rng(1)
sb = mx.maxiSampleBank(V); sb.setSample(rng.uniform(-1, 1, 441000)); sb.setPosition(v / V)
dsp = D(0.5 + (v % 97) / 96.0)
for pipe in (1, 0):
    L.mxg_tune(b"smp_pipe", pipe)
    for split in (0, 1, 2, 3, 4, 6, 8):
        L.mxg_tune(b"smp_split", split)
        row = []
        for mode in (4, 5):
            sb.setPosition(v / V * 0.5)
            row.append("mode%d %.1f" % (mode, timed(lambda: L.mxg_sample_render(mode, V, B, sb.d_samples, sb.length, 44100, dsp.ptr, 0, None, None, sb.position.ptr, out.ptr, None), 100)))
        print("smp_pipe", pipe, "smp_split", split, " | ".join(row), flush=True)
