#!/usr/bin/env python3
"""tools/sweep_grains_general.py -- the general (non-unit-increment) granular render K8a + K8b on the config-5 shape
(2048 streams x 70 560 samples, grainLength 0.05, overlaps 4): maxiStretch and maxiPitchShift against the number of
(stream, chunk) lanes of K8b (knob grain_lanes_k, in units of 1024 lanes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
S, T = 2048, 70560
rng = np.random.default_rng(0x4D415849)
Ls = 4410000; n = np.arange(Ls)
smp = 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) + 0.05 * rng.uniform(-1, 1, Ls)
sb = mx.maxiSampleBank(1); sb.setSample(smp)
speed = 0.25 + 1.5 * (np.arange(S) % 97) / 96
out = mx.DeviceBuffer((T, S), zero=False)
for lk in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]:
    L.mxg_tune(b"grain_lanes_k", lk)
    row = []
    for name in ("stretch", "pitch"):
        bank = (mx.maxiStretchBank if name == "stretch" else mx.maxiPitchShiftBank)(S, sb, "hann")
        best = 1e9
        for r in range(3):
            bank.setPosition(np.arange(S) / S)
            bank.grains.upload(np.zeros((4, 8, S)))
            L.mxg_sync(); t0 = time.perf_counter()
            if name == "stretch":
                bank.play(speed, 0.8, 0.05, 4, T, out=out)
            else:
                bank.play(speed, 0.05, 4, T, out=out)
            L.mxg_sync(); best = min(best, time.perf_counter() - t0)
        row.append("%s %.2f ms (%.0f G grain-samples/s)" % (name, best * 1e3, S * T * 4 / best / 1e9))
    print("grain_lanes_k %4d: %s" % (lk, ", ".join(row)), flush=True)

# per-kernel times of one call each (HIP events inside the library)
import ctypes
for name in ("stretch", "pitch"):
    bank = (mx.maxiStretchBank if name == "stretch" else mx.maxiPitchShiftBank)(S, sb, "hann")
    bank.setPosition(np.arange(S) / S)
    L.mxg_prof_reset(); L.mxg_prof_enable(1)
    if name == "stretch":
        bank.play(speed, 0.8, 0.05, 4, T, out=out)
    else:
        bank.play(speed, 0.05, 4, T, out=out)
    L.mxg_sync(); L.mxg_prof_enable(0)
    parts = []
    for i in range(L.mxg_prof_count()):
        lab, ms, cnt = ctypes.c_char_p(), ctypes.c_double(0), ctypes.c_size_t(0)
        L.mxg_prof_read(i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            parts.append("%s %.3f ms x%d" % (lab.value.decode(), ms.value / cnt.value, cnt.value))
    print(name, "|", "; ".join(parts), flush=True)
