#!/usr/bin/env python3
"""tools/bench_spectral.py -- config 4: maxiFFT + maxiMFCC over N x 1024-point frames on one GPU.
Reports per-kernel time (HIP events on the launch stream), frames/s, Msamples/s and algorithmic GB/s."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

N = int(os.environ.get("FRAMES", 1 << 20))
REPS = int(os.environ.get("REPS", 10))
L = mx.lib()
mx._lib.check(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
rng = np.random.default_rng(0x4D415849)
chunk = 1 << 22
sig = mx.DeviceBuffer((N * 1024,), np.float32, zero=False)
n = np.arange(chunk)
for o in range(0, N * 1024, chunk):  # config-4 signal, generated in chunks
    k = (o + n) // 1024
    x = (0.4 * np.sin(2 * np.pi * 220 * (o + n) / 44100) + 0.3 * np.sin(2 * np.pi * (440 + 0.01 * k) * (o + n) / 44100)
         + 0.1 * rng.uniform(-1, 1, chunk)).astype(np.float32)
    L.mxg_memcpy_h2d(sig.ptr + 4 * o, x.ctypes.data, x.nbytes, None)
f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
mags = mx.DeviceBuffer((N, 512), np.float32, zero=False)
phases = mx.DeviceBuffer((N, 512), np.float32, zero=False)
out = mx.DeviceBuffer((N, 13), np.float64, zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()


def timed(fn, reps=REPS):
    fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps):
        fn()
    L.mxg_event_record(e1, None)
    L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / reps


res = {"frames": N}
t = timed(lambda: L.mxg_fft_batch(f.plan, sig.ptr, 1024, N, None, None, mags.ptr, None, None))
res["fft_mags_ms"] = t; res["fft_mags_GBs"] = N * (4096 + 2048) / t / 1e6
t = timed(lambda: L.mxg_fft_batch(f.plan, sig.ptr, 1024, N, None, None, mags.ptr, phases.ptr, None))
res["fft_mags_phases_ms"] = t; res["fft_mags_phases_GBs"] = N * (4096 + 4096) / t / 1e6
L.mxg_tune(b"fft_generic", 1)
t = timed(lambda: L.mxg_fft_batch(f.plan, sig.ptr, 1024, N, None, None, mags.ptr, None, None), 3)
res["fft_generic_mags_ms"] = t
L.mxg_tune(b"fft_generic", 0)
L.mxg_fft_batch(f.plan, sig.ptr, 1024, N, None, None, mags.ptr, None, None)
t = timed(lambda: L.mxg_mfcc_batch(m.plan, mags.ptr, 512, N, None, None, out.ptr, 0, None))
res["mfcc_exact_ms"] = t; res["mfcc_exact_GBs"] = N * (215 * 4 + 104) / t / 1e6
t = timed(lambda: L.mxg_mfcc_batch(m.plan, mags.ptr, 512, N, None, None, out.ptr, 1, None), 3)
res["mfcc_mfma_ms"] = t; res["mfcc_mfma_TFLOPs"] = N * 2.0 * 216 * 48 / t / 1e9


def both():
    L.mxg_fft_batch(f.plan, sig.ptr, 1024, N, None, None, mags.ptr, None, None)
    L.mxg_mfcc_batch(m.plan, mags.ptr, 512, N, None, None, out.ptr, 0, None)


t = timed(both)
res["fft_plus_mfcc_ms"] = t
res["frames_per_s"] = N / t * 1e3
res["Msamples_per_s_in"] = N * 1024 / t / 1e3
res["algorithmic_GBs_4200B_per_frame"] = N * 4200 / t / 1e6
print(json.dumps(res, indent=1))
