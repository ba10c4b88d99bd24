#!/usr/bin/env python3
"""tools/bench_ifft.py -- maxiIFFT batch (K6i + overlap-add): NF spectra of 1024-point frames, hop 256/512/1024."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
NF = int(os.environ.get("FRAMES", 262144))
rng = np.random.default_rng(1)
mags = mx.DeviceBuffer.from_numpy(np.abs(rng.normal(0, 1, (NF, 512))).astype(np.float32))
phases = mx.DeviceBuffer.from_numpy(rng.uniform(-np.pi, np.pi, (NF, 512)).astype(np.float32))
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
for hop in (256, 512, 1024):
    f = mx.maxiIFFT(); f.setup(1024, hop, 1024)
    out = mx.DeviceBuffer(NF * hop, np.float32, zero=False)
    call = lambda: L.mxg_ifft_batch(f.plan, mags.ptr, phases.ptr, NF, f.buffer.ptr, out.ptr, None, None)
    for _ in range(3): call()
    L.mxg_event_record(e0, None)
    for _ in range(5): call()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    t = ms.value / 5
    print("hop %4d: %.3f ms per %d frames = %.1f M frames/s (%.0f GB/s of 4096 B in + %d B out per frame)" % (
        hop, t, NF, NF / t / 1e3, NF * (4096 + 4 * hop) / t / 1e6, 4 * hop))
