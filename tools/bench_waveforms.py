#!/usr/bin/env python3
"""tools/bench_waveforms.py -- K1 for every maxiOsc waveform at config-2 size (65 536 voices x 512), HIP events."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
v = np.arange(V)
freq = mx.DeviceBuffer.from_numpy(20 + v * 0.30517578125)
p1 = mx.DeviceBuffer.from_numpy(np.full(V, 0.25)); p2 = mx.DeviceBuffer.from_numpy(np.full(V, 0.75))
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
out = mx.DeviceBuffer((B, V), zero=False)
rnd = mx.DeviceBuffer.from_numpy(np.random.default_rng(1).integers(0, 2**31 - 1, (B, V)).astype(np.int32))
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=300):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:      # clock ramp: an idle MI355X needs continuous work to reach its sustained clocks
        for _ in range(20): fn()
        L.mxg_sync()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
res = {}
for name, wf in mx.OSC_WAVEFORMS.items():
    us = timed(lambda: L.mxg_osc_render(wf, V, B, freq.ptr, 0, p1.ptr, p2.ptr, phase.ptr, hold.ptr, out.ptr, None))
    res[name] = {"us_per_block": round(us, 2), "GBs_algorithmic": round(8.047 * V * B / us / 1e3, 1)}
us = timed(lambda: L.mxg_osc_noise(V, B, rnd.ptr, hold.ptr, out.ptr, None))
res["noise (4 B rand in + 8 B out)"] = {"us_per_block": round(us, 2), "GBs_algorithmic": round(12.0 * V * B / us / 1e3, 1)}
for k, d in res.items():
    print("%-32s %7.2f us  %7.1f GB/s  %.2f of 8 TB/s" % (k, d["us_per_block"], d["GBs_algorithmic"], d["GBs_algorithmic"] / 8000))
