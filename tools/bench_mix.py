#!/usr/bin/env python3
"""tools/bench_mix.py -- K1 vs K1m (fused render+mix, with/without the per-voice store) vs K1+K3."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init")
V, B = 65536, 512
wf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * 0.30517578125)
pan = mx.DeviceBuffer.from_numpy(np.arange(V) / (V - 1.0))
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
out = mx.DeviceBuffer((B, V)); mix = mx.DeviceBuffer((B, 2))
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=50):
    for _ in range(5): fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
variants = {
 "K1 render": lambda: L.mxg_osc_render(wf, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, out.ptr, None),
 "K1m render+mix (store)": lambda: L.mxg_osc_render_mix(wf, V, B, freq.ptr, None, None, phase.ptr, hold.ptr, out.ptr, pan.ptr, mix.ptr, None),
 "K1m mix only (no store)": lambda: L.mxg_osc_render_mix(wf, V, B, freq.ptr, None, None, phase.ptr, hold.ptr, None, pan.ptr, mix.ptr, None),
 "K3 mix_stereo alone": lambda: L.mxg_mix_stereo(V, B, out.ptr, pan.ptr, mix.ptr, None),
}
for rnd in range(3):
    for k, f in variants.items():
        print("%-28s %.1f us" % (k, timed(f)))
