#!/usr/bin/env python3
"""tools/sweep_banks.py -- throughput against bank size V (block = 512 samples) for BASELINE configs 2 and 3:
maxiOsc::sinebuf (K1), maxiOsc::sinewave (K1, fp64-VALU bound), the fused subtractive voice mode A (K2f) and the fused
render+mixdown (K1m).  V = 64 ... 1 048 576, including 131 072 (the >= 1e5-voice target of BASELINE.json).  HIP events
around back-to-back launches; prints a markdown table (profiles/r02_bank_size_curve.md)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
B = 512
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()


def timed(fn, reps):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:       # clock ramp: an idle MI355X needs continuous work to reach its sustained clocks
        for _ in range(10): fn()
        L.mxg_stream_sync(None)
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3


rows = []
for V in [64, 256, 1024, 4096, 16384, 65536, 131072, 262144, 524288, 1048576]:
    v = np.arange(V)
    freq_h = 20 + (v % 65536) * 0.30517578125
    freq = mx.DeviceBuffer.from_numpy(freq_h)
    pan = mx.DeviceBuffer.from_numpy(v / max(V - 1, 1))
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    out = mx.DeviceBuffer((B, V), zero=False)
    mix = mx.DeviceBuffer((B, 2), zero=False)
    reps = int(max(30, min(2000, 4e8 / (V * B))))
    r = {"V": V}
    r["sinebuf"] = timed(lambda: L.mxg_osc_render(8, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, out.ptr, None), reps)
    r["sinewave"] = timed(lambda: L.mxg_osc_render(0, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, out.ptr, None), reps)
    r["sinebuf+mix"] = timed(lambda: L.mxg_osc_render_mix(8, V, B, freq.ptr, None, None, phase.ptr, hold.ptr, out.ptr, pan.ptr, mix.ptr, None), reps)
    vb = mx.maxiVoiceBank(V)
    vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
    f3 = np.minimum(freq_h, 5000.0)
    vb.render(0, f3, 200 + 4 * f3, 1.0 + (v % 16), np.ones(1, np.int32), 1, out=mx.DeviceBuffer((1, V)))
    vf, vcu, vrs, vcoef, _ = vb._keep
    vpar, vhold = vb.env._params()
    gate = mx.DeviceBuffer.from_numpy(np.ones(B, np.int32))   # sustain: the steady state of the envelope
    r["voice A (sustain)"] = timed(lambda: L.mxg_voice_render(0, V, B, vf.ptr, vcu.ptr, vrs.ptr, vcoef.ptr, gate.ptr, 0, vpar.ptr, vhold.ptr,
                                                              vb.osc_state.ptr, vb.flt_state.ptr, vb.env.dstate.ptr, vb.env.istate.ptr,
                                                              out.ptr, None), reps)
    rows.append(r)
    del out, vb

keys = ["sinebuf", "sinewave", "sinebuf+mix", "voice A (sustain)"]
bps = {"sinebuf": 8.0 + 24.0 / B, "sinewave": 8.0 + 24.0 / B, "sinebuf+mix": 8.0 + 40.0 / B, "voice A (sustain)": 8.0 + 176.0 / B}
print("# Throughput against bank size (MI355X, block = 512 samples, fp64 out[n][v] stored)\n")
print("`python tools/sweep_banks.py`: microseconds per block, G samples/s, and the fraction of the 8 TB/s HBM peak on the algorithmic")
print("bytes per sample (8.047 / 8.078 / 8.34 B).  Real time at 44.1 kHz needs V x 44 100 samples/s: 2.9 G samples/s at 65 536 voices.\n")
print("| voices | " + " | ".join("%s us / Gs/s / frac" % k for k in keys) + " |")
print("|---|" + "---|" * len(keys))
for r in rows:
    cells = []
    for k in keys:
        us = r[k]
        gs = r["V"] * B / us / 1e3
        cells.append("%.1f / %.1f / %.3f" % (us, gs, gs * bps[k] / 8000.0))
    print("| %d | %s |" % (r["V"], " | ".join(cells)))
