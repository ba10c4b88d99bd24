#!/usr/bin/env python3
"""tools/bench_banks.py -- per-bank kernel times at V=65536, B=512 (HIP events, library stream)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
rng = np.random.default_rng(1)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=30):
    for _ in range(3): fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
x = mx.DeviceBuffer.from_numpy(rng.uniform(-1, 1, (B, V)))
out = mx.DeviceBuffer((B, V), zero=False)
v = np.arange(V)
fb = mx.maxiFilterBank(V)
cut = 200 + 4 * np.minimum(20 + v * 0.305, 5000.0); res = 1.0 + (v % 16)
print("filter lores (hoisted)  %.1f us  (wall per call INCLUDING the host-libm coefficients of 65 536 voices + upload; kernel alone: see the stats table)" % timed(lambda: fb.render("lores", x, cut, res, out=out)))
print("filter lopass           %.1f us" % timed(lambda: fb.render("lopass", x, np.full(V, 0.3), out=out)))
eb = mx.maxiEnvBank(V); eb.setAttack(10); eb.setDecay(100); eb.setSustain(0.5); eb.setRelease(500)
trig = mx.DeviceBuffer.from_numpy(((np.arange(B) % 300) < 150).astype(np.int32))
print("env adsr (gate 150/150) %.1f us  (16 B/sample; attack/decay/release every 300 samples: state-machine path)" % timed(lambda: eb.render(0, x, trig, B, out=out)))
hold = mx.DeviceBuffer.from_numpy(np.ones(B, np.int32))
for _ in range(4): eb.render(0, x, hold, B, out=out)
print("env adsr (sustain)      %.1f us  (gate held: steady-state path)" % timed(lambda: eb.render(0, x, hold, B, out=out)))
db = mx.maxiDelaylineBank(V, 2048)
print("delay dl size 1024      %.1f us  (32 B/sample)" % timed(lambda: db.dl(x, 1024, 0.5, out=out)))
sb = mx.maxiSampleBank(V); sb.setSample(rng.uniform(-1, 1, 441000)); sb.setPosition(v / V)
print("sample playAtSpeed      %.1f us" % timed(lambda: sb.playAtSpeed(0.5 + (v % 97) / 96.0, B, out=out)))
print("sample play             %.1f us  (fractional heads left by playAtSpeed: per-sample gathers)" % timed(lambda: sb.play(B, out=out)))
sb.position.upload(np.floor(v / V * 441000.0))
print("sample play (int heads) %.1f us  (heads on integer positions, as after trigger()/load(): 16-B row loads)" % timed(lambda: sb.play(B, out=out)))
