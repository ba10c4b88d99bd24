#!/usr/bin/env python3
"""tools/bench_speedplayer.py -- a bare loop of block-constant playAtSpeed launches (65 536 voices x 512) for the counter passes of
tools/pmc_sq.sh; SMP_SPLIT in the environment sets the knob."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
v = np.arange(V)
for k in ("split", "ring"):
    if os.environ.get("SMP_" + k.upper()):
        L.mxg_tune(("smp_" + k).encode(), int(os.environ["SMP_" + k.upper()]))
rng = np.random.default_rng(1)
sb = mx.maxiSampleBank(V); sb.setSample(rng.uniform(-1, 1, 441000)); sb.setPosition(v / V)
dsp = mx.DeviceBuffer.from_numpy(0.5 + (v % 97) / 96.0)
out = mx.DeviceBuffer((B, V), zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
reps = int(os.environ.get("REPS", "100")) * 20
call = lambda: L.mxg_sample_render(4, V, B, sb.d_samples, sb.length, 44100, dsp.ptr, 0, None, None, sb.position.ptr, out.ptr, None)
for _ in range(20): call()
L.mxg_event_record(e0, None)
for _ in range(reps): call()
L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
print("playAtSpeed %.1f us per block" % (ms.value / reps * 1e3))
