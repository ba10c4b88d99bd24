#!/usr/bin/env python3
"""tools/sweep_osc.py -- A/B sweep of the K1 launch knobs + HBM write-ceiling calibration.
Interleaved rounds in ONE process (cdna guide 5.4 rule 24); prints median/min per variant."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

L = mx.lib()
CAL = mx.calib()  # measurement probes: libmaxicalib.so
mx._lib.check(L.mxg_init(0), "init")
V, B = 65536, 512
wf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * 0.30517578125)
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
out = mx.DeviceBuffer((B, V), zero=True)
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
import ctypes
ms = ctypes.c_float()


def timed(fn, reps=int(os.environ.get("REPS", "5"))):
    L.mxg_event_record(e0, None)
    for _ in range(reps):
        fn()
    L.mxg_event_record(e1, None)
    L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / reps


nbytes = V * B * 8
variants = {}
for w in (8, 16):
    variants["fill%d" % w] = (lambda w=w: CAL.mxg_calib_fill(out.ptr, nbytes, w, None))
for vpl, nt, blk in itertools.product((1, 2), (0, 1), (64, 128, 256)):
    def f(vpl=vpl, nt=nt, blk=blk):
        L.mxg_tune(b"osc_vpl", vpl); L.mxg_tune(b"osc_nt", nt); L.mxg_tune(b"osc_block", blk)
        L.mxg_osc_render(wf, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, out.ptr, None)
    variants["osc vpl%d nt%d blk%d" % (vpl, nt, blk)] = f

res = {k: [] for k in variants}
for rnd in range(int(os.environ.get("ROUNDS", "8"))):
    for k, f in variants.items():
        t = timed(f)
        if rnd:
            res[k].append(t)
print("wf", wf, "bytes/launch", nbytes)
for k, ts in res.items():
    med, mn = float(np.median(ts)), float(np.min(ts))
    print("%-24s median %.4f ms  min %.4f ms  -> %.0f GB/s (median)  %.0f Msamples/s" % (
        k, med, mn, nbytes / med / 1e6, V * B / med / 1e3))
