#!/usr/bin/env python3
"""tools/sweep_heavy_osc.py [wf ...] -- the launch shapes of the VALU-heavy oscillators (sinewave 0, coswave 1, sinebuf4 9; any K1 waveform
number works) at 65 536 voices x 512: voices per lane x time parts x store flavour, interleaved rounds over ROTATED block buffers (8 x 268 MB,
so the Infinity Cache cannot absorb the stream), median / min per variant.  MODE=one: only the automatic rule, REPS launches (for rocprofv3)."""
import ctypes
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

L = mx.lib()
mx._lib.check(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
V, B = int(os.environ.get("VOICES", "65536")), 512
wfs = [int(a) for a in sys.argv[1:]] or [0, 9]
NBUF = 8
freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * (20000.0 / V))
p1 = mx.DeviceBuffer.from_numpy(np.full(V, 0.25)); p2 = mx.DeviceBuffer.from_numpy(np.full(V, 0.75))
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
outs = [mx.DeviceBuffer((B, V), zero=True) for _ in range(NBUF)]
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
KNOBS = (b"osc_vpl", b"osc_split", b"osc_store")
state = {"i": 0}


def launch(wf):
    state["i"] = (state["i"] + 1) % NBUF
    L.mxg_osc_render(wf, V, B, freq.ptr, 0, p1.ptr, p2.ptr, phase.ptr, hold.ptr, outs[state["i"]].ptr, None)


def timed(fn, reps):
    L.mxg_event_record(e0, None)
    for _ in range(reps):
        fn()
    L.mxg_event_record(e1, None)
    L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / reps * 1e3


def ramp(fn):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            fn()
        L.mxg_sync()


if os.environ.get("MODE") == "one":
    for wf in wfs:
        ramp(lambda: launch(wf))
        print("wf", wf, "auto %.2f us" % timed(lambda: launch(wf), int(os.environ.get("REPS", "50"))))
    sys.exit(0)

for wf in wfs:
    variants = {"auto": (0, 0, 0)}
    for vpl, split, store in itertools.product((1, 2), (1, 2, 3, 4), (1, 3, 4, 5)):   # knob osc_store: 1 plain 8 B, 2 nt 8 B, 3 / 4 / 5 pair rows plain / sc1 / nt
        if vpl == 2 and store >= 4:
            continue   # (two voices per lane have their own 16-byte rows: flavours 0..2)
        variants["vpl%d split%d store%d" % (vpl, split, store)] = (vpl, split, store)
    res = {k: [] for k in variants}
    ramp(lambda: launch(wf))
    for rnd in range(int(os.environ.get("ROUNDS", "5"))):
        for k, kn in variants.items():
            for name, val in zip(KNOBS, kn):
                L.mxg_tune(name, val)
            t = timed(lambda: launch(wf), 24)
            if rnd:
                res[k].append(t)
    for name in KNOBS:
        L.mxg_tune(name, 0)
    print("## wf", wf, "voices", V)
    for k, ts in sorted(res.items(), key=lambda kv: np.median(kv[1])):
        med = float(np.median(ts))
        print("%-26s median %6.2f us  min %6.2f  %.3f of 8 TB/s" % (k, med, float(np.min(ts)), 8.047 * V * B / med / 1e3 / 8000))
