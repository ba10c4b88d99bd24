#!/usr/bin/env python3
"""tools/sweep_rw_store.py -- the read + write bank kernels (8 B in + 8 B out per sample) with 8-byte streams and with the 16-byte
pair-row streams (knob rw_store), input and output blocks rotating through a 6 GiB arena each.  us per 512-sample block and the
fraction of 8 TB/s on 16 B per sample."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--voices", default="32768,65536,131072,262144")
ap.add_argument("--out", default=None)
args = ap.parse_args()
L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
B = 512
ARENA = 4 << 30
a_in, a_out = L.mxg_malloc(ARENA), L.mxg_malloc(ARENA)
chk(L.mxg_memset(a_in, 0, ARENA, None), "memset"); chk(L.mxg_memset(a_out, 0, ARENA, None), "memset"); chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
MODES = [("auto", 0), ("8-byte", 1), ("pairs plain", 2), ("pairs sc1", 3), ("pairs nt", 4)]
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


def timed(fn, reps=8):
    chk(L.mxg_event_record(e0, None), "rec")
    for _ in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps


emit("# Read + write bank kernels: 8-byte vs 16-byte pair-row streams (MI355X, 512-sample blocks; us / fraction of 8 TB/s on 16 B per sample)")
emit()
emit("| kernel | voices | " + " | ".join(m[0] for m in MODES) + " |")
emit("|---|---|" + "---|" * len(MODES))
for name, kind in (("maxiBiquad", 2), ("maxiSVF", 1), ("maxiDCBlocker", 0)):
    for V in [int(x) for x in args.voices.split(",")]:
        nb = V * B * 8
        regions = ARENA // nb
        k = [0]
        if kind == 2:
            b = mx.maxiBiquadBank(V); b.set(np.zeros(V, np.int32), np.full(V, 1200.0), np.full(V, 0.7), np.zeros(V))
            b.play(mx.DeviceBuffer((8, V)))
        elif kind == 1:
            b = mx.maxiSVFBank(V); b.setCutoff(800.0); b.setResonance(2.0); b.play(mx.DeviceBuffer((8, V)), 0.5, 0.25, 0.1, 0.1)
        else:
            b = mx.maxiDCBlockerBank(V); b.play(mx.DeviceBuffer((8, V)), 0.995)

        def run():
            k[0] += 1
            r = (k[0] % regions) * nb
            chk(L.mxg_filter2_render(kind, V, B, a_in + r, b.coef.ptr, b.state.ptr, a_out + r, None), "f2")
        res = {}
        for rnd in range(6):
            for nm, rw in MODES:
                L.mxg_tune(b"rw_store", rw)
                t = timed(run)
                if rnd:
                    res.setdefault(nm, []).append(t)
        L.mxg_tune(b"rw_store", 0)
        emit("| %s | %d | " % (name, V) + " | ".join("%.1f / %.3f" % (np.median(res[m[0]]) * 1e3, 2 * nb / np.median(res[m[0]]) / 1e6 / 8000) for m in MODES) + " |")
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
