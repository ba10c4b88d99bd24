#!/usr/bin/env python3
"""tools/bench_small_banks.py -- latency of ONE block of a SMALL bank (the polyphony of a patch: 1 ... 64 voices x 512 samples)
through the lane-per-voice kernels (bit-exact, 512 dependent steps in one wavefront) and through the time-parallel scan kernels
(knob time_parallel = 1, csrc/scan.hip: tolerance mode).  Back-to-back launches on one stream, HIP events, median of rounds."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
args = ap.parse_args()
L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    chk(L.mxg_event_record(e0, None), "rec")
    for _ in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps * 1e3


emit("# Small banks: microseconds per 512-sample block (MI355X), lane-per-voice (bit-exact) vs time-parallel scan (tolerance 1e-10)")
emit()
emit("`python tools/bench_small_banks.py`: back-to-back launches on one stream (launch overhead included), 50 per measurement, median of 7.")
emit()
emit("| filter | voices | exact us | time-parallel us | speed-up |")
emit("|---|---|---|---|---|")
N = 512
for name in ("maxiBiquad", "maxiSVF", "maxiDCBlocker", "maxiFilter::lores"):
    for V in (1, 6, 16, 64, 256, 1024):
        rng = np.random.default_rng(V)
        x = mx.DeviceBuffer.from_numpy(rng.uniform(-1, 1, (N, V)))
        out = mx.DeviceBuffer((N, V))
        if name == "maxiBiquad":
            b = mx.maxiBiquadBank(V); b.set(np.zeros(V, np.int32), np.full(V, 1200.0), np.full(V, 0.7), np.zeros(V))
            b.play(x, out=out)
            fn = lambda: chk(L.mxg_filter2_render(2, V, N, x.ptr, b.coef.ptr, b.state.ptr, out.ptr, None), "f2")
        elif name == "maxiSVF":
            b = mx.maxiSVFBank(V); b.setCutoff(800.0); b.setResonance(2.0)
            b.play(x, 0.5, 0.25, 0.1, 0.1, out=out)
            fn = lambda: chk(L.mxg_filter2_render(1, V, N, x.ptr, b.coef.ptr, b.state.ptr, out.ptr, None), "f2")
        elif name == "maxiDCBlocker":
            b = mx.maxiDCBlockerBank(V)
            b.play(x, 0.995, out=out)
            fn = lambda: chk(L.mxg_filter2_render(0, V, N, x.ptr, b.coef.ptr, b.state.ptr, out.ptr, None), "f2")
        else:
            b = mx.maxiFilterBank(V)
            cut, res = np.full(V, 900.0), np.full(V, 4.0)
            from maximilian_amd.banks import filter_coeffs
            coef = mx.DeviceBuffer.from_numpy(filter_coeffs(0, cut, res))
            dc, dr = mx.DeviceBuffer.from_numpy(cut), mx.DeviceBuffer.from_numpy(res)
            fn = lambda: chk(L.mxg_filter_render(0, V, N, x.ptr, dc.ptr, 0, dr.ptr, 0, coef.ptr, b.state.ptr, out.ptr, None), "flt")
        res_ = {}
        for knob in (0, 1):
            L.mxg_tune(b"time_parallel", knob)
            res_[knob] = float(np.median([timed(fn) for _ in range(7)]))
        L.mxg_tune(b"time_parallel", 0)
        emit("| %s | %d | %.1f | %.1f | %.1fx |" % (name, V, res_[0], res_[1], res_[0] / res_[1]))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
