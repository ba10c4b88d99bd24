#!/usr/bin/env python3
"""tools/sweep_osc_auto.py -- what mxg_osc_render does when left alone (every knob 0), by bank size: us per 512-sample block and the
fraction of the 8 TB/s HBM peak on 8 B per sample, destination rotated over a 6 GiB arena; interleaved rounds, median."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--voices", default="16384,32768,49152,65536,81920,98304,114688,131072,163840,196608,229376,262144,393216,524288,786432,1048576")
ap.add_argument("--waveforms", default="8,3,10,9,0")
ap.add_argument("--out", default=None)
args = ap.parse_args()
L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
B = 512
ARENA = 6 << 30
arena = L.mxg_malloc(ARENA)
assert arena
chk(L.mxg_memset(arena, 0, ARENA, None), "memset")
chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
NAMES = {v: k for k, v in mx.OSC_WAVEFORMS.items()}
wfs = [int(x) for x in args.waveforms.split(",")]
sizes = [int(x) for x in args.voices.split(",")]
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


res = {(wf, V): [] for wf in wfs for V in sizes}
for V in sizes:
    nbytes = V * B * 8
    regions = max(1, ARENA // nbytes)
    freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
    p1 = mx.DeviceBuffer.from_numpy(np.full(V, 0.25))
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    ctr = [0]
    for rnd in range(args.rounds + 1):
        for wf in wfs:
            chk(L.mxg_event_record(e0, None), "rec")
            for _ in range(args.reps):
                ctr[0] += 1
                chk(L.mxg_osc_render(wf, V, B, freq.ptr, 0, p1.ptr, p1.ptr, phase.ptr, hold.ptr, arena + (ctr[0] % regions) * nbytes, None), "render")
            chk(L.mxg_event_record(e1, None), "rec")
            chk(L.mxg_event_sync(e1), "sync")
            chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
            if rnd:
                res[(wf, V)].append(ms.value / args.reps)
    del freq, p1, phase, hold
emit("# mxg_osc_render, automatic (MI355X, 512-sample blocks, destination rotated): us per block / fraction of 8 TB/s on 8 B per sample")
emit()
emit("| voices | " + " | ".join(NAMES[w] for w in wfs) + " |")
emit("|---|" + "---|" * len(wfs))
for V in sizes:
    emit("| %d | " % V + " | ".join("%.1f / %.3f" % (np.median(res[(w, V)]) * 1e3, V * B * 8 / np.median(res[(w, V)]) / 1e6 / 8000) for w in wfs) + " |")
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
