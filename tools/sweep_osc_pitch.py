#!/usr/bin/env python3
"""tools/sweep_osc_pitch.py -- K1 (sinebuf) on the banks whose row pitch is a multiple of 2 MB (524 288 voices and up), where neither the
launch plan nor the single launch reaches 0.75: voices per lane x time parts x passes x XCD numbering, destination rotated over a 6 GiB
arena, interleaved rounds, median.  (Time parts put half of the wavefronts 256 rows ahead of the others.)"""
import ctypes
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
B = 512
ARENA = 6 << 30
arena = L.mxg_malloc(ARENA)
chk(L.mxg_memset(arena, 0, ARENA, None), "memset"); chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
sizes = [int(a) for a in sys.argv[1:]] or [524288, 786432, 1048576]
KN = (b"osc_vpl", b"osc_store", b"osc_split", b"osc_passes", b"osc_xcd")
for V in sizes:
    nb = V * B * 8
    regions = ARENA // nb
    freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * (20000.0 / V))
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    k = [0]

    def run():
        k[0] += 1
        chk(L.mxg_osc_render(8, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, arena + (k[0] % regions) * nb, None), "osc")

    def timed(reps=4):
        L.mxg_event_record(e0, None)
        for _ in range(reps):
            run()
        L.mxg_event_record(e1, None)
        L.mxg_event_sync(e1)
        L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        return ms.value / reps * 1e3

    variants = {"auto": (0, 0, 0, 0, 0)}
    for vpl, split, passes, xcd in itertools.product((1, 2), (1, 2, 4), (1, 2, 4), (1, 2)):
        if split > 1 and passes > 1:
            continue
        store = 4 if vpl == 1 else 3   # write-through 16-byte stores either way
        variants["vpl%d split%d passes%d xcd%d" % (vpl, split, passes, xcd - 1)] = (vpl, store, split, passes, xcd)
    res = {n: [] for n in variants}
    for _ in range(20):
        run()
    for rnd in range(5):
        for n, kn in variants.items():
            for name, val in zip(KN, kn):
                L.mxg_tune(name, val)
            t = timed()
            if rnd:
                res[n].append(t)
    for name in KN:
        L.mxg_tune(name, 0)
    print("## %d voices (row pitch %.1f MB)" % (V, V * 8 / 2**20))
    for n, ts in sorted(res.items(), key=lambda kv: np.median(kv[1])):
        med = float(np.median(ts))
        print("%-30s median %8.1f us  %.3f of 8 TB/s" % (n, med, 8.047 * V * B / med / 1e3 / 8000))
