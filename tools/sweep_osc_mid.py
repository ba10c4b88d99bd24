#!/usr/bin/env python3
"""tools/sweep_osc_mid.py [V ...] -- K1 (sinebuf) between the 65 536- and the 98 304-voice shapes (73 728 ... 90 112 voices: 2.25-2.75 wavefronts
of 128 voices per CU, the one range where the automatic rule stays below 0.70): voices per lane x store x passes, rotated destination."""
import ctypes
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
ARENA = 4 << 30
arena = L.mxg_malloc(ARENA)
chk(L.mxg_memset(arena, 0, ARENA, None), "memset"); chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
sizes = [int(a) for a in sys.argv[1:]] or [73728, 81920, 90112]
KN = (b"osc_vpl", b"osc_store", b"osc_passes", b"osc_xcd")
wf = int(os.environ.get("WF", "8"))
for V in sizes:
    nb = V * B * 8
    regions = ARENA // nb
    freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * (20000.0 / V))
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    k = [0]

    def run():
        k[0] += 1
        chk(L.mxg_osc_render(wf, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, arena + (k[0] % regions) * nb, None), "osc")

    def timed(reps=10):
        L.mxg_event_record(e0, None)
        for _ in range(reps):
            run()
        L.mxg_event_record(e1, None)
        L.mxg_event_sync(e1)
        L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        return ms.value / reps * 1e3

    variants = {"auto": (0, 0, 0, 0)}
    for vpl, store in ((1, 1), (1, 2), (1, 4), (1, 5), (2, 1), (2, 3)):   # one voice: plain / nt 8 B, pair rows sc1 / nt; two voices: plain / sc1 16 B
        for passes in (1, 2, 3):
            variants["vpl%d store%d passes%d" % (vpl, store, passes)] = (vpl, store, passes, 1)
    res = {n: [] for n in variants}
    for _ in range(30):
        run()
    for rnd in range(6):
        for n, kn in variants.items():
            for name, val in zip(KN, kn):
                L.mxg_tune(name, val)
            t = timed()
            if rnd:
                res[n].append(t)
    for name in KN:
        L.mxg_tune(name, 0)
    print("## %d voices, waveform %d" % (V, wf))
    for n, ts in sorted(res.items(), key=lambda kv: np.median(kv[1]))[:8]:
        med = float(np.median(ts))
        print("%-28s median %7.1f us  %.3f of 8 TB/s" % (n, med, 8.047 * V * B / med / 1e3 / 8000))
    print("%-28s median %7.1f us" % ("auto", float(np.median(res["auto"]))))
