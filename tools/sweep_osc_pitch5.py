#!/usr/bin/env python3
"""tools/sweep_osc_pitch5.py -- K1 (sinebuf) with padded row pitches: the bank sizes whose natural pitch is a multiple of 2 MB
(524 288 / 786 432 / 1 048 576 voices) and the ones between the 65 536- and 98 304-voice shapes (73 728, 81 920), output rotated over
>= 2 GiB of block buffers.  Prints us per block and the fraction of 8 TB/s on 8.047 B/sample per (voices, pad)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
B = 512
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
sizes = [int(x) for x in sys.argv[1:]] or [73728, 81920, 262144, 524288, 786432, 1048576]
for V in sizes:
    freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * (20000.0 / V))
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    row = []
    for pad, plan in [(pd, pl) for pd in (0, 128, 512, 1024) for pl in (1, 2, 3)]:   # doubles: 0, 1 KB, 4 KB, 8 KB; osc_plan 1 never / 2 / 3 always
        L.mxg_tune(b"osc_plan", plan)
        P = V + pad
        nbuf = max(1, -(-(1 << 31) // (P * B * 8)))
        bufs = [mx.DeviceBuffer((B, P), np.float64, zero=False) for _ in range(nbuf)]
        k = [0]
        def call():
            mx._lib.check(L.mxg_osc_render_pitch(8, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, bufs[k[0] % nbuf].ptr, P * 8, None), "render")
            k[0] += 1
        reps = max(10, int(3e9 / (V * B * 8)))
        for _ in range(max(3, reps // 4)): call()
        best = 1e9
        for _ in range(3):
            L.mxg_event_record(e0, None)
            for _ in range(reps): call()
            L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            best = min(best, ms.value / reps)
        row.append(("%d/p%d" % (pad * 8, plan), best * 1e3, V * (B * 8 + 24) / (best * 1e-3) / 8e12))
        del bufs
    L.mxg_tune(b"osc_plan", 0)
    print("V=%8d  " % V + "  ".join("%s: %.0f us %.3f" % (p, us, fr) for p, us, fr in row), flush=True)
