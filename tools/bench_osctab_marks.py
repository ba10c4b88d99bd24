#!/usr/bin/env python3
"""tools/bench_osctab_marks.py -- K1t's kernels (marks pass, main kernel) across bank sizes: mxg_osc_render_tables with the fused
mixdown, per-kernel HIP events (mxg_prof).  MXG_LIB selects the build."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
B = 512
for V in [int(x) for x in (sys.argv[1:] or ["16384", "32768", "65536", "131072", "262144"])]:
    v = np.arange(V)
    freq = mx.DeviceBuffer.from_numpy(20.0 + v * (20000.0 / V))
    pan = mx.DeviceBuffer.from_numpy(v / (V - 1.0))
    tab = mx.DeviceBuffer((V, 514), np.float64, zero=True)
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    G = L.mxg_osc_tables_groups(V)
    rows = mx.DeviceBuffer((G, B, 2), np.float64)
    call = lambda: mx._lib.check(L.mxg_osc_render_tables(V, B, freq.ptr, tab.ptr, phase.ptr, hold.ptr, None, pan.ptr, rows.ptr, None), "render")
    for _ in range(5): call()
    L.mxg_prof_enable(1); L.mxg_prof_reset()
    for _ in range(30): call()
    L.mxg_sync()
    res = {}
    for i in range(L.mxg_prof_count()):
        lab = ctypes.c_char_p(); ms = ctypes.c_double(); n = ctypes.c_size_t()
        L.mxg_prof_read(i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(n))
        if n.value: res[lab.value.decode()] = round(ms.value / n.value * 1e3, 2)
    L.mxg_prof_enable(0)
    print("V=%7d  %s" % (V, res))
