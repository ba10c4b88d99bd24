#!/usr/bin/env python3
"""tools/bench_osctab.py -- the per-voice wavetable EXTENSION (mxg_osc_render_tables): us per 512-sample block and the fraction of
the 8 TB/s HBM peak on its algorithmic READ bytes (4112 B of table + 8 B freq + 8 B pan + 8 B phase per voice and block), mixdown
form (no per-voice store) and stored form; the table array is far larger than the Infinity Cache (V x 4112 B: 539 MB at 131 072)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402
L = mx.lib(); chk = mx._lib.check; chk(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
B = 512
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
chk(L.mxg_prof_enable(1), "prof")
for V in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,131072,262144").split(",")]:
    rng = np.random.default_rng(1)
    freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
    pan = mx.DeviceBuffer.from_numpy(np.arange(V) / (V - 1.0))
    tabs = L.mxg_malloc(V * 514 * 8); assert tabs
    one = np.sin(2 * np.pi * np.arange(514) / 512.0)
    host = np.tile(one, (4096, 1))
    for c in range(0, V, 4096):
        chk(L.mxg_memcpy_h2d(tabs + c * 514 * 8, host.ctypes.data, min(4096, V - c) * 514 * 8, None), "h2d")
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    G = L.mxg_osc_tables_groups(V)
    rows = mx.DeviceBuffer((G, B, 2))
    out = mx.DeviceBuffer((B, V), zero=False)
    algo = V * (514 * 8 + 24.0)
    for name, o in (("mixdown only", None), ("mixdown + per-voice block", out.ptr)):
        fn = lambda: chk(L.mxg_osc_render_tables(V, B, freq.ptr, tabs, phase.ptr, hold.ptr, o, pan.ptr, rows.ptr, None), "render_tables")
        for _ in range(20): fn()
        L.mxg_prof_reset()
        chk(L.mxg_event_record(e0, None), "rec")
        for _ in range(100): fn()
        chk(L.mxg_event_record(e1, None), "rec"); chk(L.mxg_event_sync(e1), "sync"); chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "el")
        us = ms.value / 100 * 1e3
        kern = {}
        for i in range(L.mxg_prof_count()):
            lab, t, cnt = ctypes.c_char_p(), ctypes.c_double(0), ctypes.c_size_t(0)
            chk(L.mxg_prof_read(i, ctypes.byref(lab), ctypes.byref(t), ctypes.byref(cnt)), "prof_read")
            if cnt.value: kern[lab.value.decode()] = round(t.value / cnt.value * 1e3, 2)
        print("%8d voices, %-26s %8.1f us  read %.0f MB  %.0f GB/s  %.3f of 8 TB/s   kernels (us, with event overhead): %s"
              % (V, name, us, algo / 1e6, algo / us / 1e3, algo / us / 1e3 / 8000, kern), flush=True)
    L.mxg_free(tabs)
