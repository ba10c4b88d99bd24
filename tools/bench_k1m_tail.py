#!/usr/bin/env python3
"""tools/bench_k1m_tail.py -- does K1m's distance from K1 sit in the block's tail?  K1 and K1m (rows form, the N > 1 step's kernel) with
512-sample blocks and with ONE launch of 2048 samples (four combine windows, one tail), output rotated; us per 512 samples."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V = 65536
freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * 0.30517578125)
pan = mx.DeviceBuffer.from_numpy(np.arange(V) / (V - 1.0))
phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
G = L.mxg_osc_mix_groups(V)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
for N in (512,):
    nbuf = max(1, (1 << 31) // (V * N * 8))
    outs = [mx.DeviceBuffer((N, V), np.float64, zero=False) for _ in range(nbuf)]
    rows = mx.DeviceBuffer((G, N, 2), np.float64)
    rows32 = [mx.DeviceBuffer((G, N, 2), np.float64) for _ in range(32)]
    k = [0]
    def k1():
        mx._lib.check(L.mxg_osc_render(8, V, N, freq.ptr, 0, None, None, phase.ptr, hold.ptr, outs[k[0] % nbuf].ptr, None), "k1"); k[0] += 1
    def k1m():
        mx._lib.check(L.mxg_osc_render_mix_rows(8, V, N, freq.ptr, None, None, phase.ptr, hold.ptr, outs[k[0] % nbuf].ptr, pan.ptr, rows.ptr, None), "k1m"); k[0] += 1
    def k1m_rot():
        mx._lib.check(L.mxg_osc_render_mix_rows(8, V, N, freq.ptr, None, None, phase.ptr, hold.ptr, outs[k[0] % nbuf].ptr, pan.ptr, rows32[k[0] % 32].ptr, None), "k1m"); k[0] += 1
    for name, fn in (("K1", k1), ("K1m rows", k1m), ("K1m rows rotating over 32 row buffers", k1m_rot), ("K1", k1), ("K1m rows", k1m), ("K1m rows rotating over 32 row buffers", k1m_rot)):
        reps = 400 * 512 // N
        for _ in range(reps // 4): fn()
        L.mxg_event_record(e0, None)
        for _ in range(reps): fn()
        L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        print("N=%5d %-40s %.2f us per 512 samples" % (N, name, ms.value / reps * 1e3 * 512 / N), flush=True)
    del outs
