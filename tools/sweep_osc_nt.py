import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
B = 512
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(10): fn()
        L.mxg_stream_sync(None)
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
for V in (65536, 131072, 262144, 1048576):
    v = np.arange(V)
    freq = mx.DeviceBuffer.from_numpy(20 + (v % 65536) * 0.30517578125)
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    out = mx.DeviceBuffer((B, V), zero=False)
    for nt in (0, 1):
        L.mxg_tune(b"osc_nt", nt)
        for blk in (256, 512):
            L.mxg_tune(b"osc_block", blk)
            t = timed(lambda: L.mxg_osc_render(8, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, out.ptr, None), 60)
            print("V %8d osc_nt %d osc_block %d: sinebuf %.1f us = %.0f GB/s" % (V, nt, blk, t, 8.047 * V * B / t / 1e3), flush=True)
    del out
