import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
v = np.arange(V)
D = mx.DeviceBuffer.from_numpy
out = mx.DeviceBuffer((B, V), zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=200):
    for _ in range(100): fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
rng = np.random.default_rng(1)
dsp = D(0.5 + (v % 97) / 96.0)
L.mxg_tune(b"smp_split", 6)
for slen in (4000000, 441000, 131072, 65536, 32768):
    sb = mx.maxiSampleBank(V); sb.setSample(rng.uniform(-1, 1, slen)); sb.setPosition(v / V)
    t = timed(lambda: L.mxg_sample_render(4, V, B, sb.d_samples, sb.length, 44100, dsp.ptr, 0, None, None, sb.position.ptr, out.ptr, None))
    print("sample length %8d (%6.2f MB)  playAtSpeed %.1f us" % (slen, slen * 8 / 1e6, t), flush=True)
