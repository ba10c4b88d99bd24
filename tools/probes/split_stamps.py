"""Per-chunk cycle stamps of the two-stage K2f (build with -DMXG_SPLIT_ABLATE=8, MXG_LIB=build/ab/ab_split8.so): workgroup 0, pair 0."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import maximilian_amd as mx
L = mx.lib()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
V, N = 65536, 512
v = np.arange(V)
freq, cutoff, res = 50.0 + 7.0 * (v % 600), 300.0 + 5.0 * (v % 800), 1.0 + (v % 5)
trig = np.ones(N, dtype=np.int32)
vb = mx.maxiVoiceBank(V)
vb.env.setAttack(1); vb.env.setDecay(5); vb.env.setSustain(0.5); vb.env.setRelease(20)
cu = cutoff if mode == 0 else np.full(V, 9000.0)
L.mxg_tune(b"voice_split", 2)
for _ in range(40):
    vb.render(mode, freq, cu, res, trig, N)
import torch
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 128 * 4))()
L.mxg_debug_split_stamps.restype = ctypes.c_int
print("rc", L.mxg_debug_split_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(2, 128, 4)[:, :64]
t0 = a[0, 0, 0]
print("front: chunk  start  compute->wait  wait  write+publish | back: start fetch compute   (cycles; start relative to the front's chunk 0)")
for k in range(64):
    f, b = a[0, k], a[1, k]
    print("%3d  %7d  %5d %5d %5d  |  %7d %5d %5d" % (k, f[0] - t0, f[1] - f[0], f[2] - f[1], f[3] - f[2], b[0] - t0, b[1] - b[0], b[2] - b[1]))
print("front total", a[0, 63, 3] - a[0, 0, 0], "back total", a[1, 63, 2] - a[1, 0, 0])
