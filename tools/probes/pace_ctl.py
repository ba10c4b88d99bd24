"""The pace controller of K2f launch by launch (mxg_debug_voice_pace): P, on-schedule count, floor, age, -, -, -, mean lateness."""
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
L.mxg_tune(b"voice_diet", 2)
buf = (ctypes.c_uint * 32)()
import torch, time
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    t = time.perf_counter()
    vb.render(mode, freq, cu, res, trig, N)
    L.mxg_debug_voice_pace(None, buf)
    w = list(buf)[8 * mode: 8 * mode + 8]
    print(i, "P %d  win %d lates %d  clean %d need %d booted %d | mean late %d" % (w[0], w[1] & 255, w[1] >> 8, w[2] & 255, (w[2] >> 8) & 255, w[2] >> 16, w[7]))
