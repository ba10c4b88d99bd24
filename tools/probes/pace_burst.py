"""K2f's pace controller under back-to-back launches: bursts of 50 launches, after each the controller's words and the burst's average."""
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import maximilian_amd as mx
import torch
L = mx.lib()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
V, N = 65536, 512
v = np.arange(V)
freq, cutoff, res = 50.0 + 7.0 * (v % 600), 300.0 + 5.0 * (v % 800), 1.0 + (v % 5)
trig = np.ones(N, dtype=np.int32)
vb = mx.maxiVoiceBank(V)
vb.env.setAttack(1); vb.env.setDecay(5); vb.env.setSustain(0.5); vb.env.setRelease(20)
cu = cutoff if mode == 0 else np.full(V, 9000.0)
L.mxg_tune(b"voice_diet", 2)
if fixed: L.mxg_tune(b"voice_pace", fixed)
buf = (ctypes.c_uint * 32)()
vb.render(mode, freq, cu, res, trig, N)
for b in range(16):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(50):
        vb.render(mode, freq, cu, res, trig, N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 50 * 1e6
    L.mxg_debug_voice_pace(None, buf)
    w = list(buf)[8 * mode: 8 * mode + 8]
    print("burst %2d  %.2f us per launch | P %d  win %d lates %d booted %d | mean late %d" % (b, dt, w[0], w[1] & 255, (w[1] >> 8) & 255, w[1] >> 16, w[7]))
