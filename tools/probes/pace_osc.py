"""K1's pace controller (waveform's slot) over bursts of back-to-back launches: python tools/probes/pace_osc.py WAVEFORM VOICES [fixed P]"""
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import maximilian_amd as mx
import torch
L = mx.lib()
wfname = sys.argv[1] if len(sys.argv) > 1 else "sinebuf"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N = 512
names = ["sinewave", "coswave", "phasor", "saw", "triangle", "square", "pulse", "impulse", "sinebuf", "sinebuf4", "sawn", "phasorBetween"]
wf = names.index(wfname)
if fixed: L.mxg_tune(b"osc_pace", fixed)
freq = torch.linspace(50.0, 2000.0, V, dtype=torch.float64, device="cuda")
phase = torch.zeros(V, dtype=torch.float64, device="cuda")
hold = torch.zeros(V, dtype=torch.float64, device="cuda")
outs = [torch.empty((N, V), dtype=torch.float64, device="cuda") for _ in range(4)]
p1 = torch.full((V,), 0.3, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
def go(o):
    rc = L.mxg_osc_render(wf, V, N, P(freq.data_ptr()), 0, P(p1.data_ptr()), P(p1.data_ptr()), P(phase.data_ptr()), P(hold.data_ptr()), P(o.data_ptr()), P(st))
    assert rc == 0, rc
buf = (ctypes.c_uint * 128)()
go(outs[0])
for b in range(12):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40):
        go(outs[i & 3])
    e1.record()
    torch.cuda.synchronize()
    L.mxg_debug_osc_pace(P(st), buf)
    w = list(buf)[8 * wf: 8 * wf + 8]
    print("burst %2d | P %d  win %d lates %d booted %d | mean late %d" % (b, w[0], w[1] & 255, (w[1] >> 8) & 255, w[1] >> 16, w[7]))
