#!/usr/bin/env python3
"""tools/bench_samplebank.py -- a bare loop of playAtSpeed launches over the HBM-resident sample bank of bench.py's `sample_bank` workload
(65 536 heads over one 8.6 GB sample), for counter passes (tools/gpu_runs/r06_s.sh); REPS in the environment scales the loop."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
dev = torch.device("cuda", 0)
V, B, Ls = 65536, 512, (1 << 30) - (1 << 20)
arena = torch.empty(Ls + 64, dtype=torch.float64, device=dev)
arena[:8].zero_(); arena[Ls:].zero_()
for c0 in range(0, Ls, 1 << 26):
    c1 = min(Ls, c0 + (1 << 26))
    xs = torch.arange(c0, c1, dtype=torch.float64, device=dev)
    arena[8 + c0:8 + c1] = torch.frac(xs * 0.3183098861837907) - 0.5
    del xs
torch.cuda.synchronize()
d_smp = arena.data_ptr() + 64
vv = np.arange(V)
speed = mx.DeviceBuffer.from_numpy(0.5 + (vv * 40503 % V) / float(V))
pos = mx.DeviceBuffer.from_numpy((vv * (Ls // V)).astype(np.float64) + 0.25)
outs = [mx.DeviceBuffer((B, V), zero=False) for _ in range(8)]
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
reps = int(os.environ.get("REPS", "10")) * 20
k = [0]
def call():
    mx._lib.check(L.mxg_sample_render(4, V, B, d_smp, Ls, 44100, speed.ptr, 0, None, None, pos.ptr, outs[k[0] % 8].ptr, None), "render"); k[0] += 1
for _ in range(20): call()
L.mxg_event_record(e0, None)
for _ in range(reps): call()
L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
print("sample bank playAtSpeed %.1f us per block" % (ms.value / reps * 1e3))
