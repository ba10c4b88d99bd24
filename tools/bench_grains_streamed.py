#!/usr/bin/env python3
"""tools/bench_grains_streamed.py -- maxiStretch and maxiPitchShift on the config-5 shape (2048 streams x 70 560 samples, grainLength
0.05, overlaps 4): the call as ONE launch (knob grain_streamed 1, scheduler lanes beside the K8d tile renders) against the time
slices on the auxiliary streams (0).  Wall time per call (host clock around a synchronised call), best of five, interleaved."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
S, T = 2048, 70560
rng = np.random.default_rng(0x4D415849)
Ls = 4410000; n = np.arange(Ls)
smp = 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) + 0.05 * rng.uniform(-1, 1, Ls)
sb = mx.maxiSampleBank(1); sb.setSample(smp)
speed = 0.25 + 1.5 * (np.arange(S) % 97) / 96
out = mx.DeviceBuffer((T, S), zero=False)
banks = {"stretch": mx.maxiStretchBank(S, sb, "hann"), "pitch": mx.maxiPitchShiftBank(S, sb, "hann")}
best = {}
for r in range(5):
    for streamed in (1, 0):
        L.mxg_tune(b"grain_streamed", streamed)
        for name, bank in banks.items():
            bank.setPosition(np.arange(S) / S)
            bank.grains.upload(np.zeros((4, 8, S)))
            L.mxg_sync(); t0 = time.perf_counter()
            if name == "stretch":
                bank.play(speed, 0.8, 0.05, 4, T, out=out)
            else:
                bank.play(speed, 0.05, 4, T, out=out)
            L.mxg_sync(); dt = time.perf_counter() - t0
            k = (name, streamed)
            best[k] = min(best.get(k, 1e9), dt)
L.mxg_tune(b"grain_streamed", 1)
for name in banks:
    print("%-8s one launch %.3f ms   time slices %.3f ms" % (name, best[(name, 1)] * 1e3, best[(name, 0)] * 1e3), flush=True)
