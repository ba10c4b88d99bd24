#!/usr/bin/env python3
"""tools/bench_grains.py -- config 5 (per-GPU share): S maxiTimeStretch<hann> streams over a 100 s sample,
grainLength 0.05, overlaps 4, T samples; reports stream-samples/s and grain-samples/s."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
LK = int(os.environ.get("LANES_K", 128))
S = int(os.environ.get("STREAMS", 2048)); T = int(os.environ.get("T", 70560)); REPS = int(os.environ.get("REPS", 3))
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024); L.mxg_tune(b"grain_lanes_k", LK)
rng = np.random.default_rng(0x4D415849)
Ls = 4410000; n = np.arange(Ls)
smp = 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) + 0.05 * rng.uniform(-1, 1, Ls)
sb = mx.maxiSampleBank(1); sb.setSample(smp)
bank = mx.maxiTimeStretchBank(S, sb, "hann")
speed = 0.25 + 1.5 * (np.arange(S) % 97) / 96
out = mx.DeviceBuffer((T, S), zero=False)
best = 1e9
for r in range(REPS):
    bank.setPosition(np.arange(S) / S)
    bank.grains.upload(np.zeros((4, 8, S)))
    L.mxg_sync(); t0 = time.perf_counter()
    bank.play(speed, 0.05, 4, T, out=out)
    L.mxg_sync(); dt = time.perf_counter() - t0
    best = min(best, dt)
grains_alive = 4.0  # overlaps
print("S=%d T=%d: %.2f ms  -> %.1f M stream-samples/s, ~%.1f G grain-samples/s" % (
    S, T, best * 1e3, S * T / best / 1e6, S * T * grains_alive / best / 1e9))
