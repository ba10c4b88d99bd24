#!/usr/bin/env python3
"""tools/fused_ab.py -- config 4 through mxg_fft_mfcc_batch under several knob settings, interleaved on ONE box.

  python tools/fused_ab.py [fused_mel=1,fft_exact=1 fused_mel=3,fft_exact=1 ...]      (default: the six forms)
  FRAMES (default 2^20), REPS (launches per timing, default 10), ROUNDS (default 3), ONLY=<index> (run one setting REPS times
  and exit: the command rocprofv3 --pmc wraps)

Signal = SURVEY 8(d)'s config-4 mix, generated on the device.  Prints kernel ms per setting and round (HIP events around REPS
back-to-back launches) and, once, each setting's distance from the first one (band sums, mfcc)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

N = int(os.environ.get("FRAMES", 1 << 20))
REPS = int(os.environ.get("REPS", 10))
ROUNDS = int(os.environ.get("ROUNDS", 3))
settings = sys.argv[1:] or ["fused_mel=1,fft_exact=1", "fused_mel=2,fft_exact=1", "fused_mel=3,fft_exact=1",
                            "fused_mel=1,fft_exact=0", "fused_mel=2,fft_exact=0", "fused_mel=3,fft_exact=0"]
L = mx.lib()
mx._lib.check(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0x4D415849)
sig = torch.empty(N * 1024, dtype=torch.float32, device=dev)
chunk = 1 << 24
for o in range(0, N * 1024, chunk):
    n = torch.arange(o, min(o + chunk, N * 1024), device=dev, dtype=torch.float64)
    k = torch.floor(n / 1024)
    x = 0.4 * torch.sin(2 * np.pi * 220 * n / 44100) + 0.3 * torch.sin(2 * np.pi * (440 + 0.01 * k) * n / 44100) \
        + 0.1 * (torch.rand(n.numel(), device=dev, generator=g, dtype=torch.float64) * 2 - 1)
    sig[o:o + n.numel()] = x.to(torch.float32)
torch.cuda.synchronize()
f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
out = torch.empty((N, 13), dtype=torch.float64, device=dev)
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()


def apply(setting):
    for kv in setting.split(","):
        k, v = kv.split("=")
        L.mxg_tune(k.encode(), int(v))


def launch(raw=None):
    mx._lib.check(L.mxg_fft_mfcc_batch(f.plan, m.plan, sig.data_ptr(), 1024, N, None, raw, None, out.data_ptr(), None), "fused")


def timed():
    launch()
    L.mxg_event_record(e0, None)
    for _ in range(REPS):
        launch()
    L.mxg_event_record(e1, None)
    L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / REPS


if "ONLY" in os.environ:
    apply(settings[int(os.environ["ONLY"])])
    print(settings[int(os.environ["ONLY"])], "kernel_ms %.4f" % timed())
    sys.exit(0)

ref = ref_raw = None
nchk = min(N, 1 << 16)
for s in settings:
    apply(s)
    raw = torch.empty((N, 42), dtype=torch.float64, device=dev)
    launch(raw.data_ptr())
    L.mxg_sync()
    if ref is None:
        ref, ref_raw = out.clone(), raw.clone()
    else:
        rel = ((raw - ref_raw).abs().amax(dim=1) / ref_raw.abs().amax(dim=1).clamp_min(1e-300)).max().item()
        print("%-28s vs %s: band sums rel %.3e, mfcc abs %.3e (every one of %d frames)" % (s, settings[0], rel, (out - ref).abs().max().item(), N))
    del raw
res = {s: [] for s in settings}
for r in range(ROUNDS):
    for s in settings:
        apply(s)
        res[s].append(round(timed(), 4))
for s in settings:
    t = min(res[s])
    print("%-28s kernel_ms %s   best %.4f = %.3f of 8 TB/s on 4200 B/frame" % (s, res[s], t, N * 4200 / (t * 1e-3) / 8e12))
print(json.dumps(res))
