#!/usr/bin/env python3
"""tools/sweep_voice_store.py -- store stream of the fused voice kernel (K2f, config 3) by bank size: every setting of the knobs
voice_store / voice_xcd, destination rotated through a 6 GiB arena (HBM rates) or one reused block buffer, sustain segment of
the envelope (gate held) and the config-3 gate cycle.  Interleaved rounds in one process, median; us per 512-sample block and
the fraction of the 8 TB/s peak on 8.34 B per sample."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--voices", default="16384,32768,65536,98304,131072,262144,524288")
ap.add_argument("--out", default=None)
args = ap.parse_args()
L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
B = 512
ARENA = 6 << 30
arena = L.mxg_malloc(ARENA)
chk(L.mxg_memset(arena, 0, ARENA, None), "memset")
chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
MODES = [("auto", 0, 0), ("8B plain", 1, 1), ("8B nt", 2, 1), ("pair-rows plain", 3, 1), ("pair-rows sc1", 4, 1), ("pair-rows nt", 5, 1),
         ("8B nt xcd", 2, 2), ("pair-rows sc1 xcd", 4, 2), ("pair-rows nt xcd", 5, 2), ("8B plain xcd", 1, 2)]
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


def timed(fn, reps):
    chk(L.mxg_event_record(e0, None), "rec")
    for _ in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps


emit("# K2f (fused voice) store streams by bank size (MI355X, 512-sample blocks, mode A; us per block / fraction of 8 TB/s on 8.34 B per sample)")
emit()
emit("`python tools/sweep_voice_store.py`: gate held (sustain) after a 20-block lead-in; rotated = destination walks a 6 GiB arena.")
emit()
emit("| voices | " + " | ".join(m[0] for m in MODES) + " | best |")
emit("|---|" + "---|" * (len(MODES) + 1))
for rot in (True, False):
    for V in [int(x) for x in args.voices.split(",")]:
        nbytes = V * B * 8
        if not rot and nbytes > (300 << 20):
            continue
        regions = max(1, ARENA // nbytes)
        v = np.arange(V)
        freq = np.minimum(20.0 + (v % 65536) * 0.30517578125, 5000.0)
        vb = mx.maxiVoiceBank(V)
        vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
        cu, rs = 200 + 4 * freq, 1.0 + (v % 16)
        vb.render(0, freq, cu, rs, np.ones(1, np.int32), 1, out=mx.DeviceBuffer((1, V)))
        vf, vcu, vrs, vcoef, _ = vb._keep
        vpar, vhold = vb.env._params()
        gate = mx.DeviceBuffer.from_numpy(np.ones(B, np.int32))
        ctr = [0]

        def run(store, xcd):
            L.mxg_tune(b"voice_store", store); L.mxg_tune(b"voice_xcd", xcd)
            ctr[0] += 1
            dst = arena + ((ctr[0] % regions) * nbytes if rot else 0)
            chk(L.mxg_voice_render(0, V, B, vf.ptr, vcu.ptr, vrs.ptr, vcoef.ptr, gate.ptr, 0, vpar.ptr, vhold.ptr, vb.osc_state.ptr,
                                   vb.flt_state.ptr, vb.env.dstate.ptr, vb.env.istate.ptr, dst, None), "voice")
        for _ in range(20):
            run(0, 0)
        res = {m[0]: [] for m in MODES}
        for rnd in range(args.rounds + 1):
            for name, store, xcd in MODES:
                t = timed(lambda: run(store, xcd), args.reps)
                if rnd:
                    res[name].append(t)
        med = {k: float(np.median(x)) for k, x in res.items()}
        best = min((k for k in med if k != "auto"), key=med.get)
        emit("| %d %s | " % (V, "rotated" if rot else "same") +
             " | ".join("%.1f / %.3f" % (med[m[0]] * 1e3, (8.0 + 176.0 / B) * V * B / med[m[0]] / 1e6 / 8000) for m in MODES) + " | %s |" % best)
        L.mxg_tune(b"voice_store", 0); L.mxg_tune(b"voice_xcd", 0)
        del vb
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
