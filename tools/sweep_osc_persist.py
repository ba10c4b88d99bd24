#!/usr/bin/env python3
"""tools/sweep_osc_persist.py -- K1's grid against the bank size: one wavefront per 64 (or 128) voices, the round-3 forms,
against the persistent grid of round 4 (K1p, knob osc_persist: k wavefronts per SIMD, equal shares of voices x samples).

For V = 16 384 ... 1 048 576 voices x 512 samples every listed form of mxg_osc_render is timed with the destination rotating
through a 6 GiB arena (HBM rates).  Interleaved rounds in one process, median.  Prints a markdown table: us per block and the
fraction of the 8 TB/s peak on 8 B per sample."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--voices", default="16384,32768,49152,65536,81920,98304,131072,163840,196608,262144,393216,524288,1048576")
ap.add_argument("--waveforms", default="8")
ap.add_argument("--out", default=None)
args = ap.parse_args()

L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
B = 512
ARENA = 6 << 30
arena = L.mxg_malloc(ARENA)
assert arena
chk(L.mxg_memset(arena, 0, ARENA, None), "memset")
chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
NAMES = {8: "sinebuf", 9: "sinebuf4", 2: "saw", 0: "sinewave", 3: "saw"}
# (label, osc_persist, osc_vpl, osc_store, osc_xcd)
MODES = [("auto", 0, 0, 0, 0), ("1v 8B nt", 1, 1, 2, 1), ("1v pair sc1", 1, 1, 4, 1), ("1v pair sc1 xcd", 1, 1, 4, 2),
         ("2v sc1", 1, 2, 3, 1), ("2v sc1 xcd", 1, 2, 3, 2)]
for k, kn in ((2, "p1"), (3, "p2"), (4, "p4")):
    for st, sn in ((4, "pair sc1"), (5, "pair nt"), (3, "pair plain"), (2, "8B nt")):
        for xcd, xn in ((1, ""), (2, " xcd")):
            MODES.append(("%s %s%s" % (kn, sn, xn), k, 1, st, xcd))


def timed(fn, reps):
    chk(L.mxg_event_record(e0, None), "rec")
    for _ in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps


lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


emit("# K1 grid forms by bank size (MI355X, 512-sample blocks, destination rotated over a 6 GiB arena; us per block / fraction of 8 TB/s on 8 B per sample)")
emit()
emit("`python tools/sweep_osc_persist.py`: %d rounds x %d launches, median.  `auto` = the library's own choice (every knob 0); "
     "pK = persistent grid, K wavefronts per SIMD." % (args.rounds, args.reps))
for wf in [int(x) for x in args.waveforms.split(",")]:
    emit()
    emit("## %s" % NAMES.get(wf, str(wf)))
    emit()
    emit("| voices | " + " | ".join(m[0] for m in MODES) + " | best | best persistent |")
    emit("|---|" + "---|" * (len(MODES) + 2))
    for V in [int(x) for x in args.voices.split(",")]:
        nbytes = V * B * 8
        regions = max(1, ARENA // nbytes)
        ctr = [0]
        freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
        phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)

        def run(per, vpl, store, xcd):
            L.mxg_tune(b"osc_persist", per); L.mxg_tune(b"osc_vpl", vpl); L.mxg_tune(b"osc_store", store); L.mxg_tune(b"osc_xcd", xcd)
            ctr[0] += 1
            dst = arena + (ctr[0] % regions) * nbytes
            chk(L.mxg_osc_render(wf, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, dst, None), "render")
        res = {m[0]: [] for m in MODES}
        for rnd in range(args.rounds + 1):
            for name, per, vpl, store, xcd in MODES:
                t = timed(lambda: run(per, vpl, store, xcd), args.reps)
                if rnd:
                    res[name].append(t)
        med = {k: float(np.median(v)) for k, v in res.items()}
        best = min((k for k in med if k != "auto"), key=med.get)
        bestp = min((k for k in med if k[0] == "p"), key=med.get)
        emit("| %d | " % V + " | ".join("%.1f / %.3f" % (med[m[0]] * 1e3, nbytes / med[m[0]] / 1e6 / 8000) for m in MODES) +
             " | %s | %s |" % (best, bestp))
        for kk in (b"osc_persist", b"osc_vpl", b"osc_store", b"osc_xcd"):
            L.mxg_tune(kk, 0)
        del freq, phase, hold
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
