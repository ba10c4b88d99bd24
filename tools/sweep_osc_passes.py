#!/usr/bin/env python3
"""tools/sweep_osc_passes.py -- K1 / K1m over bank sizes: how many voice groups a wavefront renders one after the other (knobs
osc_passes / osc_mix_passes), time parts (osc_mix_split) and the store stream.  Destination rotated over a 6 GiB arena; interleaved
rounds, median; us per block and the fraction of the 8 TB/s peak on 8 B per sample."""
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
ap.add_argument("--voices", default="65536,81920,98304,131072,163840,196608,262144,393216,524288,1048576")
ap.add_argument("--mix-voices", default="65536,98304,131072,262144")
ap.add_argument("--wf", type=int, default=8)
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


def sweep(V, modes, run):
    res = {m[0]: [] for m in modes}
    for rnd in range(args.rounds + 1):
        for m in modes:
            t = timed(lambda: run(*m[1:]), args.reps)
            if rnd:
                res[m[0]].append(t)
    return {k: float(np.median(v)) for k, v in res.items()}


emit("# K1 / K1m: passes, time parts and store streams by bank size (MI355X, 512-sample blocks, rotated destination)")
emit()
emit("us per block / fraction of 8 TB/s on 8 B per sample; waves = wavefronts of the grid.")
emit()
emit("## K1 (mxg_osc_render, waveform %d)" % args.wf)
emit()
for V in [int(x) for x in args.voices.split(",") if x]:
    nbytes = V * B * 8
    regions = max(1, ARENA // nbytes)
    ctr = [0]
    freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    modes = [("auto", 0, 0, 0, 0)]
    for vpl, store, sname in ((1, 4, "1v pair sc1"), (2, 3, "2v sc1"), (1, 2, "1v 8B nt")):
        waves = V // (64 * vpl)
        for passes in (1, 2, 3, 4, 6, 8, 12, 16, 32):
            if waves // passes < 384 and passes > 1:
                continue
            for xcd in (1, 2):
                modes.append(("%s p%d%s (%d waves)" % (sname, passes, " xcd" if xcd == 2 else "", -(-waves // passes)), vpl, store, xcd, passes))

    def run(vpl, store, xcd, passes):
        L.mxg_tune(b"osc_vpl", vpl); L.mxg_tune(b"osc_store", store); L.mxg_tune(b"osc_xcd", xcd); L.mxg_tune(b"osc_passes", passes)
        ctr[0] += 1
        chk(L.mxg_osc_render(args.wf, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, arena + (ctr[0] % regions) * nbytes, None), "render")
    med = sweep(V, modes, run)
    order = sorted(med, key=med.get)
    emit("**%d voices**: auto %.1f us (%.3f); best five: " % (V, med["auto"] * 1e3, nbytes / med["auto"] / 1e6 / 8000) +
         "; ".join("%s %.1f (%.3f)" % (k, med[k] * 1e3, nbytes / med[k] / 1e6 / 8000) for k in order[:5]))
    emit()
    emit("| form | us | frac |")
    emit("|---|---|---|")
    for k in med:
        emit("| %s | %.1f | %.3f |" % (k, med[k] * 1e3, nbytes / med[k] / 1e6 / 8000))
    emit()
    for kk in (b"osc_vpl", b"osc_store", b"osc_xcd", b"osc_passes"):
        L.mxg_tune(kk, 0)
    del freq, phase, hold

emit("## K1m (mxg_osc_render_mix_rows, waveform %d, per-voice block stored)" % args.wf)
emit()
for V in [int(x) for x in args.mix_voices.split(",") if x]:
    nbytes = V * B * 8
    regions = max(1, ARENA // nbytes)
    ctr = [0]
    freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
    pan = mx.DeviceBuffer.from_numpy(np.arange(V) / (V - 1.0))
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    G = L.mxg_osc_mix_groups(V)
    rows = mx.DeviceBuffer((G, B, 2))
    modes = [("auto", 0, 0, 0, 1), ("auto mix-only", 0, 0, 0, 0)]
    for win in (128, 256):
        for split in (1, 2, 3, 4):
            modes.append(("win %d split %d" % (win, split), win, split, 1, 1))
        for passes in (2, 3, 4, 8):
            if G // passes >= 96:
                modes.append(("win %d passes %d (%d waves)" % (win, passes, 4 * -(-G // passes)), win, 1, passes, 1))
    modes.append(("win 256 split 2 mix-only", 256, 2, 1, 0))

    def run(win, split, passes, store):
        L.mxg_tune(b"osc_mix_win", win); L.mxg_tune(b"osc_mix_split", split); L.mxg_tune(b"osc_mix_passes", passes)
        ctr[0] += 1
        chk(L.mxg_osc_render_mix_rows(args.wf, V, B, freq.ptr, None, None, phase.ptr, hold.ptr,
                                      arena + (ctr[0] % regions) * nbytes if store else None, pan.ptr, rows.ptr, None), "render_mix_rows")
    med = sweep(V, modes, run)
    emit("**%d voices**" % V)
    emit()
    emit("| form | us | frac |")
    emit("|---|---|---|")
    for k in med:
        emit("| %s | %.1f | %.3f |" % (k, med[k] * 1e3, nbytes / med[k] / 1e6 / 8000))
    emit()
    for kk in (b"osc_mix_win", b"osc_mix_split", b"osc_mix_passes"):
        L.mxg_tune(kk, 0)
    del freq, phase, hold, pan, rows
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
