#!/usr/bin/env python3
"""tools/write_ceiling.py -- the HBM *write* ceiling of this box, and K1 (maxiOsc bank render) against it.

Every bank kernel's mandatory traffic is the out[n*V + v] store stream (8 B per sample).  MI355X's spec is 8 TB/s; what a
pure store stream reaches is measured here with the calibration family of csrc/calib.hip (mxg_calib_fill_ex): grid-stride
fills and column walks (K1's own shape with the arithmetic removed), 8- and 16-byte stores, six store flavours, 256 / 512 /
1024-thread workgroups, natural and XCD-contiguous workgroup numbering -- for blocks of 65 536 ... 1 048 576 voices x 512
samples (268 MB ... 4.3 GB).  The destination ROTATES through an 8 GiB arena, so a line is rewritten only after >= 8 GiB of
other stores (32x the 256 MB Infinity Cache); `same` rows rewrite one region instead (what a block renderer that reuses its
block buffer sees) to show the cache's share.  K1 rows are mxg_osc_render(sinebuf) on the same regions with its knobs.

Interleaved rounds in one process, median over rounds.  Output: a markdown table (stdout and --out)."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--arena-gib", type=float, default=8.0)
ap.add_argument("--voices", default="65536,131072,262144,1048576")
ap.add_argument("--out", default=None)
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()

L = mx.lib()
CAL = mx.calib()  # measurement probes: libmaxicalib.so
chk = mx._lib.check
chk(L.mxg_init(0), "init")
B = 512
ARENA = int(args.arena_gib * (1 << 30))
arena = L.mxg_malloc(ARENA)
assert arena, "arena"
chk(L.mxg_memset(arena, 0, ARENA, None), "memset")
chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
FLAV = {0: "plain", 1: "nt", 2: "sc1", 3: "sc0 sc1", 4: "sc1 nt", 5: "sc0"}


def timed(fn, reps):
    chk(L.mxg_event_record(e0, None), "rec")
    for i in range(reps):
        fn(i)
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps


def variants_for(V):
    nbytes = V * B * 8
    regions = max(1, ARENA // nbytes)
    ctr = [0]

    def region(rot):
        if not rot:
            return arena
        ctr[0] += 1
        return arena + (ctr[0] % regions) * nbytes
    out = {}

    def fill(name, width, flav, pattern, block, blocks=0, xcd=0, rot=True):
        out[name] = lambda i: chk(CAL.mxg_calib_fill_ex(region(rot), B, V * 8, width, flav, pattern, block, blocks, xcd, None), name)

    # grid-stride fills
    for w in (8, 16):
        for fl in ((0, 1, 2, 3) if not args.quick else (0, 1)):
            fill("flat w%d %s 2048x256" % (w, FLAV[fl]), w, fl, 0, 256, 2048)
    fill("flat w16 plain 1024x256", 16, 0, 0, 256, 1024)
    fill("flat w16 plain 4096x256", 16, 0, 0, 256, 4096)
    fill("flat w16 plain 16384x256", 16, 0, 0, 256, 16384)
    fill("flat w16 plain 2048x256 same", 16, 0, 0, 256, 2048, rot=False)
    # column walks (K1's shape)
    for w in (8, 16):
        for fl in ((0, 1, 2, 3, 4, 5) if not args.quick else (0, 1)):
            fill("cols w%d %s blk256" % (w, FLAV[fl]), w, fl, 1, 256)
    for blk in (128, 512, 1024):
        fill("cols w8 plain blk%d" % blk, 8, 0, 1, blk)
    for blk in (256,):
        fill("cols w8 plain blk%d xcd" % blk, 8, 0, 1, blk, xcd=1)
        fill("cols w16 plain blk%d xcd" % blk, 16, 0, 1, blk, xcd=1)
        fill("cols w16 sc1 blk%d xcd" % blk, 16, 2, 1, blk, xcd=1)
    fill("cols w16 sc1 blk512", 16, 2, 1, 512)
    fill("cols w16 plain blk512", 16, 0, 1, 512)
    fill("cols w8 plain blk256 halves", 8, 0, 2, 256)
    fill("cols w8 plain blk256 same", 8, 0, 1, 256, rot=False)
    fill("cols w8 nt blk256 same", 8, 1, 1, 256, rot=False)

    # K1 itself
    freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    keep = (freq, phase, hold)

    STORE1 = {1: "8B plain", 2: "8B nt", 3: "pair-rows 16B plain", 4: "pair-rows 16B sc1", 5: "pair-rows 16B nt"}
    STORE2 = {1: "16B plain", 2: "16B nt", 3: "16B sc1"}

    def k1(name, vpl, store, blk, xcd=1, rot=True, wf=8):
        def f(i):
            L.mxg_tune(b"osc_vpl", vpl); L.mxg_tune(b"osc_store", store); L.mxg_tune(b"osc_block", blk); L.mxg_tune(b"osc_xcd", xcd)
            chk(L.mxg_osc_render(wf, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, region(rot), None), name)
        out[name] = f
    for xcd in (1, 2):
        for st, nm in STORE1.items():
            k1("K1 sinebuf 1v/lane %s blk256%s" % (nm, " xcd" if xcd == 2 else ""), 1, st, 256, xcd)
        for st, nm in STORE2.items():
            for blk in (256, 512):
                k1("K1 sinebuf 2v/lane %s blk%d%s" % (nm, blk, " xcd" if xcd == 2 else ""), 2, st, blk, xcd)
    k1("K1 sinebuf 1v/lane pair-rows 16B sc1 blk128", 1, 4, 128, 1)
    k1("K1 sinebuf 1v/lane pair-rows 16B plain blk512", 1, 3, 512, 1)
    k1("K1 sinebuf default knobs", 0, 0, 256, xcd=0)
    k1("K1 sinebuf default knobs same", 0, 0, 256, xcd=0, rot=False)
    k1("K1 sinebuf4 default knobs", 0, 0, 256, xcd=0, wf=9)
    k1("K1 saw default knobs", 0, 0, 256, xcd=0, wf=2)
    return out, nbytes, keep


lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


emit("# HBM write ceiling and K1 against it (MI355X, block = 512 samples, fp64)")
emit()
emit("`python tools/write_ceiling.py` -- arena %.0f GiB, destination rotated region by region unless marked `same`; "
     "%d interleaved rounds x %d launches, median.  GB/s = V x 512 x 8 B / time." % (args.arena_gib, args.rounds, args.reps))
summary = []
for V in [int(x) for x in args.voices.split(",")]:
    var, nbytes, keep = variants_for(V)
    res = {k: [] for k in var}
    for rnd in range(args.rounds + 1):
        for k, f in var.items():
            t = timed(f, args.reps)
            if rnd:
                res[k].append(t)
    emit()
    emit("## %d voices (%.0f MB per block)" % (V, nbytes / 1e6))
    emit()
    emit("| variant | us (median) | us (min) | GB/s | of 8 TB/s |")
    emit("|---|---|---|---|---|")
    best_fill, best_k1 = None, None
    for k, ts in res.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        gbs = nbytes / med / 1e6
        emit("| %s | %.1f | %.1f | %.0f | %.3f |" % (k, med * 1e3, mn * 1e3, gbs, gbs / 8000))
        if "same" in k:
            continue
        if k.startswith("K1 sinebuf 1v") or k.startswith("K1 sinebuf 2v") or k == "K1 sinebuf default knobs":
            if best_k1 is None or med < best_k1[1]:
                best_k1 = (k, med)
        elif not k.startswith("K1"):
            if best_fill is None or med < best_fill[1]:
                best_fill = (k, med)
    d = float(np.median(res["K1 sinebuf default knobs"]))
    summary.append((V, nbytes, best_fill, best_k1, d))
    # knobs back to their defaults
    L.mxg_tune(b"osc_vpl", 0); L.mxg_tune(b"osc_store", 0); L.mxg_tune(b"osc_block", 256); L.mxg_tune(b"osc_xcd", 0)
    del keep

emit()
emit("## Summary (rotated destinations only)")
emit()
emit("| voices | best pure store stream | GB/s | K1 best knobs | GB/s | K1 default | GB/s | K1 default / ceiling |")
emit("|---|---|---|---|---|---|---|---|")
for V, nbytes, bf, bk, d in summary:
    emit("| %d | %s | %.0f | %s | %.0f | %.1f us | %.0f | %.3f |" % (V, bf[0], nbytes / bf[1] / 1e6, bk[0], nbytes / bk[1] / 1e6,
                                                                    d * 1e3, nbytes / d / 1e6, bf[1] / d))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
