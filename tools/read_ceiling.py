#!/usr/bin/env python3
"""tools/read_ceiling.py -- the HBM *read* ceiling of this box for the fused FFT + MFCC kernel's input stream.

Config 4 reads 4096 bytes per frame and writes 104: its mandatory traffic is a read stream.  MI355X's spec is 8 TB/s; what a pure
LOAD stream of the kernel's own shape reaches is measured here with csrc/calib.hip (mxg_calib_read_ex): the grid-stride read of the
flat region and the frame stream (persistent wavefronts, 8 consecutive 4 KB frames per wavefront and group, the next frame in
flight), 8- and 16-byte loads, plain and non-temporal, at the kernel's launch shapes (512 workgroups x 256 threads = two per CU;
256 x 768 = the 12-wave layout) and larger ones.  The region is 1 048 576 frames = 4.3 GB (17 Infinity Caches), so nothing is
re-read from a cache.  Then mxg_fft_mfcc_batch itself (exact and tolerance mode) on the same buffer, in the same process.

Interleaved rounds, median over rounds.  Output: a markdown table (stdout and --out)."""
import argparse
import ctypes
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--frames", type=int, default=1 << 20)
ap.add_argument("--out", default=None)
args = ap.parse_args()

L = mx.lib()
CAL = mx.calib()  # measurement probes: libmaxicalib.so
chk = mx._lib.check
chk(L.mxg_init(0), "init")
NF = args.frames
NBYTES = NF * 4096
src = L.mxg_malloc(NBYTES + 4096)
sink = L.mxg_malloc(64)
assert src and sink
# a signal, not zeros: the product kernel runs on the same buffer
rng = np.random.default_rng(1)
chunk = rng.uniform(-1, 1, 1 << 22).astype(np.float32)
for off in range(0, NBYTES, chunk.nbytes):
    n = min(chunk.nbytes, NBYTES - off)
    chk(L.mxg_memcpy_h2d(src + off, chunk.ctypes.data, n, None), "h2d")
chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()


def timed(fn, reps):
    chk(L.mxg_event_record(e0, None), "rec")
    for i in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps


variants = []
for pattern, pname in ((1, "frame stream"), (0, "grid-stride")):
    for width in (8, 16):
        for flav, fname in ((0, "plain"), (1, "nt")):
            for block, blocks in ((256, 512), (768, 256), (256, 1024), (256, 2048), (512, 1024)):
                name = "%s, %d-byte %s loads, %d x %d threads" % (pname, width, fname, blocks, block)
                variants.append((name, lambda p=pattern, w=width, f=flav, b=block, g=blocks:
                                 chk(CAL.mxg_calib_read_ex(src, NBYTES, w, f, p, b, g, sink, None), "read")))

f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
m = mx.maxiMFCC(); m.setup(512, 42, 13, 20.0, 20000.0)
d_mfcc = L.mxg_malloc(NF * 13 * 8)
assert d_mfcc


def product(exact, layout):
    def run():
        p1 = L.mxg_tune(b"fft_exact", exact); p2 = L.mxg_tune(b"fused_layout", layout)
        chk(L.mxg_fft_mfcc_batch(f.plan, m.plan, src, 1024, NF, None, None, None, d_mfcc, None), "mxg_fft_mfcc_batch")
        L.mxg_tune(b"fft_exact", p1); L.mxg_tune(b"fused_layout", p2)
    return run


for exact, ename in ((1, "exact"), (0, "tolerance mode")):
    for layout, lname in ((1, "two frames in flight, 8 waves per CU"), (2, "one frame in flight, 12 waves per CU")):
        variants.append(("mxg_fft_mfcc_batch, %s, %s" % (ename, lname), product(exact, layout)))

res = {name: [] for name, _ in variants}
for name, fn in variants:
    fn()  # warm (plans, code objects)
chk(L.mxg_sync(), "sync")
for r in range(args.rounds):
    for name, fn in variants:
        res[name].append(timed(fn, args.reps))
lines = ["# The HBM read ceiling for config 4's input stream (MI355X; %d frames x 4096 B = %.2f GB per pass; ms, median of %d rounds x %d passes)"
         % (NF, NBYTES / 1e9, args.rounds, args.reps), "",
         "Produced by `python tools/read_ceiling.py` (csrc/calib.hip, `mxg_calib_read_ex`).  `TB/s` counts the 4096 input bytes per frame only "
         "(the product kernel also writes 104 B per frame: 4200 B algorithmic).", "",
         "| stream | ms | TB/s | of 8 TB/s |", "|---|---|---|---|"]
best = None
for name, _ in variants:
    t = statistics.median(res[name])
    tb = NBYTES / (t * 1e-3) / 1e12
    lines.append("| %s | %.4f | %.2f | %.3f |" % (name, t, tb, tb / 8.0))
    if not name.startswith("mxg_") and (best is None or t < best[1]):
        best = (name, t)
lines += ["", "Fastest pure load stream: **%s: %.4f ms = %.2f TB/s = %.3f of the spec**." % (best[0], best[1], NBYTES / (best[1] * 1e-3) / 1e12,
                                                                                           NBYTES / (best[1] * 1e-3) / 8e12)]
for name, _ in variants:
    if name.startswith("mxg_"):
        t = statistics.median(res[name])
        lines.append("`%s`: %.4f ms = %.3f of that ceiling." % (name, t, best[1] / t))
text = "\n".join(lines) + "\n"
print(text)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write(text)
