#!/usr/bin/env python3
"""tools/sweep_placement.py -- does the RELATIVE placement of a kernel's read stream and write stream matter?  One pool, the input
block at its 2 MB-aligned start, the output block after it at a 2 MB-aligned address plus a byte offset; filter (lores, hoisted
coefficients: 8 B in + 8 B out per sample) at 65 536 voices x 512 timed for each offset."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
nbytes = V * B * 8
MB2 = 2 << 20
L.mxg_malloc.restype = ctypes.c_void_p
POOL_EXTRA = int(os.environ.get('POOL_GB', '5')) << 30
pool = L.mxg_malloc(ctypes.c_size_t(2 * nbytes + POOL_EXTRA))
assert pool, 'pool allocation failed'
base = (pool + MB2 - 1) // MB2 * MB2
rng = np.random.default_rng(1)
xin = rng.uniform(-1, 1, (B, V))
L.mxg_memcpy_h2d(ctypes.c_void_p(base), xin.ctypes.data, ctypes.c_size_t(nbytes), None)
v = np.arange(V)
cut = 200 + 4 * np.minimum(20 + v * 0.305, 5000.0); res = 1.0 + (v % 16)
coef = np.zeros((3, V)); L.mxg_filter_coeffs_host(0, V, cut.ctypes.data, res.ctypes.data, coef.ctypes.data)
D = mx.DeviceBuffer.from_numpy
dcut, dres, dcoef, fst = D(cut), D(res), D(coef), mx.DeviceBuffer((5, V))
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
def timed(fn, reps=150):
    for _ in range(60): fn()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3
obase = base + (nbytes + MB2 - 1) // MB2 * MB2
OFFS = [0, 256, 1024, 2048, 4096, 1 << 17, 1 << 18, 1 << 19, 1 << 24, 1 << 25, 1 << 26, (1 << 25) + (1 << 18), (1 << 25) + (1 << 11), (1 << 25) + (1 << 18) + (1 << 11),
        (1 << 18) + (1 << 11), 1 << 31, 1 << 32, (1 << 32) + (1 << 25), 3 << 11, 1 << 12, 1 << 10]
if len(sys.argv) > 1:
    OFFS = [int(a, 0) for a in sys.argv[1:]]
for off in OFFS:
    if off + nbytes > nbytes + POOL_EXTRA - (64 << 20):
        continue   # beyond the pool
    out = ctypes.c_void_p(obase + off)
    t = timed(lambda: L.mxg_filter_render(0, V, B, ctypes.c_void_p(base), dcut.ptr, 0, dres.ptr, 0, dcoef.ptr, fst.ptr, out, None))
    print("out = in + %d MB + %8d B   filter lores %.1f us  (%.0f GB/s of 16 B/sample)" % ((obase - base) >> 20, off, t, 16 * V * B / t / 1e3), flush=True)

# ---- separately allocated blocks: what a host gets from hipMalloc as it comes (arguments: none) ---------------------------------------
if len(sys.argv) == 1:
    L.mxg_free(ctypes.c_void_p(pool))
    times, keep = [], []
    for trial in range(12):
        a = L.mxg_malloc(ctypes.c_size_t(nbytes)); b = L.mxg_malloc(ctypes.c_size_t(nbytes))
        keep.append(L.mxg_malloc(ctypes.c_size_t((trial % 5 + 1) * (33 << 20))))   # odd-sized neighbours: vary where the next pair lands
        L.mxg_memcpy_h2d(ctypes.c_void_p(a), xin.ctypes.data, ctypes.c_size_t(nbytes), None)
        t0 = timed(lambda: L.mxg_filter_render(0, V, B, ctypes.c_void_p(a), dcut.ptr, 0, dres.ptr, 0, dcoef.ptr, fst.ptr, ctypes.c_void_p(b), None), 60)
        times.append(t0)
        L.mxg_free(ctypes.c_void_p(a)); L.mxg_free(ctypes.c_void_p(b))
    for k in keep: L.mxg_free(ctypes.c_void_p(k))
    print("filter lores over 12 separately allocated pairs of blocks: %s us" % " ".join("%.0f" % t for t in times), flush=True)
