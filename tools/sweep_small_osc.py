#!/usr/bin/env python3
"""tools/sweep_small_osc.py -- small banks of the table oscillators: one part against the automatic time parts (mxg_osc_render and
mxg_osc_render_mix, 512-sample blocks, MI355X).  A block is one chain of 512 dependent steps per wavefront whatever the bank size;
banks with fewer wavefronts than the machine has SIMDs are cut along time (part p skips to its first sample with the same
additions: same bits).  Output: a markdown table (stdout and --out)."""
import argparse
import ctypes
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
B = 512


def timed(fn, reps):
    for _ in range(5):
        fn()
    chk(L.mxg_event_record(e0, None), "rec")
    for _ in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps * 1e3


rows = []
for wfname in ("sinebuf", "sinebuf4", "sawn", "sinewave"):
    wf = mx.OSC_WAVEFORMS[wfname]
    for V in (64, 256, 1024, 4096, 16384, 32768):
        rng = np.random.default_rng(V)
        freq = rng.uniform(20, 15000, V)
        pan = rng.uniform(0, 1, V)
        res = {}
        DB = mx.DeviceBuffer
        d_f, d_p = DB.from_numpy(freq), DB.from_numpy(pan)
        d_ph, d_hd = DB(V), DB(V)
        d_out, d_mix = DB((B, V), np.float64, zero=False), DB((B, 2), np.float64, zero=False)
        for mix in (False, True):
            if mix:
                fn = lambda: chk(L.mxg_osc_render_mix(wf, V, B, d_f.ptr, None, None, d_ph.ptr, d_hd.ptr, d_out.ptr, d_p.ptr, d_mix.ptr, None), "mix")
            else:
                fn = lambda: chk(L.mxg_osc_render(wf, V, B, d_f.ptr, 0, None, None, d_ph.ptr, d_hd.ptr, d_out.ptr, None), "render")
            knob = b"osc_mix_split" if mix else b"osc_split"
            for label, val in (("one", 1), ("auto", 0)):
                prev = L.mxg_tune(knob, val)
                ts = [timed(fn, args.reps) for _ in range(args.rounds)]
                L.mxg_tune(knob, prev)
                res[(mix, label)] = statistics.median(ts)
        rows.append((wfname, V, res))
lines = ["# Small banks of the table oscillators: one part against the automatic time parts (MI355X, 512-sample blocks; us per C-ABI call, back to back on one stream)", "",
         "`python tools/sweep_small_osc.py` (median of %d rounds x %d calls)." % (args.rounds, args.reps), "",
         "| waveform | voices | render: one part | render: automatic | render + mixdown: one part | render + mixdown: automatic |", "|---|---|---|---|---|---|"]
for wfname, V, r in rows:
    lines.append("| %s | %d | %.1f | %.1f | %.1f | %.1f |" % (wfname, V, r[(False, "one")], r[(False, "auto")], r[(True, "one")], r[(True, "auto")]))
text = "\n".join(lines) + "\n"
print(text)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write(text)
