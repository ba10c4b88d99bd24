#!/usr/bin/env python3
"""tools/sweep_rw_store.py -- the read + write bank kernels (8 B in + 8 B out per sample) with 8-byte streams and with the 16-byte
pair-row streams (knob rw_store), input and output blocks rotating through a 6 GiB arena each.  us per 512-sample block and the
fraction of 8 TB/s on 16 B per sample."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--voices", default="32768,65536,131072,262144")
ap.add_argument("--out", default=None)
args = ap.parse_args()
L = mx.lib()
chk = mx._lib.check
chk(L.mxg_init(0), "init")
mx.maxiSettings.setup(44100, 2, 1024)
B = 512
ARENA = 4 << 30
a_in, a_out = L.mxg_malloc(ARENA), L.mxg_malloc(ARENA)
chk(L.mxg_memset(a_in, 0, ARENA, None), "memset"); chk(L.mxg_memset(a_out, 0, ARENA, None), "memset"); chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create()
ms = ctypes.c_float()
MODES = [("auto", 0), ("8-byte", 1), ("pairs plain", 2), ("pairs sc1", 3), ("pairs nt", 4)]
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


def timed(fn, reps=8):
    chk(L.mxg_event_record(e0, None), "rec")
    for _ in range(reps):
        fn()
    chk(L.mxg_event_record(e1, None), "rec")
    chk(L.mxg_event_sync(e1), "sync")
    chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
    return ms.value / reps


emit("# Read + write bank kernels: 8-byte vs 16-byte pair-row streams (MI355X, 512-sample blocks; us / fraction of 8 TB/s on 16 B per sample)")
emit()
emit("| kernel | voices | " + " | ".join(m[0] for m in MODES) + " |")
emit("|---|---|" + "---|" * len(MODES))
for name, kind in (("maxiBiquad", 2), ("maxiSVF", 1), ("maxiDCBlocker", 0)):
    for V in [int(x) for x in args.voices.split(",")]:
        nb = V * B * 8
        regions = ARENA // nb
        k = [0]
        if kind == 2:
            b = mx.maxiBiquadBank(V); b.set(np.zeros(V, np.int32), np.full(V, 1200.0), np.full(V, 0.7), np.zeros(V))
            b.play(mx.DeviceBuffer((8, V)))
        elif kind == 1:
            b = mx.maxiSVFBank(V); b.setCutoff(800.0); b.setResonance(2.0); b.play(mx.DeviceBuffer((8, V)), 0.5, 0.25, 0.1, 0.1)
        else:
            b = mx.maxiDCBlockerBank(V); b.play(mx.DeviceBuffer((8, V)), 0.995)

        def run():
            k[0] += 1
            r = (k[0] % regions) * nb
            chk(L.mxg_filter2_render(kind, V, B, a_in + r, b.coef.ptr, b.state.ptr, a_out + r, None), "f2")
        res = {}
        for rnd in range(6):
            for nm, rw in MODES:
                L.mxg_tune(b"rw_store", rw)
                t = timed(run)
                if rnd:
                    res.setdefault(nm, []).append(t)
        L.mxg_tune(b"rw_store", 0)
        emit("| %s | %d | " % (name, V) + " | ".join("%.1f / %.3f" % (np.median(res[m[0]]) * 1e3, 2 * nb / np.median(res[m[0]]) / 1e6 / 8000) for m in MODES) + " |")

# ---- round 4: the other bank kernels that got the pair-row streams (maxiFilter, maxiEnv, maxiDelayline, maxiEnvGen, maxiSample) ----------
emit()
emit("| kernel (B per sample) | voices | " + " | ".join(m[0] for m in MODES) + " |")
emit("|---|---|" + "---|" * len(MODES))
D = mx.DeviceBuffer.from_numpy
for V in [int(x) for x in args.voices.split(",")]:
    nb = V * B * 8
    regions = ARENA // nb
    v = np.arange(V)
    rng = np.random.default_rng(1)
    cut = 200 + 4 * np.minimum(20 + v * 0.305, 5000.0); res_ = 1.0 + (v % 16)
    coef = np.zeros((3, V)); L.mxg_filter_coeffs_host(0, V, cut.ctypes.data, res_.ctypes.data, coef.ctypes.data)
    dcut, dres, dcoef, fst, fst2 = D(cut), D(res_), D(coef), mx.DeviceBuffer((5, V)), mx.DeviceBuffer((5, V))
    dlp = D(np.full(V, 0.3))
    eb = mx.maxiEnvBank(V); eb.setAttack(10); eb.setDecay(100); eb.setSustain(0.5); eb.setRelease(500)
    dpar, dhold = eb._params()
    trig = D(((np.arange(B) % 300) < 150).astype(np.int32)); held_i = D(np.ones(B, np.int32))
    eg = mx.maxiEnvGenBank(V); eg.setupADSR(10, 100, 0.5, 500)
    gate = D(np.where((np.arange(B) % 300) < 150, 1.0, -1.0)); held = D(np.ones(B))
    db = mx.maxiDelaylineBank(V, 2048); dsz, dfb = D(np.full(V, 1024, np.int32)), D(np.full(V, 0.5))
    sb = mx.maxiSampleBank(V); sb.setSample(rng.uniform(-1, 1, 441000)); sb.setPosition(v / V)
    dsp = D(0.5 + (v % 97) / 96.0)
    sb2 = mx.maxiSampleBank(V); sb2.setSample(rng.uniform(-1, 1, 441000)); sb2.position.upload(np.floor(v / V * 400000.0))
    cases = [
        ("maxiFilter lores (16)", 16, lambda i, o: L.mxg_filter_render(0, V, B, i, dcut.ptr, 0, dres.ptr, 0, dcoef.ptr, fst.ptr, o, None)),
        ("maxiFilter lopass (16)", 16, lambda i, o: L.mxg_filter_render(3, V, B, i, dlp.ptr, 0, None, 0, None, fst2.ptr, o, None)),
        ("maxiEnv adsr, gate 150/150 (16)", 16, lambda i, o: L.mxg_env_render(0, V, B, i, trig.ptr, 0, dpar.ptr, dhold.ptr, eb.dstate.ptr, eb.istate.ptr, o, None)),
        ("maxiEnv adsr, sustain (16)", 16, lambda i, o: L.mxg_env_render(0, V, B, i, held_i.ptr, 0, dpar.ptr, dhold.ptr, eb.dstate.ptr, eb.istate.ptr, o, None)),
        ("maxiEnvGen ADSR, gate 150/150 (8)", 8, lambda i, o: L.mxg_envgen_render(V, B, gate.ptr, 0, eg.stages.ptr, 4, 0, 0, eg.dstate.ptr, eg.istate.ptr, o, None)),
        ("maxiEnvGen ADSR, holding (8)", 8, lambda i, o: L.mxg_envgen_render(V, B, held.ptr, 0, eg.stages.ptr, 4, 0, 0, eg.dstate.ptr, eg.istate.ptr, o, None)),
        ("maxiDelayline dl, size 1024 (32)", 32, lambda i, o: L.mxg_delay_render(0, V, B, i, dsz.ptr, dfb.ptr, None, db.memory.ptr, 2048, db.phase.ptr, o, None)),
        ("maxiSample playAtSpeed (8)", 8, lambda i, o: L.mxg_sample_render(4, V, B, sb.d_samples, sb.length, 44100, dsp.ptr, 0, None, None, sb.position.ptr, o, None)),
        ("maxiSample play, integer heads (8)", 8, lambda i, o: L.mxg_sample_render(0, V, B, sb2.d_samples, sb2.length, 44100, None, 0, None, None, sb2.position.ptr, o, None)),
    ]
    for name, bps, call in cases:
        k = [0]

        def run():
            k[0] += 1
            r = (k[0] % regions) * nb
            chk(call(a_in + r, a_out + r), name)
        for _ in range(30):
            run()
        res = {}
        for rnd in range(6):
            for nm, rw in MODES:
                L.mxg_tune(b"rw_store", rw)
                t = timed(run)
                if rnd:
                    res.setdefault(nm, []).append(t)
        L.mxg_tune(b"rw_store", 0)
        emit("| %s | %d | " % (name, V) + " | ".join("%.1f / %.3f" % (np.median(res[m[0]]) * 1e3, bps * V * B / np.median(res[m[0]]) / 1e6 / 8000) for m in MODES) + " |")
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
