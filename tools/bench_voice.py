#!/usr/bin/env python3
"""tools/bench_voice.py -- config 3: fused subtractive voice (saw->lores->adsr), 65 536 voices, block 512.

Two figures per mode:
  sequence : SURVEY 8(d) config 3 as written -- K=128 consecutive blocks (65 536 samples) from a fresh bank,
             gate(n) = (n mod 44100) < 22050, so the run contains two attacks/decays, sustain and a release;
  sustain  : the steady state (every voice in sustain, gate held) launched back to back.
"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B, K = 65536, 512, 128
v = np.arange(V)
freq = np.minimum(20 + v * 0.30517578125, 5000.0); cutoff = 200 + 4 * freq; res = 1.0 + (v % 16)
out = mx.DeviceBuffer((B, V), zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
gate = mx.DeviceBuffer.from_numpy(((np.arange(K * B) % 44100) < 22050).astype(np.int32))
res_json = {}


def stats(us):
    return {"us_per_block": round(us, 2), "Msamples_per_s": round(V * B / us, 1),
            "algorithmic_GBs_8.34B": round(8.34 * V * B / us / 1e3, 1), "frac_of_8TBs": round(8.34 * V * B / us / 1e3 / 8000, 3)}


for mode, name in ((0, "modeA_hoisted"), (1, "modeB_modulated")):
    cu = cutoff if mode == 0 else np.full(V, 10000.0)

    def fresh():
        vb = mx.maxiVoiceBank(V)
        vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
        vb.render(mode, freq, cu, res, np.zeros(1, np.int32), 1, out=out)   # uploads parameters (one idle sample)
        return vb

    def raw(vb, trig_ptr):
        f, dcu, drs, coef, _ = vb._keep
        dpar, dhold = vb.env._params()
        L.mxg_voice_render(mode, V, B, f.ptr, dcu.ptr, drs.ptr, coef.ptr if coef is not None else None, trig_ptr, 0,
                           dpar.ptr, dhold.ptr, vb.osc_state.ptr, vb.flt_state.ptr, vb.env.dstate.ptr,
                           vb.env.istate.ptr, out.ptr, None)

    # config-3 sequence, repeated from fresh state
    tot, reps = 0.0, 3
    for r in range(reps + 1):
        vb = fresh()
        L.mxg_sync()
        L.mxg_event_record(e0, None)
        for k in range(K):
            raw(vb, gate.ptr + 4 * k * B)
        L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        if r:
            tot += ms.value
    res_json[name + "_sequence"] = stats(tot / reps / K * 1e3)
    # steady sustain: the state the sequence is in at block 20 (gate on, decay finished)
    vb = fresh()
    for k in range(20):
        raw(vb, gate.ptr + 4 * k * B)
    n = 300
    for _ in range(50):
        raw(vb, gate.ptr + 4 * 20 * B)
    L.mxg_event_record(e0, None)
    for _ in range(n):
        raw(vb, gate.ptr + 4 * 20 * B)
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    res_json[name + "_sustain"] = stats(ms.value / n * 1e3)
print(json.dumps(res_json))
