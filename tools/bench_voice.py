#!/usr/bin/env python3
"""tools/bench_voice.py -- config 3: fused subtractive voice (saw->lores->adsr), 65 536 voices, block 512."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
v = np.arange(V)
freq = np.minimum(20 + v * 0.30517578125, 5000.0); cutoff = 200 + 4 * freq; res = 1.0 + (v % 16)
vb = mx.maxiVoiceBank(V)
vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
out = mx.DeviceBuffer((B, V), zero=False)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
res_json = {}
for mode, name in ((0, "modeA_hoisted"), (1, "modeB_modulated")):
    cu = cutoff if mode == 0 else np.full(V, 10000.0)
    blk = [0]
    def step():
        n0 = blk[0] * B
        trig = ((np.arange(n0, n0 + B) % 44100) < 22050).astype(np.int32)
        vb.render(mode, freq, cu, res, trig, B, out=out); blk[0] += 1
    # steady-state launches reuse device-side parameters: time the raw C-ABI call
    step()
    f, dcu, drs, coef, trig = vb._keep
    dpar, dhold = vb.env._params()
    def raw():
        L.mxg_voice_render(mode, V, B, f.ptr, dcu.ptr, drs.ptr, coef.ptr if coef is not None else None, trig.ptr, 0, dpar.ptr, dhold.ptr,
                           vb.osc_state.ptr, vb.flt_state.ptr, vb.env.dstate.ptr, vb.env.istate.ptr, out.ptr, None)
    for _ in range(200): raw()
    reps = 500 if mode == 0 else 50
    L.mxg_event_record(e0, None)
    for _ in range(reps): raw()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    us = ms.value / reps * 1e3
    res_json[name] = {"us_per_block": round(us, 2), "Msamples_per_s": round(V * B / us, 1),
                      "algorithmic_GBs_8.34B": round(8.34 * V * B / us / 1e3, 1), "frac_of_8TBs": round(8.34 * V * B / us / 1e3 / 8000, 3)}
print(json.dumps(res_json))
