#!/usr/bin/env python3
"""tools/sweep_osc_plan.py -- K1 on large banks: the plan of launches (knob osc_plan) against single launches; rotated destination."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx  # noqa: E402
L = mx.lib(); chk = mx._lib.check; chk(L.mxg_init(0), "init")
B = 512; ARENA = 8 << 30
arena = L.mxg_malloc(ARENA); assert arena
chk(L.mxg_memset(arena, 0, ARENA, None), "memset"); chk(L.mxg_sync(), "sync")
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "196608,262144,327680,393216,458752,524288,655360,786432,1048576").split(",")]
MODES = [("auto", {}), ("single (plan off)", {"osc_plan": 1}), ("plan", {"osc_plan": 2}), ("plan xcd", {"osc_plan": 3}),
         ("2v sc1 xcd p1", {"osc_vpl": 2, "osc_store": 3, "osc_xcd": 2}), ("1v 8B nt xcd p8", {"osc_vpl": 1, "osc_store": 2, "osc_xcd": 2, "osc_passes": 8})]
KNOBS = [b"osc_plan", b"osc_vpl", b"osc_store", b"osc_xcd", b"osc_passes"]
print("| voices | " + " | ".join(m[0] for m in MODES) + " |"); print("|---|" + "---|" * len(MODES))
for V in sizes:
    nbytes = V * B * 8; regions = max(1, ARENA // nbytes); ctr = [0]
    freq = mx.DeviceBuffer.from_numpy(20.0 + (np.arange(V) % 65536) * 0.30517578125)
    phase, hold = mx.DeviceBuffer(V), mx.DeviceBuffer(V)
    res = {m[0]: [] for m in MODES}
    for rnd in range(6):
        for name, kn in MODES:
            for k in KNOBS: L.mxg_tune(k, 0)
            for k, v in kn.items(): L.mxg_tune(k.encode(), v)
            chk(L.mxg_event_record(e0, None), "rec")
            for _ in range(5):
                ctr[0] += 1
                chk(L.mxg_osc_render(8, V, B, freq.ptr, 0, None, None, phase.ptr, hold.ptr, arena + (ctr[0] % regions) * nbytes, None), "render")
            chk(L.mxg_event_record(e1, None), "rec"); chk(L.mxg_event_sync(e1), "sync"); chk(L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "el")
            if rnd: res[name].append(ms.value / 5)
    print("| %d | " % V + " | ".join("%.1f / %.3f" % (np.median(res[m[0]]) * 1e3, nbytes / np.median(res[m[0]]) / 1e6 / 8000) for m in MODES) + " |", flush=True)
    for k in KNOBS: L.mxg_tune(k, 0)
    del freq, phase, hold
