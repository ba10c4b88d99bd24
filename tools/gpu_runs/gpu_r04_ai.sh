#!/bin/bash
# round 4: maxiOsc::noise with 16-byte streams -- parity, then the store flavours (rotated output, one reused draw block)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ai
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_osc.py -q -x -k "noise" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/noise.txt
import ctypes, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init")
for V in (65536, 131072):
    B = 512
    rnd = [mx.DeviceBuffer.from_numpy(np.random.default_rng(i).integers(0, 2**31 - 1, (B, V)).astype(np.int32)) for i in range(4)]
    outs = [mx.DeviceBuffer((B, V), zero=True) for _ in range(8)]
    hold = mx.DeviceBuffer(V)
    e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
    k = [0]
    def run():
        k[0] += 1
        L.mxg_osc_noise(V, B, rnd[k[0] % 4].ptr, hold.ptr, outs[k[0] % 8].ptr, None)
    res = {}
    for rnd_ in range(6):
        for rw in (1, 2, 3, 4, 0):
            L.mxg_tune(b"rw_store", rw)
            for _ in range(3): run()
            L.mxg_event_record(e0, None)
            for _ in range(20): run()
            L.mxg_event_record(e1, None); L.mxg_event_sync(e1); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            if rnd_: res.setdefault(rw, []).append(ms.value / 20 * 1e3)
    L.mxg_tune(b"rw_store", 0)
    for rw, ts in res.items():
        med = float(np.median(ts))
        print("noise %d voices rw_store %d: %.1f us  %.3f of 8 TB/s on 12 B/sample" % (V, rw, med, 12.0 * V * B / med / 1e3 / 8000))
PY
