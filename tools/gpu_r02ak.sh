#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ak
mkdir -p $O
cd $R
python - > $O/sin_split.txt 2>&1 <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(-1), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
freq = mx.DeviceBuffer.from_numpy(20.0 + np.arange(V) * 0.30517578125)
bank = mx.maxiOscBank(V)
out = mx.DeviceBuffer((B, V), zero=False)
for wf in ("sinewave", "coswave"):
    for sp in (1, 2, 3, 4, 6, 8):
        L.mxg_tune(b"osc_split", sp)
        call = lambda: bank.render(wf, freq, B, out=out)
        for _ in range(30): call()
        L.mxg_stream_sync(None); t = time.perf_counter()
        for _ in range(300): call()
        L.mxg_stream_sync(None); print("%s split %d: %.1f us" % (wf, sp, (time.perf_counter() - t) / 300 * 1e6), flush=True)
PY
grep -v amdgpu $O/sin_split.txt
