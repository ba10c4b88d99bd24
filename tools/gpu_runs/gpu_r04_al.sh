#!/bin/bash
# round 4: maxiFilter pair-row kernel, the chunk's stores back to back against a store per pair of samples (A/B builds, rotated arenas)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04al
mkdir -p $O
cd $R
for round in 1 2; do for lib in libmaxigpu.so ab_fltold.so; do
MXG_LIB=$R/maximilian_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu | tee -a $O/ab.txt
import ctypes, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); chk = mx._lib.check; chk(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
B = 512
ARENA = 4 << 30
a_in, a_out = L.mxg_malloc(ARENA), L.mxg_malloc(ARENA)
chk(L.mxg_memset(a_in, 0, ARENA, None), "m"); chk(L.mxg_memset(a_out, 0, ARENA, None), "m"); chk(L.mxg_sync(), "s")
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
D = mx.DeviceBuffer.from_numpy
out = []
for V in (65536, 131072):
    nb = V * B * 8; regions = ARENA // nb
    v = np.arange(V)
    cut = 200 + 4 * np.minimum(20 + v * 0.305, 5000.0); res_ = 1.0 + (v % 16)
    coef = np.zeros((3, V)); L.mxg_filter_coeffs_host(0, V, cut.ctypes.data, res_.ctypes.data, coef.ctypes.data)
    dcut, dres, dcoef, fst, fst2 = D(cut), D(res_), D(coef), mx.DeviceBuffer((5, V)), mx.DeviceBuffer((5, V))
    dlp = D(np.full(V, 0.3))
    for name, call in (("lores", lambda i, o: L.mxg_filter_render(0, V, B, i, dcut.ptr, 0, dres.ptr, 0, dcoef.ptr, fst.ptr, o, None)),
                       ("lopass", lambda i, o: L.mxg_filter_render(3, V, B, i, dlp.ptr, 0, None, 0, None, fst2.ptr, o, None))):
        k = [0]
        def run():
            k[0] += 1
            r = (k[0] % regions) * nb
            chk(call(a_in + r, a_out + r), name)
        ts = []
        for rnd in range(6):
            for _ in range(3): run()
            L.mxg_event_record(e0, None)
            for _ in range(12): run()
            L.mxg_event_record(e1, None); L.mxg_event_sync(e1); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            if rnd: ts.append(ms.value / 12 * 1e3)
        out.append("%s %d: %.1f us" % (name, V, np.median(ts)))
print(os.path.basename(os.environ.get("MXG_LIB", "")), " | ".join(out))
PY
done; done
