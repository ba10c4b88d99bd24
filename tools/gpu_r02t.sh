#!/bin/bash
# round 2, call q: window staging for the interpolating maxiSample players
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_gpu_extra.py tests/test_gpu_edges.py tests/test_gpu_host.py -m gpu -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log


for sp in 1 4 8; do echo "== smp_split=$sp" >> $O/split.txt; MXG_SMP_SPLIT=$sp timeout 300 python - >> $O/split.txt 2>&1 <<PY
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(-1), "init"); mx.maxiSettings.setup(44100, 2, 1024)
L.mxg_tune(b"smp_split", int(os.environ["MXG_SMP_SPLIT"]))
V, B = 65536, 512
v = np.arange(V)
sb = mx.maxiSampleBank(V); sb.setSample(np.random.default_rng(1).uniform(-1, 1, 441000)); sb.setPosition(v / V)
dsp = mx.DeviceBuffer.from_numpy(0.5 + (v % 97) / 96.0)
out = mx.DeviceBuffer((B, V))
for mode in (4,):
    end = mx.DeviceBuffer.from_numpy(np.ones(V))
    call = lambda: L.mxg_sample_render(mode, V, B, sb.d_samples, sb.length, 44100, dsp.ptr, 0, None, end.ptr, sb.position.ptr, out.ptr, None)
    for _ in range(20): call()
    L.mxg_stream_sync(None); t = time.perf_counter()
    for _ in range(200): call()
    L.mxg_stream_sync(None); print("mode %d: %.1f us" % (mode, (time.perf_counter() - t) / 200 * 1e6))
PY
done; cat $O/split.txt
timeout 600 python tools/bench_banks.py 2>/dev/null | grep sample
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_host.py tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -2
