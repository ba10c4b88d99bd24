"""tools/fft_only.py -- a few launches of the 1024-point FFT batch kernel (for rocprofv3 --pmc passes)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); L.mxg_init(0)
N = 1 << 18
host = np.random.default_rng(0).uniform(-1, 1, N * 1024).astype(np.float32)
sig = mx.DeviceBuffer.from_numpy(host)
f = mx.maxiFFT(); f.setup(1024, 1024, 1024)
mags = mx.DeviceBuffer((N, 512), np.float32, zero=False)
for _ in range(5):
    L.mxg_fft_batch(f.plan, sig.ptr, 1024, N, None, None, mags.ptr, None, None)
L.mxg_sync()
