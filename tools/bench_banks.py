#!/usr/bin/env python3
"""tools/bench_banks.py -- per-bank kernel times at V=65536, B=512 (HIP events on the library stream).

Every line times the raw C-ABI call with all parameters already resident on the device (what a host does in
steady state), so the figure is the kernel plus launch overhead, not Python-side uploads."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
V, B = 65536, 512
if len(sys.argv) > 2 and sys.argv[1] == "--rw":    # knob rw_store: 1 = 8-byte streams, 2 / 3 / 4 = 16-byte pair rows plain / sc1 / nt (0 automatic)
    L.mxg_tune(b"rw_store", int(sys.argv[2]))
    print("# rw_store", sys.argv[2])
for a in sys.argv[1:]:                             # KNOB=VALUE: any other knob of mxg_tune (A/B runs)
    if "=" in a:
        k, val = a.split("=")
        L.mxg_tune(k.encode(), int(val))
        print("#", a)
rng = np.random.default_rng(1)
e0, e1 = L.mxg_event_create(), L.mxg_event_create(); ms = ctypes.c_float()
D = mx.DeviceBuffer.from_numpy


def timed(fn, reps=100):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:      # clock ramp: an idle MI355X needs continuous work to reach its sustained clocks
        for _ in range(20): fn()
        L.mxg_sync()
    L.mxg_event_record(e0, None)
    for _ in range(reps): fn()
    L.mxg_event_record(e1, None); L.mxg_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps * 1e3


def line(name, us, bytes_per_sample, note=""):
    print("%-26s %7.1f us  %6.0f GB/s of %2d B/sample %s" % (name, us, bytes_per_sample * V * B / us / 1e3, bytes_per_sample, note))


x = D(rng.uniform(-1, 1, (B, V)))
out = mx.DeviceBuffer((B, V), zero=False)
v = np.arange(V)

# maxiFilter: coefficients from the host libm once, then the raw render call
cut = 200 + 4 * np.minimum(20 + v * 0.305, 5000.0); res = 1.0 + (v % 16)
coef = np.zeros((3, V))
L.mxg_filter_coeffs_host(0, V, cut.ctypes.data, res.ctypes.data, coef.ctypes.data)
dcut, dres, dcoef, fst = D(cut), D(res), D(coef), mx.DeviceBuffer((5, V))
line("filter lores (hoisted)", timed(lambda: L.mxg_filter_render(0, V, B, x.ptr, dcut.ptr, 0, dres.ptr, 0, dcoef.ptr, fst.ptr, out.ptr, None)), 16)
dlp = D(np.full(V, 0.3)); fst2 = mx.DeviceBuffer((5, V))
line("filter lopass", timed(lambda: L.mxg_filter_render(3, V, B, x.ptr, dlp.ptr, 0, None, 0, None, fst2.ptr, out.ptr, None)), 16)

# maxiSVF / maxiBiquad
svf = mx.maxiSVFBank(V); svf.setCutoff(cut); svf.setResonance(res)
svf.play(x, 1.0, 0.0, 0.0, 0.0, out=out)
line("maxiSVF", timed(lambda: L.mxg_filter2_render(1, V, B, x.ptr, svf.coef.ptr, svf.state.ptr, out.ptr, None)), 16)
bq = mx.maxiBiquadBank(V); bq.set(bq.LOWPASS, cut, np.full(V, 0.7), 0.0)
line("maxiBiquad", timed(lambda: L.mxg_filter2_render(2, V, B, x.ptr, bq.coef.ptr, bq.state.ptr, out.ptr, None)), 16)

# maxiEnv
eb = mx.maxiEnvBank(V); eb.setAttack(10); eb.setDecay(100); eb.setSustain(0.5); eb.setRelease(500)
dpar, dhold = eb._params()
trig = D(((np.arange(B) % 300) < 150).astype(np.int32))
env_call = lambda t: L.mxg_env_render(0, V, B, x.ptr, t.ptr, 0, dpar.ptr, dhold.ptr, eb.dstate.ptr, eb.istate.ptr, out.ptr, None)
line("env adsr (gate 150/150)", timed(lambda: env_call(trig)), 16, "(attack/decay/release every 300 samples: state-machine path)")
hold = D(np.ones(B, np.int32))
for _ in range(4): env_call(hold)
line("env adsr (sustain)", timed(lambda: env_call(hold)), 16, "(gate held: steady-state path)")

# maxiEnvGen, shared gate
eg = mx.maxiEnvGenBank(V); eg.setupADSR(10, 100, 0.5, 500)
gate = D(np.where((np.arange(B) % 300) < 150, 1.0, -1.0))
eg_call = lambda g: L.mxg_envgen_render(V, B, g.ptr, 0, eg.stages.ptr, 4, 0, 0, eg.dstate.ptr, eg.istate.ptr, out.ptr, None)
line("maxiEnvGen ADSR (150/150)", timed(lambda: eg_call(gate)), 8, "(gate toggling every 150 samples: stage machine + ramps)")
held = D(np.ones(B))
low = D(-np.ones(B))
for _ in range(60): eg_call(low)     # finish the 500 ms release: every envelope back in WAITING
for _ in range(20): eg_call(held)    # trigger, attack + decay: every envelope ends in HOLD
line("maxiEnvGen ADSR (holding)", timed(lambda: eg_call(held)), 8, "(gate held, every envelope in its HOLD stage: steady-state path)")

# maxiDelayline
db = mx.maxiDelaylineBank(V, 2048)
dsz, dfb = D(np.full(V, 1024, np.int32)), D(np.full(V, 0.5))
line("delay dl size 1024", timed(lambda: L.mxg_delay_render(0, V, B, x.ptr, dsz.ptr, dfb.ptr, None, db.memory.ptr, 2048, db.phase.ptr, out.ptr, None)), 32)

# maxiSample
sb = mx.maxiSampleBank(V); sb.setSample(rng.uniform(-1, 1, 441000)); sb.setPosition(v / V)
dsp = D(0.5 + (v % 97) / 96.0)
smp_call = lambda mode, a: L.mxg_sample_render(mode, V, B, sb.d_samples, sb.length, 44100, a, 0, None, None, sb.position.ptr, out.ptr, None)
line("sample playAtSpeed", timed(lambda: smp_call(4, dsp.ptr)), 8, "(+ gathers)")
line("sample play", timed(lambda: smp_call(0, None)), 8, "(fractional heads left by playAtSpeed: per-sample gathers)")
sb.position.upload(np.floor(v / V * 441000.0))
line("sample play (int heads)", timed(lambda: smp_call(0, None)), 8, "(heads on integer positions, as after trigger()/load(): 16-B row loads)")
