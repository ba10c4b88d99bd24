#!/usr/bin/env python3
"""bench.py -- headline measurement: Msamples/s of the voice-bank render (BASELINE.json).

Workload (BASELINE.json configs[1]): a 65 536-voice maxiOsc::sinebuf
wavetable bank per GPU, block = 512 samples, freq[v] = 20 + v*0.30517578125 Hz, state carried
from block to block.  One "step" = one block of the whole bank through the hot path
(libmaxigpu.so, kernel K1 `osc_kernel<sinebuf>` + the stereo mixdown partials), inputs and
state resident in HBM.  With --gpus N (one process per GPU, launched by torch.distributed.run)
every rank renders its own 65 536-voice shard (weak scaling); the only exchange is the
[512 x 2] fp64 mixdown, reduced to rank 0 over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/...,
plus "roofline" (dominant kernel vs the 8 TB/s HBM peak, algorithmic bytes 8.047 B/sample,
duration from HIP events on the launch stream) and "cpu_baseline" (the reference's own CPU
loop -- oracle/_ref when present, else the plain-C port -- timed on this host on a bounded
sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOICES_PER_GPU = 65536
BLOCK = 512
ALGO_BYTES_PER_SAMPLE = 8.0 + 24.0 / BLOCK  # 8 B store + (freq, phase rd, phase wr)/block = 8.047
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(freq):
    """The reference CPU loop on this host: bounded sample of the same workload."""
    from oracle import pyoracle
    cores = usable_cores()
    # bounded sample: ~3.2 G samples = roughly 20 core-seconds of the reference loop
    if pyoracle.have_reference():
        o, kind, threads, blocks = pyoracle.reference(), "reference", cores, 96
    else:
        o, kind, threads, blocks = pyoracle.port(), "port", 1, 48
    o.settings(44100, 2, 1024)
    nsamp = BLOCK * blocks
    secs = o.time_osc(8, freq, nsamp, threads=threads)
    return {
        "value": round(freq.size * nsamp / secs / 1e6, 2), "unit": "Msamples/s", "cores": threads,
        "kind": kind,
        "sample": "%d voices x %d samples (%d blocks of %d) maxiOsc::sinebuf, voice-inner loop, "
                  "voices sharded over %d thread(s); %.2f s wall" % (freq.size, nsamp, blocks, BLOCK,
                                                                     threads, secs),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--waveform", default="sinebuf")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config5"],
                    help="BASELINE.json config to run; the default (config2) is the one the headline metric is quoted on")
    ap.add_argument("--mixdown", nargs="?", const="fused", default=None, choices=["fused", "separate"],
                    help="also produce the stereo mixdown each step and reduce it to rank 0 over RCCL")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import maximilian_amd as mx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    L = mx.lib()
    mx._lib.check(L.mxg_init(local), "mxg_init")
    mx.maxiSettings.setup(44100, 2, 1024)
    dev = torch.device("cuda", local)
    # Launch on a non-default torch stream and time with events recorded on the SAME stream
    # (the C-ABI treats a NULL stream as "the library's own stream").
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    from maximilian_amd.dist import MixReducer, bank_parameters, shard_range

    V, B = VOICES_PER_GPU, BLOCK
    wf = mx.OSC_WAVEFORMS[args.waveform]
    # this rank's voice shard of the global bank: voices [rank*V, (rank+1)*V)
    lo, hi = shard_range(rank, world, V)
    freq_h, pan_h = bank_parameters(lo, hi, V * world)
    freq = torch.from_numpy(freq_h).to(dev)
    phase = torch.zeros(V, dtype=torch.float64, device=dev)
    hold = torch.zeros(V, dtype=torch.float64, device=dev)
    out = torch.empty((B, V), dtype=torch.float64, device=dev)
    pan = torch.from_numpy(pan_h).to(dev)
    reducer = MixReducer(dist if world > 1 else None,
                         lambda: torch.zeros((B, 2), dtype=torch.float64, device=dev))

    def render():
        mx._lib.check(L.mxg_osc_render(wf, V, B, freq.data_ptr(), 0, None, None, phase.data_ptr(),
                                       hold.data_ptr(), out.data_ptr(), stream), "mxg_osc_render")

    # ---- the other BASELINE configs, same JSON shape, same unit (a "sample" is one voice output,
    # one FFT input sample, one grain-sample).  They are parity-test cases; the headline stays config2. ----
    alt = None
    if args.workload == "config3":
        K = 128
        vb = mx.maxiVoiceBank(V, stream=stream)
        vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
        f3 = np.minimum(freq_h, 5000.0)
        vb.render(0, f3, 200 + 4 * f3, 1.0 + (np.arange(lo, hi) % 16), np.zeros(1, np.int32), 1, out=mx.DeviceBuffer((1, V)))
        vf, vcu, vrs, vcoef, _ = vb._keep
        vpar, vhold = vb.env._params()
        gate = mx.DeviceBuffer.from_numpy(((np.arange(K * B) % 44100) < 22050).astype(np.int32))
        blk = [0]

        def step3():
            mx._lib.check(L.mxg_voice_render(0, V, B, vf.ptr, vcu.ptr, vrs.ptr, vcoef.ptr, gate.ptr + 4 * (blk[0] % K) * B, 0,
                                             vpar.ptr, vhold.ptr, vb.osc_state.ptr, vb.flt_state.ptr, vb.env.dstate.ptr,
                                             vb.env.istate.ptr, out.data_ptr(), stream), "mxg_voice_render")
            blk[0] += 1
        alt = dict(step=step3, samples=V * B, bytes=(8.0 + 176.0 / B) * V * B, kernel="voice_kernel<0> (saw->lores->adsr, hoisted)",
                   workload="configs[2]: fused subtractive voice maxiOsc::saw -> maxiFilter::lores -> maxiEnv::adsr, %d voices "
                            "per GPU, block=512, gate(n) = (n mod 44100) < 22050 cycled over 128 blocks" % V, dtype="f64")
    elif args.workload == "config4":
        NF = 1 << 20
        g = torch.Generator(device=dev); g.manual_seed(0x4D415849 + rank)
        sig = torch.empty(NF * 1024, dtype=torch.float32, device=dev)
        for c0 in range(0, NF, 1 << 16):
            n = torch.arange(c0 * 1024, (c0 + (1 << 16)) * 1024, dtype=torch.float64, device=dev)
            k = torch.div(n, 1024, rounding_mode="floor")
            sig[c0 * 1024:(c0 + (1 << 16)) * 1024] = (0.4 * torch.sin(2 * np.pi * 220 * n / 44100) + 0.3 * torch.sin(
                2 * np.pi * (440 + 0.01 * k) * n / 44100) + 0.1 * (2 * torch.rand(n.numel(), dtype=torch.float64, device=dev,
                                                                                  generator=g) - 1)).to(torch.float32)
            del n, k
        mags = torch.empty((NF, 512), dtype=torch.float32, device=dev)
        mfcc = torch.empty((NF, 13), dtype=torch.float64, device=dev)
        fplan = mx.maxiFFT(); fplan.setup(1024, 1024, 1024)
        mplan = mx.maxiMFCC(); mplan.setup(512, 42, 13, 20.0, 20000.0)

        def step4():
            mx._lib.check(L.mxg_fft_batch(fplan.plan, sig.data_ptr(), 1024, NF, None, None, mags.data_ptr(), None, stream), "fft")
            mx._lib.check(L.mxg_mfcc_batch(mplan.plan, mags.data_ptr(), 512, NF, None, None, mfcc.data_ptr(), 0, stream), "mfcc")
        alt = dict(step=step4, samples=NF * 1024, bytes=4200.0 * NF, kernel="fft1024_kernel + mfcc_stream_tiled_kernel",
                   workload="configs[3]: maxiFFT(1024,1024,1024) + maxiMFCC(512,42,13,20,20000) over %d frames per GPU per step" % NF,
                   dtype="f32 (FFT) / f64 (MFCC)")
    elif args.workload == "config5":
        S, T, Ls = 2048, 70560, 4410000
        rng5 = np.random.default_rng(0x4D415849)
        n5 = np.arange(Ls)
        smp = 0.5 * np.sin(2 * np.pi * 110 * n5 / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n5 / 44100) + 0.05 * rng5.uniform(-1, 1, Ls)
        sb5 = mx.maxiSampleBank(1, stream=stream); sb5.setSample(smp)
        gb = mx.maxiTimeStretchBank(S, sb5, "hann", stream=stream)
        s_glob = np.arange(rank * S, (rank + 1) * S)
        gb.setPosition(s_glob / float(S * world))
        sp5 = mx.DeviceBuffer.from_numpy(0.25 + 1.5 * (s_glob % 97) / 96)
        out5 = mx.DeviceBuffer((T, S), zero=False)
        plan5 = gb._plan(0.05)

        def step5():
            mx._lib.check(L.mxg_granular_render(plan5, 0, S, T, sb5.d_samples, Ls, 4, sp5.ptr, None, None, None, 0,
                                                gb.state.ptr, gb.grains.ptr, out5.ptr, stream), "mxg_granular_render")
        alt = dict(step=step5, samples=S * T * 4, bytes=(8.0 * 4 + 8.0) * S * T, kernel="granular_sched_kernel + granular_unit_kernel",
                   workload="configs[4]: %d maxiTimeStretch<hann> streams per GPU x %d samples per step, grainLength 0.05, overlaps 4 "
                            "(4 live grains per stream-sample counted)" % (S, T), dtype="f64")

    def step():
        if alt is not None:
            alt["step"]()
            return
        if not args.mixdown:
            render()
            return
        # render + maxiMix::stereo mixdown of this rank's voices fused in one pass (K1m), then the
        # single exchange of the path: an asynchronous RCCL reduce of the [512 x 2] block to rank 0
        mixbuf = reducer.next_buffer()
        if args.mixdown == "fused":
            mx._lib.check(L.mxg_osc_render_mix(wf, V, B, freq.data_ptr(), None, None, phase.data_ptr(),
                                               hold.data_ptr(), out.data_ptr(), pan.data_ptr(),
                                               mixbuf.data_ptr(), stream), "mxg_osc_render_mix")
        else:  # "separate": K1 then K3 re-reading the block
            render()
            mx._lib.check(L.mxg_mix_stereo(V, B, out.data_ptr(), pan.data_ptr(), mixbuf.data_ptr(), stream),
                          "mxg_mix_stereo")
        reducer.submit()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed clock ramp: an idle MI355X sits in a low-power state (sclk ~500 MHz) and takes
    # milliseconds of continuous work to reach its sustained clocks; a short --steps run would
    # otherwise time the ramp instead of the kernel.  Then the W requested warmup steps.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # Average launch duration of the dominant kernel (K1): two HIP events on the launch stream
    # bracketing the K back-to-back launches of the timed region (per-launch event pairs would
    # put ~10 us of host/marker gaps between the kernels; measured in profiles/).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    ev1.record()
    reducer.drain()
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    k1_ms = ev0.elapsed_time(ev1) / args.steps
    if world > 1:
        t = torch.tensor([elapsed, k1_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, k1_ms = float(t[0]), float(t[1])

    samples_per_step = (alt["samples"] if alt else V * B) * world
    value = samples_per_step * args.steps / elapsed / 1e6
    achieved = (alt["bytes"] if alt else ALGO_BYTES_PER_SAMPLE * V * B) / (k1_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and alt is None:
        try:
            traffic = json.load(open(pmc)).get("k1_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        res = {
            "metric": "Msamples/s (voice-bank render)", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": alt["dtype"] if alt else "f64", "data": "synthetic",
            "config": {"workload": alt["workload"] if alt else "configs[1]: 65536-voice maxiOsc::%s wavetable bank per GPU, "
                                   "block=512, fp64 out[n][v] stored" % args.waveform,
                       "voices_per_gpu": V, "block": B, "sample_rate": 44100,
                       "parallelism": "voices sharded x%d" % world,
                       "mixdown": "maxiMix::stereo + RCCL reduce of [512x2] per step" if args.mixdown else "off"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": alt["kernel"] if alt else "osc_kernel<%s>" % args.waveform, "kernel_ms": round(k1_ms, 5),
                         "algorithmic_bytes_per_launch": round(alt["bytes"] if alt else ALGO_BYTES_PER_SAMPLE * V * B)},
            "realtime_voices_at_44k1": int(value * 1e6 / 44100),
        }
        if world == 1 and not args.no_cpu_baseline and alt is None:
            res["cpu_baseline"] = cpu_baseline(freq_h)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
