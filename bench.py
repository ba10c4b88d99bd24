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

    def step():
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

    samples_per_step = V * B * world
    value = samples_per_step * args.steps / elapsed / 1e6
    achieved = ALGO_BYTES_PER_SAMPLE * V * B / (k1_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("k1_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        res = {
            "metric": "Msamples/s (voice-bank render)", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: 65536-voice maxiOsc::%s wavetable bank per GPU, "
                                   "block=512, fp64 out[n][v] stored" % args.waveform,
                       "voices_per_gpu": V, "block": B, "sample_rate": 44100,
                       "parallelism": "voices sharded x%d" % world,
                       "mixdown": "maxiMix::stereo + RCCL reduce of [512x2] per step" if args.mixdown else "off"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "osc_kernel<%s>" % args.waveform, "kernel_ms": round(k1_ms, 5),
                         "algorithmic_bytes_per_launch": round(ALGO_BYTES_PER_SAMPLE * V * B)},
            "realtime_voices_at_44k1": int(value * 1e6 / 44100),
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(freq_h)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
