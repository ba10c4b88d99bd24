#!/usr/bin/env python3
"""bench.py -- headline measurement: Msamples/s of the voice-bank render (BASELINE.json).

Default workload = BASELINE.json configs[1]: a 65 536-voice maxiOsc::sinebuf wavetable bank per
GPU, block = 512 samples, freq[v] = 20 + v*0.30517578125 Hz, state carried from block to block.  One "step" = one
block of the whole bank through the hot path (libmaxigpu.so): kernel K1 renders the block and stores every voice's fp64
sample (out[n][v], the HBM-bound stream).  With more than one GPU (one process per GPU, weak scaling: 65 536 voices
each) the step ALSO carries the path's one exchange: kernel K1m renders, stores and, in the same pass, forms the
maxiMix::stereo mixdown of the rank's voices; the [512 x 2] mixes of 16 consecutive blocks are staged in the C-ABI's mix
queue and summed onto rank 0 with ONE ncclReduce per 16 blocks on the queue's own stream (RCCL over xGMI), overlapped
with the next blocks' render.  So an N > 1 step does more than an N = 1 step (the fused mixdown costs K1m 48 us against
K1's 42 us on one MI355X: `--mixdown fused` at N = 1 shows it); `--mixdown off|fused|separate` forces either form.

`--workload config3|config4|config5` runs the other BASELINE configs through the same harness and JSON shape (same
unit; SURVEY 8d: a "sample" is one voice output, one FFT input sample, one grain-sample).

Prints ONE JSON line on rank 0: metric/value/unit/..., plus
  "roofline"      the dominant kernel against the 8 TB/s HBM peak (or the fp64 MFMA peak for --mfcc-method mfma):
                  algorithmic bytes per launch / average launch duration, the duration measured with HIP events the
                  library records around that kernel on its launch stream (mxg_prof_*) inside the timed region;
                  "traffic" = HBM bytes per launch from the rocprofv3 PMC passes kept in profiles/pmc_traffic.json;
  "kernels"       average duration and launches per step of every kernel the step launches;
  "cpu_baseline"  the reference's own per-sample CPU loop (oracle/_ref, the compiled reference; else the plain-C
                  port) timed on this host's cores on a bounded sample of the same workload (rank 0, N = 1 only),
                  multi-threaded ("cores") and single-threaded ("single_thread").

Round 6: several of the kernels timed here run on the library's paced store schedule (csrc/mxg_pace.h) -- a controller, or for the
headline's size a trial, kept in device scratch per stream; they settle within the first dozens of launches of a stream, which the
warm-up loops cover.  MXG_PRINT_PACE=1 prints their words to stderr after the timed loop (diagnostics); `--tune voice_pace=1`,
`osc_pace=1`, `smp_pace=1` switch the schedule off.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOICES_PER_GPU = 65536
BLOCK = 512
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6  # dense fp64 matrix peak


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


def _oracle():
    from oracle import pyoracle
    if pyoracle.have_reference():
        return pyoracle.reference(), "reference"
    return pyoracle.port(), "port"


def _baseline(run, units_of, sizes, unit_name, what, target_s=6.0):
    """Time `run(size, threads)` -> seconds on a bounded sample.  A small probe sets the sample size so that the
    multi-threaded run takes about target_s (and the single-threaded one about half that)."""
    o, kind = _oracle()
    o.settings(44100, 2, 1024)
    cores = usable_cores() if kind == "reference" else 1
    probe, cap = sizes
    t = max(run(o, probe, 1), 1e-4)
    rate1 = units_of(probe) / t
    n1 = int(min(cap, max(probe, probe * (0.5 * target_s) / t)))
    s1 = run(o, n1, 1)
    res = {"unit": "Msamples/s", "kind": kind}
    if cores > 1:
        nm = int(min(cap, max(probe, n1 * 2 * cores * 0.7)))
        sm = run(o, nm, cores)
        res.update(value=round(units_of(nm) / sm / 1e6, 2), cores=cores,
                   sample="%s, %s = %d, units sharded over %d threads; %.2f s wall" % (what, unit_name, nm, cores, sm))
        res["single_thread"] = {"value": round(units_of(n1) / s1 / 1e6, 2), "cores": 1,
                                "sample": "%s = %d; %.2f s wall" % (unit_name, n1, s1)}
    else:
        res.update(value=round(units_of(n1) / s1 / 1e6, 2), cores=1,
                   sample="%s, %s = %d, one thread; %.2f s wall" % (what, unit_name, n1, s1))
    return res


def share_estimate(kernels, dom, step_ms):
    """(kernel_ms, step-bound) of the dominant kernel.  A step that launches ONLY this kernel: the GPU-side step time / launches per step
    (no per-kernel markers in the timed region; an upper bound on the launch duration: it carries the launch gaps).  A step of several
    kernels (some of them on other streams, overlapping): the HIP events around this kernel's own launches (they carry the marker
    pair, ~2-4 us: conservative)."""
    if dom not in kernels:
        return step_ms, step_ms
    lps = max(kernels[dom]["launches_per_step"], 1e-9)
    bound = step_ms / max(lps, 1.0)
    only = all(k == dom or v["launches_per_step"] * v["ms"] < 0.02 * step_ms for k, v in kernels.items())
    return (bound if only else min(kernels[dom]["ms"], bound)), bound


ESTIMATOR = ("single-kernel step: GPU-side step time (two HIP events around the timed steps, no markers inside) / launches per step = an "
             "upper bound on the average launch duration (launch gaps included); step of several kernels: HIP events around this "
             "kernel's own launches (marker pair included: conservative), never more than the step")


def _sig(x, n=5):
    return float("%.*g" % (n, x)) if isinstance(x, float) else x


def compact_line(res, tag):
    """The default (short) form of the JSON line: the same numbers, texts said once.  The driver keeps an 8 KB tail of the line, so every
    `configs.*.{ms_per_step, value, roofline.{frac, step_frac, kernel_ms, traffic}, cpu_baseline.value}` has to fit in well under that;
    `--verbose` prints the long form (workload texts, estimator notes, per-kernel tables)."""
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data")}
    cfg = dict(res["config"])
    if tag:
        cfg["workload"] = tag
    out["config"] = cfg
    r = res["roofline"]
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "launches_per_step", "algorithmic_bytes_per_launch",
            "flops_per_launch", "frac_wall", "frac_single_buffer", "frac_of_measured_write_ceiling", "frac_of_measured_read_ceiling")
    ro = {k: _sig(r[k]) for k in keep if r.get(k) is not None or k == "traffic"}
    for c in ("write_ceiling", "read_ceiling"):
        if c in r:
            ro[c + "_GB/s"] = r[c]["GB/s"]
    if "l2_gather_model" in r:
        ro["l2_gather_GB/s"] = r["l2_gather_model"]["achieved_GB/s"]
    out["roofline"] = ro
    cb = res.get("cpu_baseline")
    if cb:
        o = {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb}
        o["sample"] = cb.get("sample", "")[:150]
        if "single_thread" in cb:
            o["single_thread"] = cb["single_thread"]["value"]
        out["cpu_baseline"] = o
    elif "cpu_baseline" in res:
        out["cpu_baseline"] = None
    for k in ("exchange", "rccl_ranks", "step_ms_gpu", "realtime_voices_at_44k1", "per_gpu_efficiency", "step_ms_without_reduce",
              "value_like_for_like_n1", "share_gpu_test"):
        if res.get(k) is not None:
            out[k] = res[k]
    if "north_star_bank" in res:
        out["north_star_bank"] = {k: res["north_star_bank"][k] for k in ("voices", "ms_per_step", "value", "frac_hbm_peak",
                                                                        "frac_of_measured_write_ceiling", "realtime_factor_at_44k1")}
    out["notes"] = {"kernel_ms": "single-kernel step: GPU-side step time / launches (upper bound, gaps included); multi-kernel step: HIP "
                                 "events around the kernel's own launches, never more than the step",
                    "traffic": "HBM bytes per launch, rocprofv3 --pmc passes of the same commands (profiles/pmc_traffic.json)",
                    "cpu_baseline": "the compiled reference on this host's cores; per config a ~1 s sample",
                    "long_form": "bench.py --verbose"}
    cs = {}
    for name, c in res.get("configs", {}).items():
        if "error" in c:
            cs[name] = c
            continue
        e = {"workload": c.get("tag") or c["workload"][:110], "ms_per_step": c["ms_per_step"], "value": c["value"]}
        for k in ("n_gpus", "scaling", "steps", "step_vs_headline", "step_vs_config3"):
            if k in c:
                e[k] = c[k]
        if "roofline" in c:
            q = c["roofline"]
            e["roofline"] = {k: _sig(q[k]) for k in ("bound", "kernel", "kernel_ms", "achieved", "unit", "frac", "step_frac", "kernel_frac", "traffic",
                                                    "l2_gather_GB/s") if k in q}
            if "matrix_pipe" in q:
                e["roofline"]["matrix_pipe"] = {"achieved_TFLOP/s": q["matrix_pipe"]["achieved_TFLOP/s"], "utilisation": q["matrix_pipe"]["utilisation"]}
            if "fp64_valu" in q:
                e["roofline"]["fp64_valu"] = {"achieved_TFLOP/s": q["fp64_valu"]["achieved_TFLOP/s"], "frac": q["fp64_valu"]["frac"]}
        if c.get("cpu_baseline"):
            b = c["cpu_baseline"]
            e["cpu_baseline"] = {k: b[k] for k in ("value", "cores", "kind", "error") if k in b}
        cs[name] = e
    if cs:
        out["configs"] = cs
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--waveform", default="sinebuf")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config5", "tables", "sample_bank"],
                    help="BASELINE.json config to run; the default (config2) is the one the headline metric is quoted on")
    ap.add_argument("--mixdown", default=None, choices=["fused", "separate", "off"],
                    help="stereo mixdown + cross-GPU reduce in the step (default: on whenever --gpus > 1 -- fused into the "
                         "render for config2, K3 for config3 -- and always for config5; off on one GPU)")
    ap.add_argument("--mix-depth", type=int, default=32,
                    help="blocks per ncclReduce (M of SURVEY 8e).  32 since round 6: every batch costs the render stream a fixed ~50-85 us (its "
                         "event record drains the back-to-back kernels, the fold shares the machine) -- the N > 1 step on one GPU, M = 8 / 16 / 32 / "
                         "64: 51.7 / 49.7 / 44.4 / 44.0 us against K1's 40.6 (profiles/r06_mix_depth.md)")
    ap.add_argument("--voice-mode", type=int, default=0, choices=[0, 1], help="config3: 0 = hoisted coefficients, 1 = 14.monosynth order")
    ap.add_argument("--mfcc-method", default="sparse", choices=["sparse", "walk", "mfma", "mfma-gemm"],
                    help="config4: the fused kernel with the exact sparse mel walk (default), the fused kernel with the mel contraction "
                         "and the DCT on the matrix pipe (mfma: knob fused_mel 3, banded v_mfma_f64_4x4x4), or the round-2 two-kernel "
                         "route (mfma-gemm: FFT kernel + dense fp64 MFMA GEMM over the stored magnitudes)")
    ap.add_argument("--mix-only", action="store_true", help="config2 fused: do not store the per-voice block (VALU/LDS time of K1m)")
    ap.add_argument("--out-buffers", type=int, default=0,
                    help="config2/3: rotate the per-voice output over this many block buffers.  0 (default) = as many as it "
                         "takes to touch >= 2 GiB before a line is rewritten (8 at 65 536 voices: 8x the 256 MB Infinity "
                         "Cache, so every block's stores reach HBM); 1 = one block buffer reused (a block renderer's own "
                         "steady state, partly absorbed by the Infinity Cache: reported as roofline.frac_single_buffer)")
    ap.add_argument("--voices", type=int, default=0,
                    help="config2/3: voices per GPU (default 65 536 = BASELINE configs[1]); the default config2 run ALSO measures "
                         "a 131 072-voice bank (north_star: >= 10^5 voices) and reports it as `north_star_bank`")
    ap.add_argument("--no-extras", action="store_true",
                    help="config2, one GPU: skip the extra measurements of the default run (single-buffer pass, write-ceiling fills, "
                         "the 131 072-voice bank) -- for profiler passes, so that only the headline launches are traced")
    ap.add_argument("--no-configs", action="store_true",
                    help="config2, one GPU: skip the `configs` object of the default run (configs 3, 4, 4-mfma, 5 and the fused-mixdown "
                         "step measured the short way in the same process)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: ranks of a --gpus N run share the visible GPUs (rank r uses device r mod count), the "
                         "id/barrier traffic goes over gloo and the mix queue reduces locally (no RCCL: two ranks cannot share "
                         "one device in a communicator); exercises the launch + N-rank harness on a 1-GPU box")
    ap.add_argument("--tune", action="append", default=[], help="KEY=VALUE passed to mxg_tune (A/B experiments)")
    ap.add_argument("--mfma-fullk", action="store_true",
                    help="config4 mfma: contract over all 512 bins instead of the 256 that carry mel weight")
    ap.add_argument("--verbose", action="store_true",
                    help="print the long form of the line (full workload texts, estimator notes, per-kernel tables in every `configs` entry); "
                         "the default line is the compact form (< 7000 characters: the driver keeps an 8 KB tail)")
    ap.add_argument("--kernel-events", default="pass", choices=["inline", "pass", "off"],
                    help="per-kernel HIP events inside the timed region (inline), in a separate pass, or not at all")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher.  One process per GPU under torch.distributed.run
        # (the same command line the driver uses for N > 1); rank 0 prints the JSON line.
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
        env.setdefault("OMP_NUM_THREADS", "1")
        os.execvpe(cmd[0], cmd, env)
    # (under torch.distributed.run the environment is the launcher's: make sure of dmabuf IPC before HIP / RCCL are loaded)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    defaults = {"config2": (2000, 100), "config3": (1280, 128), "config4": (20, 3), "config5": (20, 3), "tables": (200, 20), "sample_bank": (200, 20)}
    if args.steps is None:
        args.steps = defaults[args.workload][0]
    if args.warmup is None:
        args.warmup = defaults[args.workload][1]

    import torch
    import torch.distributed as dist
    import maximilian_amd as mx
    from maximilian_amd.dist import (MixdownStep, RcclMixQueue, TorchMixQueue, bank_parameters, create_comm, shard_range,
                                     stream_parameters)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists for the product path)")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` by itself, or under "
                         "torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if local >= ndev:
        if not args.share_gpu:
            raise SystemExit("bench.py: rank %d needs GPU %d but this node shows %d GPU(s): --gpus %d needs %d GPUs"
                             % (rank, local, ndev, args.gpus, args.gpus))
        local = local % ndev
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    L = mx.lib()
    CAL = mx.calib()  # measurement probes: libmaxicalib.so (include/maxicalib.h)
    chk = mx._lib.check
    chk(L.mxg_init(local), "mxg_init")
    mx.maxiSettings.setup(44100, 2, 1024)
    for kv in args.tune:
        k, v = kv.split("=")
        if k.startswith("tables_"):  # (bench switches, not library knobs)
            continue
        chk(L.mxg_tune(k.encode(), int(v)), "mxg_tune " + kv)
    dev = torch.device("cuda", local)
    # Launch on a non-default torch stream (the C-ABI treats a NULL stream as "the library's own stream").
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    # the product's communicator: RCCL through the C-ABI (torch.distributed only carries the 128-byte id)
    comm, exchange = None, None
    force_torch = os.environ.get("MXG_BENCH_FORCE_TORCH_EXCHANGE") == "1"  # (tests: exercise the fallback)
    if world > 1 and force_torch:
        exchange = "FALLBACK: torch.distributed.reduce (forced by MXG_BENCH_FORCE_TORCH_EXCHANGE)"
    elif world > 1 and not args.share_gpu:
        try:
            comm = create_comm(dist, rank, world, dev)
            exchange = "ncclReduce on the library's own RCCL communicator (mxg_comm / mxg_mixq)"
        except Exception as e:  # (never seen; a scaling run that dies here would measure nothing at all)
            exchange = "FALLBACK: torch.distributed.reduce (mxg_comm_create failed on rank %d: %s)" % (rank, e)
    if world > 1 and not args.share_gpu and not force_torch:
        # every rank takes the same path, or the reduces would never meet
        ok = torch.tensor([1 if comm else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm:
                L.mxg_comm_destroy(comm)
                exchange = "FALLBACK: torch.distributed.reduce (mxg_comm_create failed on another rank)"
            comm = None

    def make_queue(block_doubles, depth, groups=1):
        """The mix queue of the step: the library's (RCCL communicator inside libmaxigpu.so); the torch fallback only if that could not be made."""
        if world > 1 and comm is None and (force_torch or not args.share_gpu):
            return TorchMixQueue(dist, block_doubles, depth, 0, stream, dev, groups=groups)
        return RcclMixQueue(comm, block_doubles, depth, 0, stream, groups=groups)

    V, B = (args.voices or VOICES_PER_GPU), BLOCK
    nbuf = args.out_buffers if args.out_buffers > 0 else max(1, -(-(1 << 31) // (V * B * 8)))
    wf = mx.OSC_WAVEFORMS[args.waveform]
    lo, hi = shard_range(rank, world, V)  # this rank's voice shard of the global bank
    freq_h, pan_h = bank_parameters(lo, hi, V * world)
    # config 2/3: the cross-GPU mixdown (and its RCCL reduce) is part of the step whenever there is more than one GPU;
    # a single GPU renders the bank without it unless asked (--mixdown fused|separate).  config 5 is DEFINED with the
    # stereo mixdown (BASELINE configs[4]), so it always mixes.
    mixdown = args.mixdown or {"config2": "fused" if world > 1 else "off", "config3": "fused" if world > 1 else "off",
                               "config4": "off", "config5": "fused", "tables": "off", "sample_bank": "off"}[args.workload]

    class OscBank:
        """One rank's maxiOsc bank of `Vb` voices with its block buffers; step() renders one block (K1, or K1m / K1 + K3 with
        the mixdown into `q`'s current slot)."""

        def __init__(self, Vb, buffers, mix, q, lo_v=0, total=None):
            fh, ph = bank_parameters(lo_v, lo_v + Vb, total or Vb)
            self.V, self.freq_h, self.mix, self.q = Vb, fh, mix, q
            self.freq = torch.from_numpy(fh).to(dev)
            self.pan = torch.from_numpy(ph).to(dev)
            self.phase = torch.zeros(Vb, dtype=torch.float64, device=dev)
            self.hold = torch.zeros(Vb, dtype=torch.float64, device=dev)
            self.outs = [torch.empty((B, Vb), dtype=torch.float64, device=dev) for _ in range(buffers)]
            self.i = 0
            self.mstep = MixdownStep(self.render_mix, q) if mix != "off" else None
            self.algo = (8.0 + 24.0 / B) * Vb * B  # 8 B store + (freq, phase rd, phase wr)/block = 8.047 B/sample
            if mix == "fused":
                self.algo += 16.0 * Vb + 16.0 * B * (Vb / 256.0)  # + gains read, per-workgroup mix partials written
            self.dominant = "osc_mix_kernel" if mix == "fused" else "osc_kernel"

        def out_ptr(self):
            return self.outs[self.i % len(self.outs)].data_ptr()

        def render_mix(self, slot):
            if self.mix == "fused":  # K1m: render + store + per-workgroup mix rows in one pass; the (grouped) queue adds the rows
                chk(L.mxg_osc_render_mix_rows(wf, self.V, B, self.freq.data_ptr(), None, None, self.phase.data_ptr(),
                                              self.hold.data_ptr(), None if args.mix_only else self.out_ptr(), self.pan.data_ptr(),
                                              slot, stream), "mxg_osc_render_mix_rows")
            else:                    # K1 then K3 re-reading the block
                chk(L.mxg_osc_render(wf, self.V, B, self.freq.data_ptr(), 0, None, None, self.phase.data_ptr(),
                                     self.hold.data_ptr(), self.out_ptr(), stream), "mxg_osc_render")
                chk(L.mxg_mix_stereo(self.V, B, self.out_ptr(), self.pan.data_ptr(), slot, stream), "mxg_mix_stereo")

        def step(self):
            if self.mstep is None:
                chk(L.mxg_osc_render(wf, self.V, B, self.freq.data_ptr(), 0, None, None, self.phase.data_ptr(),
                                     self.hold.data_ptr(), self.out_ptr(), stream), "mxg_osc_render")
            else:
                self.mstep()
            self.i += 1

        def with_queue(self, q):
            """The same bank and buffers, mixing into another queue (the like-for-like pass without the reduce)."""
            o = OscBank.__new__(OscBank)
            o.__dict__.update(self.__dict__)
            o.q, o.mstep = q, MixdownStep(o.render_mix, q)
            return o

    def time_steps(fn, n, warm=20):
        """ms per call of fn over n back-to-back calls (HIP events on the launch stream)."""
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def build_workload(workload, mixdown, mfcc_method="sparse", voice_mode=0):
        """One BASELINE config as a dict: step(), samples per step, the dominant kernel and its algorithmic bytes (or flops) per
        step, the CPU baseline closure, the mix queue(s)."""
        queue = None
        local_queue = None  # the same mix queue without a communicator (device copies): the step with the reduce taken out
        W = {}
        if workload == "config2":
            # fused: the render leaves one partial mix row per workgroup of 256 voices in the queue's slot and the queue adds the rows of
            # a whole batch on ITS stream in front of the reduce -- the render stream carries one kernel per block
            groups2 = L.mxg_osc_mix_groups(V) if mixdown == "fused" else 1
            if mixdown != "off":
                queue = make_queue(B * 2, args.mix_depth, groups2)
            bank = OscBank(V, nbuf, mixdown, queue, lo, V * world)
            if queue is not None and world > 1:
                local_queue = RcclMixQueue(None, B * 2, args.mix_depth, 0, stream, groups=groups2)

            def cpu(target_s=6.0):
                return _baseline(lambda o, n, th: o.time_osc(wf, bank.freq_h, n, threads=th), lambda n: V * n, (256, 1 << 18),
                                 "samples per voice", "%d voices maxiOsc::%s, voice-inner loop" % (V, args.waveform), target_s)
            W = dict(bank=bank, step=bank.step, samples=V * B, dominant=bank.dominant, algo_bytes=bank.algo, dtype="f64", cpu=cpu,
                     tag="configs[1]: %d-voice maxiOsc::%s bank, block 512, out[n][v] stored%s, %d rotated buffers"
                         % (V, args.waveform, {"fused": " + fused stereo mixdown", "separate": " + K3 mixdown", "off": ""}[mixdown], nbuf),
                     local_step=bank.with_queue(local_queue).step if local_queue is not None else None,
                     workload="configs[1]: %d-voice maxiOsc::%s wavetable bank per GPU, block=512, fp64 out[n][v] stored%s, output "
                              "rotated over %d block buffer(s) = %.2f GB touched before a line is rewritten"
                              % (V, args.waveform, {"fused": " + fused maxiMix::stereo mixdown", "separate": " + K3 mixdown",
                                                    "off": ""}[mixdown], nbuf, nbuf * V * B * 8 / 1e9))
        elif workload == "tables":
            # EXTENSION (north_star's "HBM-read roofline on the wavetable path"): every voice its own 514-entry table,
            # 131 072 voices (>= 10^5), the fused maxiMix::stereo mixdown as output -- the READ stream is the bound
            Vt = 131072
            ft = torch.from_numpy(bank_parameters(0, Vt, Vt)[0]).to(dev)
            pt = torch.from_numpy(bank_parameters(0, Vt, Vt)[1]).to(dev)
            one = torch.sin(2 * np.pi * torch.arange(514, dtype=torch.float64, device=dev) / 512.0)
            tabs = (one[None, :] * (1.0 + 1e-3 * torch.arange(Vt, dtype=torch.float64, device=dev)[:, None] / Vt)).contiguous()  # 539 MB
            pht = torch.zeros(Vt, dtype=torch.float64, device=dev)
            hdt = torch.zeros(Vt, dtype=torch.float64, device=dev)
            Gt = L.mxg_osc_tables_groups(Vt)
            rowst = torch.zeros((Gt, B, 2), dtype=torch.float64, device=dev)
            mixt = torch.zeros((B, 2), dtype=torch.float64, device=dev)

            tk = dict(kv.split("=") for kv in args.tune)  # (A/B switches of this bench, not library knobs: tables_ahead, tables_sum)
            # (measured, one box, 131 072 voices: serial 146 us; pipelined 120; row sum inside the kernel +8.5 us on either -- its two
            # cross-XCD hand-offs are write-through round trips -- so the step keeps the 6-us row-sum kernel: profiles/r06_k1t.md)
            t_ahead, t_sum = int(tk.get("tables_ahead", 1)), int(tk.get("tables_sum", 0))

            # tables_queue=1 (default): the rows go into a slot of a GROUPED mix queue, as K1m's and K2f's do -- the queue adds the rows of 16
            # blocks with one kernel on ITS stream (and would reduce them over RCCL); tables_queue=0: the row-sum kernel behind every render
            t_queue = int(tk.get("tables_queue", 1)) and not t_sum
            if t_queue:
                queue = make_queue(B * 2, args.mix_depth, Gt)

            def render_tables(rows_ptr, mix_ptr):
                # round 6: pipelined blocks -- the next block's phase recurrence walks beside this block's table traffic (tables_ahead=0:
                # round 5's marks kernel in front of every render); tables_sum=1: the row sum inside the render kernel
                chk(L.mxg_osc_render_tables_ex(Vt, B, ft.data_ptr(), tabs.data_ptr(), pht.data_ptr(), hdt.data_ptr(), None, pt.data_ptr(),
                                               rows_ptr, mix_ptr, t_ahead, stream), "mxg_osc_render_tables_ex")

            def step_plain():
                render_tables(rowst.data_ptr(), mixt.data_ptr() if t_sum else None)
                if not t_sum:
                    chk(L.mxg_mix_rows_sum(Gt, B * 2, rowst.data_ptr(), mixt.data_ptr(), stream), "mxg_mix_rows_sum")
            step_tables = MixdownStep(lambda slot: render_tables(slot, None), queue) if t_queue else step_plain
            W = dict(step=step_tables, samples=Vt * B, dominant="osctab_kernel", algo_bytes=Vt * (514 * 8.0 + 24.0), dtype="f64", cpu=None,
                     tag="EXTENSION: %d-voice sinebuf bank, one 514-entry table PER VOICE (HBM-read form), fused mixdown out" % Vt,
                     local_step=None,
                     workload="EXTENSION of configs[1] (not in the reference, which has one shared sineBuffer): %d-voice maxiOsc::sinebuf bank "
                              "with a 514-entry table PER VOICE (4112 B read per voice and block, 8.03 B per sample), block=512, fused "
                              "maxiMix::stereo mixdown as output (no per-voice store): the HBM-READ form of the wavetable path" % Vt)
        elif workload == "sample_bank":
            # north_star: "coalesced HBM reads of ... maxiSample buffers" -- measured where the samples cannot live on chip: 65 536 play
            # heads of maxiSample::playAtSpeed (C:1060-1075) over ONE 8.6 GB sample, each head in its own region (16 368 elements apart),
            # speeds spread over [0.5, 1.5] (mean 1: 8 B of sample read + 8 B written per output sample).  A region is revisited after
            # ~32 blocks = 8.6 GB of other traffic: every sample byte comes from HBM.
            Vs, Ls = 65536, (1 << 30) - (1 << 20)
            arena = torch.empty(Ls + 64, dtype=torch.float64, device=dev)  # the guarded layout of mxg_sample_upload: [-4, len + 5] valid
            arena[:8].zero_(); arena[Ls:].zero_()
            for c0 in range(0, Ls, 1 << 26):
                c1 = min(Ls, c0 + (1 << 26))
                xs = torch.arange(c0, c1, dtype=torch.float64, device=dev)
                arena[8 + c0:8 + c1] = torch.frac(xs * 0.3183098861837907) - 0.5
                del xs
            d_smp = arena.data_ptr() + 64
            vv = np.arange(Vs)
            speed_h = 0.5 + (vv * 40503 % Vs) / float(Vs)          # a permutation of the grid over [0.5, 1.5)
            pos_h = (vv * (Ls // Vs)).astype(np.float64) + 0.25
            d_speed = mx.DeviceBuffer.from_numpy(speed_h)
            d_pos = mx.DeviceBuffer.from_numpy(pos_h)
            outs_s = [mx.DeviceBuffer((B, Vs), zero=False) for _ in range(max(1, -(-(1 << 31) // (Vs * B * 8))))]
            kb = [0]

            def step_sample():
                o = outs_s[kb[0] % len(outs_s)]
                chk(L.mxg_sample_render(4, Vs, B, d_smp, Ls, 44100, d_speed.ptr, 0, None, None, d_pos.ptr, o.ptr, stream), "mxg_sample_render")
                kb[0] += 1

            def cpu(target_s=6.0):
                # (the plain-C port: V heads over ONE sample array, like the bank; the compiled reference's harness gives every voice its
                # own maxiSample, i.e. a copy of the sample per voice, which would time the copies)
                from oracle import pyoracle
                o, kind = pyoracle.port(), "port"
                o.settings(44100, 2, 1024)
                Vc, Nc, Lc = 4096, 4096, 1 << 22
                smp_c = np.modf(np.arange(Lc) * 0.3183098861837907)[0] - 0.5
                pos_c = (np.arange(Vc) * (Lc // Vc)).astype(np.float64) + 0.25
                o.sample(4, smp_c, 16, pos_c, a=speed_h[:Vc])
                t0 = time.perf_counter()
                o.sample(4, smp_c, Nc, pos_c, a=speed_h[:Vc])
                dt = time.perf_counter() - t0
                return {"unit": "Msamples/s", "kind": kind, "value": round(Vc * Nc / dt / 1e6, 2), "cores": 1,
                        "sample": "maxiSample::playAtSpeed, %d heads x %d samples over a 4 Mi-element sample, one thread; %.2f s wall" % (Vc, Nc, dt)}
            W = dict(step=step_sample, samples=Vs * B, dominant="sample_parts_kernel", algo_bytes=(16.0 + 24.0 / B) * Vs * B, dtype="f64", cpu=cpu,
                     local_step=None, keep=(arena,), traffic_key="sample_parts_kernel@sample_bank",
                     tag="maxiSample::playAtSpeed, %d heads over one 8.6 GB sample (HBM-resident: each head its own region), block 512" % Vs,
                     workload="north_star's sample path where the sample cannot live on chip: %d play heads of maxiSample::playAtSpeed over one "
                              "%.1f GB sample, heads %d elements apart, speeds over [0.5, 1.5) (mean 1: 8 B read + 8 B written per sample), "
                              "block=512, output rotated over %d block buffers" % (Vs, Ls * 8 / 1e9, Ls // Vs, len(outs_s)))
        elif workload == "config3":
            K = 128
            mode = voice_mode
            vb = mx.maxiVoiceBank(V, stream=stream)
            vb.env.setAttack(10); vb.env.setDecay(100); vb.env.setSustain(0.5); vb.env.setRelease(500)
            f3 = np.minimum(freq_h, 5000.0)
            cu3, rs3 = (200 + 4 * f3) if mode == 0 else np.full(V, 10000.0), 1.0 + (np.arange(lo, hi) % 16)
            outs3 = [mx.DeviceBuffer((B, V), zero=False) for _ in range(nbuf)]
            vb.render(mode, f3, cu3, rs3, np.zeros(1, np.int32), 1, out=mx.DeviceBuffer((1, V)))
            vf, vcu, vrs, vcoef, _ = vb._keep
            vpar, vhold = vb.env._params()
            gate = mx.DeviceBuffer.from_numpy(((np.arange(K * B) % 44100) < 22050).astype(np.int32))
            pan3 = mx.DeviceBuffer.from_numpy(pan_h)
            blk = [0]
            # fused (round 6): K2f leaves one partial mix row per workgroup of 256 voices in the grouped queue's slot, as K1m does
            groups3 = L.mxg_osc_mix_groups(V) if mixdown == "fused" else 1
            if mixdown != "off":
                queue = make_queue(B * 2, args.mix_depth, groups3)
                if world > 1:
                    local_queue = RcclMixQueue(None, B * 2, args.mix_depth, 0, stream, groups=groups3)

            def voice_block():
                o3 = outs3[blk[0] % nbuf]
                chk(L.mxg_voice_render(mode, V, B, vf.ptr, vcu.ptr, vrs.ptr, vcoef.ptr if vcoef is not None else None,
                                       gate.ptr + 4 * (blk[0] % K) * B, 0, vpar.ptr, vhold.ptr, vb.osc_state.ptr, vb.flt_state.ptr,
                                       vb.env.dstate.ptr, vb.env.istate.ptr, o3.ptr, stream), "mxg_voice_render")
                blk[0] += 1
                return o3

            def render_mix3(slot):
                if mixdown == "fused":  # render + store + per-workgroup mix rows in one pass; the (grouped) queue adds the rows
                    o3 = outs3[blk[0] % nbuf]
                    chk(L.mxg_voice_render_mix_rows(mode, V, B, vf.ptr, vcu.ptr, vrs.ptr, vcoef.ptr if vcoef is not None else None,
                                                    gate.ptr + 4 * (blk[0] % K) * B, 0, vpar.ptr, vhold.ptr, vb.osc_state.ptr,
                                                    vb.flt_state.ptr, vb.env.dstate.ptr, vb.env.istate.ptr, o3.ptr, pan3.ptr, slot, stream),
                        "mxg_voice_render_mix_rows")
                    blk[0] += 1
                    return
                o3 = voice_block()  # K2f then K3 re-reading the block
                chk(L.mxg_mix_stereo(V, B, o3.ptr, pan3.ptr, slot, stream), "mxg_mix_stereo")
            step = voice_block if mixdown == "off" else MixdownStep(render_mix3, queue)

            def cpu(target_s=6.0):
                return _baseline(lambda o, n, th: o.time_voice(mode, f3, cu3, rs3, n, threads=th), lambda n: V * n, (32, 1 << 16),
                                 "samples per voice", "%d voices saw->lores->adsr (mode %d), voice-inner loop of 15.polysynth" % (V, mode), target_s)
            fps3 = None
            if mode == 1:  # report fp64 FLOP/s against the 78.6 TFLOP/s vector peak; flops per sample from the counters
                try:
                    fps3 = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["voice_kernel_modB"]["fp64_flops_per_sample"]
                except Exception:
                    fps3 = None
            algo3 = (8.0 + 176.0 / B) * V * B
            if mixdown == "fused":
                algo3 += 8.0 * V + 16.0 * B * (V / 256.0)  # + pan read, per-workgroup mix rows written
            W = dict(step=step, samples=V * B, dominant="voice_kernel", algo_bytes=algo3, dtype="f64", cpu=cpu,
                     traffic_key="voice_kernel@config3_modB" if mode == 1 else ("voice_kernel@config3_mix" if mixdown == "fused" else None),
                     tag="configs[2]: %d voices saw->lores->adsr (mode %s), block 512%s" % (V, "AB"[mode], {"fused": " + fused stereo mixdown",
                                                                                                       "separate": " + K3 mixdown", "off": ""}[mixdown]),
                     fp64_flops=fps3 * V * B if fps3 else None,
                     fp64_note="%.1f fp64 flops per sample = (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 FMA_F64) x 64 lanes / samples, rocprofv3 --pmc of this "
                               "workload (profiles/pmc_traffic.json); compute-bound: cos / pow / sqrt per sample on the device" % fps3 if fps3 else None,
                     local_step=MixdownStep(render_mix3, local_queue) if local_queue is not None else None,
                     workload="configs[2]: fused subtractive voice maxiOsc::saw -> maxiFilter::lores -> maxiEnv::adsr (mode %d), %d voices "
                              "per GPU, block=512, gate(n) = (n mod 44100) < 22050 cycled over 128 blocks, output rotated over %d block "
                              "buffer(s)%s" % (mode, V, nbuf, {"fused": " + fused maxiMix::stereo mixdown", "separate": " + K3 mixdown",
                                                               "off": ""}[mixdown]))
        elif workload == "config4":
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
            mfma = mfcc_method == "mfma-gemm"   # the two-kernel dense route
            mm = mfcc_method == "mfma"          # the fused kernel's matrix-pipe form
            kdim = 512 if args.mfma_fullk else 256  # bins the MFMA GEMM contracts over (weights beyond bin 232 are zero)
            if mfma:
                L.mxg_tune(b"mfcc_mfma_fullk", 1 if args.mfma_fullk else 0)
            mm_batches = 0
            if mm:
                import ctypes as _ct
                _nb = (_ct.c_int * 6)()
                _pl = mx.maxiMFCC(); _pl.setup(512, 42, 13, 20.0, 20000.0)
                mm_batches = L.mxg_mfcc_plan_matrix_tables(_pl.plan, _nb, None, None, 0, None)
                _pl.close()
            mfcc = torch.empty((NF, 13), dtype=torch.float64, device=dev)
            fplan = mx.maxiFFT(); fplan.setup(1024, 1024, 1024)
            mplan = mx.maxiMFCC(); mplan.setup(512, 42, 13, 20.0, 20000.0)
            fused_ok = hasattr(L, "mxg_fft_mfcc_batch") and not mfma
            mags = None if fused_ok else torch.empty((NF, 512), dtype=torch.float32, device=dev)

            fused_mel_knob = 3 if mm else (2 if mfcc_method == "walk" else int(dict(kv.split("=") for kv in args.tune).get("fused_mel", 0)))

            def step():
                if fused_ok:
                    L.mxg_tune(b"fused_mel", fused_mel_knob)  # (the knob is process-wide and the default line measures both forms)
                    chk(L.mxg_fft_mfcc_batch(fplan.plan, mplan.plan, sig.data_ptr(), 1024, NF, None, None, None, mfcc.data_ptr(), stream),
                        "mxg_fft_mfcc_batch")
                else:
                    chk(L.mxg_fft_batch(fplan.plan, sig.data_ptr(), 1024, NF, None, None, mags.data_ptr(), None, stream), "fft")
                    chk(L.mxg_mfcc_batch(mplan.plan, mags.data_ptr(), 512, NF, None, None, mfcc.data_ptr(), 1 if mfma else 0, stream), "mfcc")

            sig_h = [None]

            def cpu(target_s=6.0):
                cap = 262144 if target_s > 2 else 32768
                if sig_h[0] is None or sig_h[0].size < 1024 * cap:
                    sig_h[0] = sig[:1024 * cap].cpu().numpy()
                return _baseline(lambda o, n, th: o.time_spectral(sig_h[0][:n * 1024], threads=th), lambda n: n * 1024, (512, cap),
                                 "frames", "maxiFFT(1024,1024,1024)::process per sample + maxiMFCC(512,42,13)::mfcc per frame (mfcctest loop)",
                                 target_s)
            def read_ceiling():  # the kernel's own input stream with everything but the loads removed (csrc/calib.hip), same buffer
                sink = torch.zeros(8, dtype=torch.float64, device=dev)
                out = {}
                for name, (w, fl, blk, blks) in {"8 B plain loads, 512 x 256 threads (the launch shape)": (8, 0, 256, 512),
                                                 "8 B non-temporal loads, 1024 x 512 threads": (8, 1, 512, 1024)}.items():
                    out[name] = time_steps(lambda: chk(CAL.mxg_calib_read_ex(sig.data_ptr(), NF * 4096, w, fl, 1, blk, blks, sink.data_ptr(),
                                                                           stream), "calib_read"), 6, warm=3)
                return out
            W = dict(step=step, samples=NF * 1024, dtype="f32 (FFT) / f64 (MFCC)", cpu=cpu, read_ceiling=read_ceiling if fused_ok else None,
                     dominant="mfcc_mfma_gemm_kernel" if mfma else ("fft_mfcc_kernel" if fused_ok else "fft1024_kernel"),
                     traffic_key="fft_mfcc_kernel@config4_mfma" if mm else ("fft_mfcc_kernel@config4_walk" if mfcc_method == "walk" else None),
                     tag="configs[3]: maxiFFT(1024)+maxiMFCC(512,42,13) x %d frames, %s" % (NF, "FFT kernel + dense MFMA GEMM" if mfma else (
                         "fused, exact FFT, matrix-pipe mel+DCT" if mm else ("fused, exact FFT, exact band sums (walk), matrix DCT" if
                                                                            mfcc_method == "walk" else "fused, library default form"))),
                     algo_bytes=4200.0 * NF if fused_ok else 6144.0 * NF,
                     # MFMA flops ISSUED: 2 x K x 48 (42 filters padded to 3 column blocks of 16) per frame
                     mfma_flops=2.0 * kdim * 48 * NF if mfma else None,
                     mfma_note="issued 2*%d*48 flops/frame (useful dense 2*512*42 = 43008)" % kdim if mfma else None,
                     # the fused matrix form stays HBM-bound (4200 B per frame); what its matrix pipe does is reported beside the roofline:
                     # per 8 frames 4 x batches (mel) + 24 (DCT) v_mfma_f64_4x4x4_4b instructions of 512 flops each
                     matrix_flops=(4 * mm_batches + 24) * 512.0 * NF / 8 if mm else (24 * 512.0 * NF / 8 if mfcc_method == "walk" else None),
                     workload="configs[3]: maxiFFT(1024,1024,1024) + maxiMFCC(512,42,13,20,20000) over %d frames per GPU per step, %s"
                              % (NF, "FFT kernel + dense fp64 MFMA mel contraction (K = %d bins)" % kdim if mfma else
                                 ("one fused kernel, exact FFT, mel contraction + DCT on the matrix pipe (v_mfma_f64_4x4x4_4b, banded per quad of "
                                  "filters: band sums within 1e-13, mfcc within 1e-11 of the exact form)" if mm else
                                  ("one fused kernel, exact FFT, sparse mel walk (band sums bit for bit the reference's sequential sums), DCT on "
                                   "the matrix pipe (fused_mel 2)" if mfcc_method == "walk" else
                                   ("one fused kernel in the library's default form (exact FFT: magnitudes bit-exact; only the coefficients "
                                    "are requested, so the mel contraction and the DCT run on the matrix pipe -- fused_mel 0 = 3 here; "
                                    "band sums within 1e-13, mfcc within the device log's 1e-12 of the reference)" if fused_ok
                                    else "FFT kernel + exact sparse mel walk")))))
        else:  # config5
            S, T, Ls = 2048, 70560, 4410000
            rng5 = np.random.default_rng(0x4D415849)
            n5 = np.arange(Ls)
            smp = 0.5 * np.sin(2 * np.pi * 110 * n5 / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n5 / 44100) + 0.05 * rng5.uniform(-1, 1, Ls)
            sb5 = mx.maxiSampleBank(1, stream=stream); sb5.setSample(smp)
            gb = mx.maxiTimeStretchBank(S, sb5, "hann", stream=stream)
            lo5, hi5 = shard_range(rank, world, S)
            pos5, speed5, pan5_h = stream_parameters(lo5, hi5, S * world)
            gb.setPosition(pos5)
            sp5 = mx.DeviceBuffer.from_numpy(speed5)
            pan5 = mx.DeviceBuffer.from_numpy(pan5_h)
            out5 = mx.DeviceBuffer((T, S), zero=False)
            plan5 = gb._plan(0.05)
            if mixdown != "off":
                queue = make_queue(T * 2, 1)  # one [T][2] = 1.13 MB reduce per render
                if world > 1:
                    local_queue = RcclMixQueue(None, T * 2, 1, 0, stream)

            def grains():
                chk(L.mxg_granular_render(plan5, 0, S, T, sb5.d_samples, Ls, 4, sp5.ptr, None, None, None, 0,
                                          gb.state.ptr, gb.grains.ptr, out5.ptr, stream), "mxg_granular_render")

            def render_mix5(slot):
                if mixdown == "fused":  # the unit kernel mixes each 64 x 64 tile while it is in LDS
                    chk(L.mxg_granular_render_mix(plan5, 0, S, T, sb5.d_samples, Ls, 4, sp5.ptr, None, None, None, 0, gb.state.ptr,
                                                  gb.grains.ptr, out5.ptr, pan5.ptr, slot, stream), "mxg_granular_render_mix")
                else:
                    grains()
                    chk(L.mxg_mix_stereo(S, T, out5.ptr, pan5.ptr, slot, stream), "mxg_mix_stereo")
            step = grains if mixdown == "off" else MixdownStep(render_mix5, queue)

            def cpu(target_s=6.0):
                Sc = 2048  # streams of the bounded sample: this rank's whole share
                return _baseline(lambda o, n, th: o.time_grains(smp, speed5[:Sc], pos5[:Sc], n, threads=th), lambda n: Sc * n * 4,
                                 (256, 8 * 70560), "samples per stream", "%d maxiTimeStretch<hann> streams, play(speed,0.05,4), stream-inner "
                                 "loop, 4 grain-samples per stream-sample" % Sc, target_s)
            W = dict(step=step, samples=S * T * 4, dominant="granular_unit_kernel", algo_bytes=(8.0 * 4 + 8.0) * S * T, dtype="f64", cpu=cpu,
                     tag="configs[4]: %d maxiTimeStretch<hann> streams x %d samples, 4 overlaps%s" % (S, T, "" if mixdown == "off" else " + stereo mixdown"),
                     # what the render actually moves, all of it from L2 / Infinity Cache (the 35 MB sample and the 17.6 KB window are resident):
                     # per stream-sample 4 live grains x (buffer[a], buffer[a+1] = 16 B + 8 B of window) + the 8-byte store
                     l2_bytes=(4 * 24.0 + 8.0) * S * T,
                     local_step=MixdownStep(render_mix5, local_queue) if local_queue is not None else None,
                     workload="configs[4]: %d maxiTimeStretch<hann> streams per GPU x %d samples per step, grainLength 0.05, overlaps 4 "
                              "(4 live grains per stream-sample counted)%s" % (S, T, "" if mixdown == "off" else
                                                                              ", maxiMix::stereo to [T][2] + one RCCL reduce per render"))

        W["queue"], W["local_queue"], W["mixdown"] = queue, local_queue, mixdown
        return W

    W = build_workload(args.workload, mixdown, args.mfcc_method, args.voice_mode)
    queue, local_queue = W["queue"], W["local_queue"]
    bank = W.get("bank")
    step = W["step"]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def read_kernels(nsteps):
        # every event pair carries some cost of the two markers themselves: measured on empty pairs and REPORTED (subtracting it
        # over-corrects -- with a kernel between the markers most of it is hidden -- and made kernel_ms disagree with rocprofv3)
        ovh = ctypes.c_double(0)
        chk(L.mxg_prof_overhead_ms(stream, 256, ctypes.byref(ovh)), "mxg_prof_overhead_ms")
        res = {"_event_pair_overhead_ms": ovh.value}
        for i in range(L.mxg_prof_count()):
            lab, ms, cnt = ctypes.c_char_p(), ctypes.c_double(0), ctypes.c_size_t(0)
            chk(L.mxg_prof_read(i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(cnt)), "mxg_prof_read")
            if cnt.value:
                res[lab.value.decode()] = {"ms": ms.value / cnt.value, "launches_per_step": cnt.value / float(nsteps)}
        return res

    # Untimed clock ramp: an idle MI355X sits in a low-power state (sclk ~500 MHz) and takes milliseconds of continuous
    # work to reach its sustained clocks; a short --steps run would otherwise time the ramp instead of the kernel.
    # With more than one rank the NUMBER of ramp steps must be the same everywhere -- every full batch of the mix queue is a collective,
    # matched by order: ranks that leave a wall-clock loop after different step counts leave unmatched reduces behind and the run hangs
    # at its first fence -- so rank 0's clock decides, chunk by chunk, and tells the others.
    t_ramp = time.perf_counter()
    while True:
        for _ in range(50 if args.workload in ("config2", "config3") else 1):
            step()
        torch.cuda.synchronize()
        go = time.perf_counter() - t_ramp < 0.3
        if world > 1:
            flag = torch.tensor([1 if go else 0], device=dev, dtype=torch.int32)
            dist.broadcast(flag, src=0)
            go = bool(int(flag.item()))
        if not go:
            break
    for _ in range(args.warmup):
        step()
    if queue is not None:
        queue.flush()
    inline = args.kernel_events == "inline"
    L.mxg_prof_reset()
    L.mxg_prof_enable(1 if inline else 0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    if queue is not None:
        queue.flush()  # the last (partial) batch's reduce; the launch stream waits for every outstanding reduce
    ev1.record()
    fence()
    t1 = time.perf_counter()
    L.mxg_prof_enable(0)
    if os.environ.get("MXG_PRINT_PACE") and args.workload == "config2":  # (diagnostics: K1's pace controllers, csrc/mxg_pace.h)
        buf = (ctypes.c_uint * 128)()
        L.mxg_debug_osc_pace(ctypes.c_void_p(stream), buf)
        w = list(buf)
        for f in range(16):
            q = w[8 * f: 8 * f + 8]
            if f == 14 and q[1]:
                print("trial[sinebuf, headline size]: verdict %d (1 free-running, else the period)  launches %d  ticks per phase of 32 launches: free-running %d, the best period %d took %d" % (
                    q[0], q[1] & 0xffff, q[4], q[6], q[5]), file=sys.stderr)
            elif q[0]:
                print("pace[waveform %d]: P %d  window %d lates %d booted %d  last mean lateness %d" % (
                    f, q[0], q[1] & 255, (q[1] >> 8) & 255, (q[1] >> 16) & 255, q[7]), file=sys.stderr)
    if os.environ.get("MXG_PRINT_PACE") and args.workload == "config3":  # (diagnostics: K2f's pace controllers, csrc/mxg_pace.h)
        buf = (ctypes.c_uint * 32)()
        L.mxg_debug_voice_pace(ctypes.c_void_p(stream), buf)
        w = list(buf)
        for f, name in enumerate(("mode A", "mode B", "mode A + mix", "mode B + mix")):
            q = w[8 * f: 8 * f + 8]
            if q[0]:
                print("pace[%s]: P %d  window %d lates %d booted %d  last mean lateness %d" % (
                    name, q[0], q[1] & 255, (q[1] >> 8) & 255, q[1] >> 16, q[7]), file=sys.stderr)
    elapsed = t1 - t0
    step_ms_events = ev0.elapsed_time(ev1) / args.steps
    kernels = read_kernels(args.steps) if inline else {}
    if args.kernel_events == "pass":  # same steps again, with the per-kernel events on, outside the timed region
        psteps = min(args.steps, 200)
        L.mxg_prof_reset()
        L.mxg_prof_enable(1)
        for i in range(psteps):
            step()
        if queue is not None:
            queue.flush()
        torch.cuda.synchronize()
        L.mxg_prof_enable(0)
        kernels = read_kernels(psteps)
    event_overhead = kernels.pop("_event_pair_overhead_ms", None)
    dom = W["dominant"]
    dom_ms_events = kernels[dom]["ms"] if dom in kernels else None
    # ONE estimator for the dominant kernel's average launch duration (VERDICT r04 weak #8a), the same for every entry of the line:
    # the GPU-side step time (two HIP events around the timed region's K back-to-back steps, no per-kernel markers inside) times
    # the kernel's share of the step's per-kernel event time, per launch.  For a single-kernel step this is step / launches -- an
    # UPPER bound on the kernel's duration (it carries the launch gaps); the raw per-launch event figure (which carries a marker
    # pair, ~2-4 us) stays beside it as kernel_ms_events.
    dom_ms, dom_ms_step = share_estimate(kernels, dom, step_ms_events)

    # ---- N > 1: the SAME step with the reduce taken out (mix queue without a communicator), same run, same buffers ----
    local_ms = None
    if W.get("local_step") is not None:
        fence()
        ls = W["local_step"]
        nl = max(10, min(args.steps, 400))
        local_ms = time_steps(lambda: ls(), nl, warm=min(20, nl))
        local_queue.flush()
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([elapsed, dom_ms, local_ms or 0.0], dtype=torch.float64)
        if not args.share_gpu:
            t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dom_ms, local_ms = float(t[0]), float(t[1]), (float(t[2]) or None)

    # ---- N = 1, config 2: what the headline is made of -----------------------------------------------------------------
    extras = {}
    if world == 1 and W.get("read_ceiling") and not args.no_extras and hasattr(CAL, "mxg_calib_read_ex"):
        rc = W["read_ceiling"]()
        best = min(rc, key=rc.get)
        nb_r = (1 << 20) * 4096
        extras["read_ceiling"] = {"GB/s": round(nb_r / rc[best] / 1e6, 1), "pattern": best,
                                  "all_GB/s": {k_: round(nb_r / v_ / 1e6, 1) for k_, v_ in rc.items()},
                                  "note": "pure load streams of the fused kernel's shape (persistent wavefronts, groups of 8 consecutive 4 KB frames, "
                                          "one frame ahead in flight) over this run's 4.3 GB signal; profiles/r03_read_ceiling.md has the full family"}
    if world == 1 and args.workload == "config2" and mixdown == "off" and not args.tune and not args.no_extras:
        nb = V * B * 8
        n_x = max(50, min(args.steps, 500))
        # (a) the same kernel into ONE reused block buffer (partly absorbed by the 256 MB Infinity Cache)
        one = OscBank(V, 1, "off", None)
        extras["single_buffer_ms"] = time_steps(one.step, n_x)
        del one
        # (b) the measured write ceiling: the best pure store streams of csrc/calib.hip over the SAME rotating buffers
        ceil_ms = {}
        for name, (w, fl, pat, blk) in {"grid-stride fill, 16 B plain": (16, 0, 0, 256), "column walk, 8 B plain": (8, 0, 1, 256),
                                        "column walk, 16 B plain": (16, 0, 1, 256), "column walk, 16 B sc1 (write-through)": (16, 2, 1, 256),
                                        "column walk, 16 B sc1 nt": (16, 4, 1, 256)}.items():
            k = [0]

            def fill():
                chk(CAL.mxg_calib_fill_ex(bank.outs[k[0] % len(bank.outs)].data_ptr(), B, V * 8, w, fl, pat, blk, 0, 0, stream), "calib")
                k[0] += 1
            ceil_ms[name] = time_steps(fill, n_x)
        best = min(ceil_ms, key=ceil_ms.get)
        extras["write_ceiling"] = {"GB/s": round(nb / ceil_ms[best] / 1e6, 1), "pattern": best,
                                   "all_GB/s": {k_: round(nb / v_ / 1e6, 1) for k_, v_ in ceil_ms.items()},
                                   "note": "the fastest of the pure store streams of csrc/calib.hip (no arithmetic, a lane owns 8 or 16 bytes of a row "
                                           "and walks down the rows) over this run's rotating block buffers; profiles/r03_write_ceiling.md has the full family"}
        # (c) the north star's bank size: >= 10^5 voices (131 072), block buffers rotated the same way
        if not args.voices:
            V2 = 131072
            nb2 = V2 * B * 8
            b2 = OscBank(V2, max(1, -(-(1 << 31) // nb2)), "off", None)
            ms2 = time_steps(b2.step, n_x, warm=int(os.environ.get("MXG_NS_WARM", "96")))  # (this bank runs on the paced schedule, csrc/mxg_pace.h: its controller finds the period within ~15 launches of a stream's first)
            if os.environ.get("MXG_PRINT_PACE"):
                buf = (ctypes.c_uint * 128)()
                L.mxg_debug_osc_pace(ctypes.c_void_p(stream), buf)
                q = list(buf)[8 * 8: 8 * 8 + 8]
                print("pace[north_star_bank]: P %d  window %d lates %d booted %d  last mean lateness %d; %.2f us" % (
                    q[0], q[1] & 255, (q[1] >> 8) & 255, (q[1] >> 16) & 255, q[7], ms2 * 1e3), file=sys.stderr)
            k2 = [0]

            def fill2():
                chk(CAL.mxg_calib_fill_ex(b2.outs[k2[0] % len(b2.outs)].data_ptr(), B, V2 * 8, 16, 2, 1, 256, 0, 0, stream), "calib")
                k2[0] += 1
            c2 = time_steps(fill2, n_x)
            extras["north_star_bank"] = {
                "voices": V2, "block": B, "ms_per_step": round(ms2, 5), "value": round(V2 * B / ms2 / 1e3, 1), "unit": "Msamples/s",
                "block_buffers": len(b2.outs), "frac_hbm_peak": round(b2.algo / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "write_ceiling_GB/s": round(nb2 / c2 / 1e6, 1), "write_ceiling_pattern": "column walk, 16 B sc1 (write-through)", "frac_of_measured_write_ceiling": round(c2 / ms2 * b2.algo / nb2, 4),
                "realtime_factor_at_44k1": round(B / 44100.0 / (ms2 * 1e-3), 1)}
            del b2

    # ---- N = 1 default run: every other GPU config of BASELINE.json, in the same line ("configs") -------------------------------
    def quick(Wq, steps, warm):
        """One config measured the short way: `steps` timed steps between two events on the launch stream (queue flushed inside the
        timed region), then a pass with the library's per-kernel events on; the roofline of its dominant kernel as for the headline."""
        q, st = Wq["queue"], Wq["step"]
        t_r = time.perf_counter()  # the clock ramp again: building the workload left the GPU idle for a while
        while time.perf_counter() - t_r < 0.25:
            for _ in range(20 if Wq["samples"] < (1 << 28) else 1):
                st()
            torch.cuda.synchronize()
        for _ in range(warm):
            st()
        if q is not None:
            q.flush()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            st()
        if q is not None:
            q.flush()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / steps
        psteps = min(steps, 100)
        L.mxg_prof_reset()
        L.mxg_prof_enable(1)
        for _ in range(psteps):
            st()
        if q is not None:
            q.flush()
        torch.cuda.synchronize()
        L.mxg_prof_enable(0)
        kern = read_kernels(psteps)
        kern.pop("_event_pair_overhead_ms", None)
        dom_k = Wq["dominant"]
        lps = kern.get(dom_k, {}).get("launches_per_step", 1.0)
        k_ms, _ = share_estimate(kern, dom_k, ms)
        ent = {"workload": Wq["workload"], "tag": Wq.get("tag"), "steps": steps, "warmup": warm, "ms_per_step": round(ms, 5),
               "value": round(Wq["samples"] / ms / 1e3, 1), "unit": "Msamples/s", "dtype": Wq["dtype"]}
        tr = None
        try:  # (a variant of a workload keeps its own counter figure under kernel@workload: tools/summarize_rocprof.py)
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            tr = tj.get(Wq.get("traffic_key") or dom_k, tj.get(dom_k))
        except Exception:
            pass
        if Wq.get("mfma_flops"):
            ach_k = Wq["mfma_flops"] / lps / (k_ms * 1e-3) / 1e12
            ent["roofline"] = {"bound": "mfma", "kernel": dom_k, "kernel_ms": round(k_ms, 5), "launches_per_step": round(lps, 3),
                               "flops_per_launch": Wq["mfma_flops"] / lps, "achieved": round(ach_k, 2), "peak": MFMA_F64_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach_k / MFMA_F64_PEAK_TFLOPS, 4), "note": Wq.get("mfma_note"), "traffic": tr}
        else:
            apl = Wq["algo_bytes"] / lps
            ach_k = apl / (k_ms * 1e-3) / 1e9
            ent["roofline"] = {"bound": "hbm", "kernel": dom_k, "kernel_ms": round(k_ms, 5), "launches_per_step": round(lps, 3),
                               "algorithmic_bytes_per_launch": round(apl), "achieved": round(ach_k, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach_k / HBM_PEAK_GBS, 4), "traffic": tr}
            # the whole step (every kernel it launches, gaps included) against the same peak
            ent["roofline"]["step_frac"] = round(Wq["algo_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if Wq.get("l2_bytes"):
                # an L2 / Infinity-Cache gather (PMC HBM traffic well below the algorithmic bytes): the figure to read is the STEP's
                # fraction -- it leads (VERDICT r04 weak #8d); the kernel-level one stays beside it
                ent["roofline"]["l2_gather_GB/s"] = round(Wq["l2_bytes"] / lps / (k_ms * 1e-3) / 1e9, 1)
                ent["roofline"]["kernel_frac"] = ent["roofline"]["frac"]
                ent["roofline"]["frac"] = ent["roofline"]["step_frac"]
                ent["roofline"]["frac_is"] = "step_frac (whole step against the HBM peak): the dominant kernel gathers from L2 / Infinity Cache"
            if Wq.get("matrix_flops"):
                tf = Wq["matrix_flops"] / lps / (k_ms * 1e-3) / 1e12
                mm = {"instruction": "v_mfma_f64_4x4x4_4b_f64", "flops_per_launch": Wq["matrix_flops"] / lps, "achieved_TFLOP/s": round(tf, 2),
                      "peak_TFLOP/s": MFMA_F64_PEAK_TFLOPS, "utilisation": round(tf / MFMA_F64_PEAK_TFLOPS, 4),
                      "note": "the banded contraction issues ~5 x fewer multiply-adds than the dense 512 x 48 product; the kernel is "
                              "bound by its frame loads, not by the matrix pipe"}
                try:
                    mm["pmc"] = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("fft_mfcc_kernel_matrix_pipe")
                except Exception:
                    pass
                ent["roofline"]["matrix_pipe"] = mm
        if Wq.get("fp64_flops"):
            tf = Wq["fp64_flops"] / (ms * 1e-3) / 1e12
            ent["roofline"]["fp64_valu"] = {"flops_per_step": Wq["fp64_flops"], "achieved_TFLOP/s": round(tf, 2), "peak_TFLOP/s": MFMA_F64_PEAK_TFLOPS,
                                            "frac": round(tf / MFMA_F64_PEAK_TFLOPS, 4), "note": Wq.get("fp64_note")}
        ent["roofline"]["estimator"] = ESTIMATOR
        if not args.no_cpu_baseline and Wq.get("cpu"):
            try:
                ent["cpu_baseline"] = Wq["cpu"](1.0)  # the compiled reference on this box's host cores, a ~1 s sample per config
            except Exception as e:
                ent["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        ent["kernels"] = {k: {"ms": round(v["ms"], 5), "launches_per_step": round(v["launches_per_step"], 3)} for k, v in sorted(kern.items())}
        return ent

    configs = {}
    if (world == 1 and args.workload == "config2" and mixdown == "off" and not args.tune and not args.no_extras and not args.voices
            and not args.no_configs):
        for name, (wl, md, meth, st_, wm_) in {
                "config2_mixdown": ("config2", "fused", "sparse", 400, 50),  # the N > 1 step on one GPU: K1m + grouped mix queue, no communicator
                "config2_tables": ("tables", "off", "sparse", 100, 20),  # the per-voice wavetable extension: the HBM-read roofline
                "sample_bank": ("sample_bank", "off", "sparse", 100, 20),  # maxiSample::playAtSpeed over an HBM-resident 8.6 GB sample
                "config3": ("config3", "off", "sparse", 256, 64),
                "config3_mixdown": ("config3", "fused", "sparse", 256, 64),  # config 3's N > 1 step on one GPU: K2f with the mixdown fused + grouped queue
                # cutoff modulated per sample (14.monosynth/main.cpp:53).  128 steps = one whole cycle of the gate (the
                # cost of a block depends on the envelope's stage: half a cycle measured 43 or 59 us depending on where it started)
                "config3_modB": ("config3", "off", "modB", 128, 16),
                "config4": ("config4", "off", "sparse", 10, 2),
                "config4_walk": ("config4", "off", "walk", 10, 2),   # band sums bit-exact (sparse walk), DCT on the matrix pipe
                "config4_mfma": ("config4", "off", "mfma", 10, 2),   # mel contraction + DCT on the matrix pipe, the knob set explicitly
                "config5": ("config5", "fused", "sparse", 10, 2)}.items():
            try:
                Wq = build_workload(wl, md, "sparse" if meth == "modB" else meth, 1 if meth == "modB" else 0)
                configs[name] = quick(Wq, st_, wm_)
                for qq in (Wq["queue"], Wq["local_queue"]):
                    if qq is not None:
                        qq.close()
                del Wq
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            except Exception as e:  # a failing extra must not take the headline with it: it is reported, loudly, in its place
                configs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        if "config2_mixdown" in configs and "ms_per_step" in configs["config2_mixdown"]:
            # against the headline's GPU-side step (events): both are event-timed; the wall-clock ms_per_step of a 20-step run carries the fence
            configs["config2_mixdown"]["step_vs_headline"] = round(configs["config2_mixdown"]["ms_per_step"] / step_ms_events, 4)
        if all("ms_per_step" in configs.get(k_, {}) for k_ in ("config3", "config3_mixdown")):
            configs["config3_mixdown"]["step_vs_config3"] = round(configs["config3_mixdown"]["ms_per_step"] / configs["config3"]["ms_per_step"], 4)

    # ---- N > 1 default run: BASELINE's multi-GPU config (configs[4]: granular time-stretch, streams sharded over the ranks, ONE
    # stereo reduce per render) measured in the same launch: every rank the same fixed number of steps (the reduce is a collective),
    # barrier + synchronize on both sides, the slowest rank's time
    if (world > 1 and args.workload == "config2" and not args.tune and not args.no_configs and not args.voices):
        def sharded(wl, n5, w5, what):
            """One more BASELINE config in the same N-rank launch: units sharded over the ranks, its mixdown reduced through the library's
            queue; every rank the same fixed number of steps, barrier + synchronize on both sides, the slowest rank's wall clock."""
            try:
                W5 = build_workload(wl, "fused")
                for _ in range(w5):
                    W5["step"]()
                W5["queue"].flush()
                fence()
                t5 = time.perf_counter()
                for _ in range(n5):
                    W5["step"]()
                W5["queue"].flush()
                fence()
                e5 = time.perf_counter() - t5
                t = torch.tensor([e5], dtype=torch.float64)
                if not args.share_gpu:
                    t = t.to(dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e5 = float(t[0])
                c5 = {"workload": W5["workload"], "tag": W5.get("tag"), "steps": n5, "warmup": w5, "n_gpus": world, "ms_per_step": round(e5 / n5 * 1e3, 4),
                      "value": round(W5["samples"] * world * n5 / e5 / 1e6, 1), "unit": "Msamples/s", "dtype": W5["dtype"], "scaling": "weak",
                      "note": "whole-job %s per second over %d ranks, barrier-bracketed wall clock, max over ranks" % (what, world)}
                for qq in (W5["queue"], W5["local_queue"]):
                    if qq is not None:
                        qq.close()
                del W5
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                return c5
            except Exception as e:
                return {"error": "%s: %s" % (type(e).__name__, e)}
        configs["config5"] = sharded("config5", 6, 2, "grain-samples")
        # config 3 sharded the same way: K2f with the maxiMix::stereo mixdown fused (mxg_voice_render_mix_rows), one reduce per 16 blocks
        configs["config3"] = sharded("config3", 640, 128, "voice samples")

    value = W["samples"] * world * args.steps / elapsed / 1e6
    dom_launches = kernels.get(dom, {}).get("launches_per_step", 1.0)
    algo_per_launch = W["algo_bytes"] / dom_launches
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        traffic = tj.get(W.get("traffic_key") or dom, tj.get(dom))
    except Exception:
        traffic = None

    if rank == 0:
        if W.get("mfma_flops"):
            ach = W["mfma_flops"] / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F64_PEAK_TFLOPS, 4), "traffic": traffic,
                    "flops_per_launch": W["mfma_flops"], "note": W.get("mfma_note")}
        else:
            ach = algo_per_launch / (dom_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command)"
                    if traffic is not None else None,
                    "algorithmic_bytes_per_launch": round(algo_per_launch)}
            if "single_buffer_ms" in extras:
                roof["frac_single_buffer"] = round(algo_per_launch / (extras["single_buffer_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                roof["single_buffer_note"] = ("the same launches into ONE reused block buffer: the 256 MB Infinity Cache absorbs part of a "
                                              "268 MB block rewritten every step -- not an HBM rate")
            if "write_ceiling" in extras:
                roof["write_ceiling"] = extras["write_ceiling"]
                roof["frac_of_measured_write_ceiling"] = round(ach / extras["write_ceiling"]["GB/s"], 4)
            if "read_ceiling" in extras:
                roof["read_ceiling"] = extras["read_ceiling"]
                roof["frac_of_measured_read_ceiling"] = round(ach / extras["read_ceiling"]["GB/s"], 4)
        if W.get("l2_bytes"):
            l2 = W["l2_bytes"] / dom_launches / (dom_ms * 1e-3) / 1e9
            roof["l2_gather_model"] = {"bytes_per_launch": round(W["l2_bytes"] / dom_launches), "achieved_GB/s": round(l2, 1), "l2_peak_GB/s": 34500.0,
                                       "frac_of_l2_peak": round(l2 / 34500.0, 4),
                                       "note": "this kernel is a latency-bound GATHER from L2 / Infinity Cache (PMC HBM traffic is ~0.11 x the "
                                               "algorithmic bytes): read `frac` (algorithmic bytes against the HBM peak, as the contract asks) "
                                               "together with this L2 figure, not as an HBM utilisation"}
        roof.update(kernel=dom, kernel_ms=round(dom_ms, 5), launches_per_step=round(dom_launches, 3),
                    kernel_ms_events=round(dom_ms_events, 5) if dom_ms_events is not None else None,
                    kernel_ms_step_bound=round(dom_ms_step, 5), estimator=ESTIMATOR,
                    timing="kernel_ms: see estimator; kernel_ms_events: HIP events around each launch on its stream (%s pass; an empty "
                           "event pair measures %.2f us)" % (args.kernel_events, (event_overhead or 0) * 1e3)
                    if dom in kernels else "HIP events around the whole step")
        if not W.get("mfma_flops"):
            # the same algorithmic bytes against the WALL clock of the timed region (what the driver's own clock sees, fences included)
            roof["frac_wall"] = round(W["algo_bytes"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
        res = {
            "metric": "Msamples/s (voice-bank render)", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": W["dtype"], "data": "synthetic",
            "config": {"workload": W["workload"], "voices_per_gpu": V, "block": B, "sample_rate": 44100,
                       "parallelism": "units sharded x%d" % world,
                       "mixdown": {"off": "off"}.get(mixdown, "maxiMix::stereo per block; %s"
                                                     % ("one ncclReduce per %d blocks on the mix queue's stream" % queue.depth
                                                        if queue is not None else ""))},
            "exchange": exchange,
            "rccl_ranks": (world if (queue is not None and world > 1 and comm is not None) else
                           (1 if (queue is not None and comm is not None) else 0)),
            "roofline": roof,
            "kernels": {k: {"ms": round(v["ms"], 5), "launches_per_step": round(v["launches_per_step"], 3)}
                        for k, v in sorted(kernels.items())},
            "step_ms_gpu": round(step_ms_events, 5),
            "realtime_voices_at_44k1": int(value * 1e6 / 44100) if args.workload in ("config2", "config3") else None,
        }
        if local_ms:
            # per-GPU efficiency of the exchange: the identical step (render + local mixdown into the queue) with the reduce
            # replaced by a device copy, timed in this run on every rank (max over ranks), over the step with the reduce
            step_ms = elapsed / args.steps * 1e3
            res["per_gpu_efficiency"] = round(min(1.0, local_ms / step_ms), 4)
            res["step_ms_without_reduce"] = round(local_ms, 5)
            res["value_like_for_like_n1"] = round(W["samples"] / local_ms / 1e3, 1)
            res["efficiency_note"] = ("value_like_for_like_n1 = one GPU running this exact step (mixdown included) without the "
                                      "reduce; value / (n_gpus x value_like_for_like_n1) = per_gpu_efficiency.  The default N=1 line "
                                      "renders the bank WITHOUT the mixdown (BASELINE configs[1] as named)")
        if args.share_gpu:
            res["share_gpu_test"] = True
        if "north_star_bank" in extras:
            res["north_star_bank"] = extras["north_star_bank"]
        if configs:
            res["configs"] = configs
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = W["cpu"]() if W.get("cpu") else None
        if not args.verbose:
            res = compact_line(res, W.get("tag"))
        print(json.dumps(res, separators=(",", ":")), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    main()
