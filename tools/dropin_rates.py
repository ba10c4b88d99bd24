#!/usr/bin/env python3
"""tools/dropin_rates.py -- real-time factor of the DROP-IN boundary: every host/dropin_* binary (a reference patch compiled verbatim
against include/maximilian.h, one C++ call per sample, blocks rendered on the GPU behind it) is run for a few seconds of audio;
frames per second, launches per frame and the ratio to 44 100 frames/s are tabulated next to the same patch linked with the
reference itself (oracle/_ref/example_*: the compiled reference on this host's CPU, present where /root/reference was at build time).
Output: markdown (stdout and --out)."""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"01": "cpp/commandline/main.cpp (sinewave)", "02": "2.TwoTones", "03": "3.AM1", "04": "4.AM2", "05": "5.FM1", "06": "6.FM2",
         "08b": "8.Counting2", "08c": "8.Counting3", "08d": "8.Counting4", "10": "10.Filters", "11": "11.Mixing", "12": "12.SamplePlayer",
         "13": "13.Advanced-Filters", "14": "14.monosynth", "15": "15.polysynth", "16": "16.Replicant", "20": "20.FFT_example",
         "p1": "tests/patches/filters2_patch", "p2": "tests/patches/granular_patch", "p3": "tests/patches/sampler_zx_patch",
         "p4": "tests/patches/convolve_sampler_patch", "tfft": "tests/ffttest", "tmfcc": "tests/mfcctest", "tsvf": "tests/svftest"}
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=88200)
ap.add_argument("--out", default=None)
args = ap.parse_args()
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


emit("# Drop-in boundary: frames per second of reference patches compiled against include/maximilian.h (one call per sample)")
emit()
emit("`python tools/dropin_rates.py --frames %d`: wall time of the SECOND HALF of the render loop (mxg_host_render, the loop of cpp/commandline/player.cpp:25-44); "
     "real time = 44 100 frames/s; launches = block renders the per-sample engine issued (osc + env + filter pools, from the binary's own "
     "counters; other pools not counted)." % args.frames)
emit()
emit("| patch | drop-in frames/s | x real time | launches / 1000 frames | reference on the host CPU, frames/s | drop-in / reference |")
emit("|---|---|---|---|---|---|")
with tempfile.TemporaryDirectory() as td:
    cwd = os.path.join(td, "a", "b", "c")
    os.makedirs(cwd)
    wav = os.path.join(ROOT, "tests", "golden", "wav", "mono.wav")
    shutil.copy(wav, os.path.join(td, "beat2.wav"))
    shutil.copy(wav, os.path.join(cwd, "mono.wav"))
    for tag, name in NAMES.items():
        exe = os.path.join(ROOT, "host", "dropin_" + tag)
        if not os.path.exists(exe):
            continue
        out = os.path.join(td, "o.f64")
        frames = args.frames if tag not in ("10", "13", "14", "p3", "20", "tfft", "tsvf") else max(4410, args.frames // 8)
        try:
            r = subprocess.run([exe, str(frames), out], capture_output=True, text=True, timeout=600, cwd=cwd)
        except subprocess.TimeoutExpired:
            emit("| %s | timed out | | | | |" % name)
            continue
        m = re.search(r"rendered (\d+) frames x \d+ channels in ([0-9.]+) s; launches: osc (\d+) .*?env (\d+) .*?filter (\d+)", r.stderr)
        if r.returncode != 0 or not m:
            emit("| %s | failed (rc %d) | | | | |" % (name, r.returncode))
            continue
        secs = max(float(m.group(2)), 1e-9)
        fps = frames / secs
        ms_ = re.search(r"steady state: (\d+) frames in ([0-9.]+) s", r.stderr)
        if ms_:  # the second half of the run by itself (the first launches pay for the HIP runtime's start-up)
            fps = int(ms_.group(1)) / max(float(ms_.group(2)), 1e-9)
        launches = int(m.group(3)) + int(m.group(4)) + int(m.group(5))
        ref = os.path.join(ROOT, "oracle", "_ref", "example_" + tag)
        rfps = None
        if os.path.exists(ref):
            rr = subprocess.run([ref, str(frames), out], capture_output=True, text=True, timeout=600, cwd=cwd)
            mm = re.search(r"in ([0-9.]+) s", rr.stderr)
            if rr.returncode == 0 and mm:
                rfps = frames / max(float(mm.group(1)), 1e-9)
        emit("| %s | %.0f | %.2f | %.1f | %s | %s |" % (name, fps, fps / 44100.0, 1000.0 * launches / frames,
                                                    "%.0f" % rfps if rfps else "-", "%.4f" % (fps / rfps) if rfps else "-"))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
