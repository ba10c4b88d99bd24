#!/usr/bin/env python3
"""oracle/gen_golden_r3.py -- TEST INFRASTRUCTURE.  Golden streams of the round-3 drop-in patches (tests/patches/granular_patch.cpp,
sampler_zx_patch.cpp, convolve_sampler_patch.cpp) compiled against the UNMODIFIED reference (oracle/Makefile _ref/example_p2..p4,
only where /root/reference exists) and run through oracle/example_host.cpp -> tests/golden/dropin_r3.npz (+ MANIFEST entry).
The GPU tests (tests/test_gpu_dropin.py) run the same source files compiled against the drop-in headers and compare."""
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
FRAMES = {"p2": 24000, "p3": 24000, "p4": 24000}


def main():
    subprocess.check_call(["make", "-C", HERE, "_ref/example_p2", "_ref/example_p3", "_ref/example_p4"])
    d = {}
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(os.path.join(GOLD, "wav", "mono.wav"), os.path.join(td, "mono.wav"))
        for tag, frames in FRAMES.items():
            out = os.path.join(td, tag + ".f64")
            subprocess.run([os.path.join(HERE, "_ref", "example_" + tag), str(frames), out], check=True, cwd=td,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            a = np.fromfile(out, np.float64).reshape(frames, 2)
            d["ex" + tag + "_l"], d["ex" + tag + "_r"] = a[:, 0].copy(), a[:, 1].copy()
    np.savez_compressed(os.path.join(GOLD, "dropin_r3.npz"), **d)
    man_path = os.path.join(GOLD, "MANIFEST.json")
    man = json.load(open(man_path))
    sha = hashlib.sha256()
    for f in ("tests/patches/granular_patch.cpp", "tests/patches/sampler_zx_patch.cpp", "tests/patches/convolve_sampler_patch.cpp"):
        sha.update(open(os.path.join(os.path.dirname(HERE), f), "rb").read())
    for f in ("src/libs/maxiGrains.h", "src/libs/maxiConvolve.cpp", "src/libs/maxiSynths.cpp", "src/maximilian.cpp", "src/maximilian.h"):
        sha.update(open(os.path.join("/root/reference", f), "rb").read())
    man.setdefault("files", {})["dropin_r3.npz"] = (
        "tests/patches/granular_patch.cpp (maxiTimeStretch / maxiPitchShift / maxiStretch + maxiOsc::noise sharing rand()), sampler_zx_patch.cpp "
        "(maxiSample playOnZX family, playWithPhasor, FromPos, normalise, loopRecord, operator=) and convolve_sampler_patch.cpp (maxiConvolve + "
        "maxiSampler over tests/golden/wav/mono.wav) compiled with the unmodified reference sources and run through oracle/example_host.cpp: "
        "24000 frames, both channels; sha256 of patches + reference sources " + sha.hexdigest())
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLD, "dropin_r3.npz"), {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
