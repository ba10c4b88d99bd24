#!/usr/bin/env python3
"""oracle/gen_golden_r4.py -- TEST INFRASTRUCTURE.  Golden stream of the round-4 drop-in patch tests/patches/public_members_patch.cpp
(every public member of maxiOsc / maxiFilter / maxiSample / maxiEnv; the classes used as value types) compiled against the UNMODIFIED
reference (oracle/Makefile _ref/example_p5, only where /root/reference exists) and run through oracle/example_host.cpp ->
tests/golden/dropin_r4.npz (+ MANIFEST entry).  tests/test_gpu_dropin.py runs the same source compiled against include/maximilian.h."""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
FRAMES = 24000


def main():
    subprocess.check_call(["make", "-C", HERE, "_ref/example_p5"])
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "p5.f64")
        subprocess.run([os.path.join(HERE, "_ref", "example_p5"), str(FRAMES), out], check=True, cwd=td, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        a = np.fromfile(out, np.float64).reshape(FRAMES, 2)
    np.savez_compressed(os.path.join(GOLD, "dropin_r4.npz"), exp5_l=a[:, 0].copy(), exp5_r=a[:, 1].copy())
    man_path = os.path.join(GOLD, "MANIFEST.json")
    man = json.load(open(man_path))
    sha = hashlib.sha256()
    sha.update(open(os.path.join(os.path.dirname(HERE), "tests/patches/public_members_patch.cpp"), "rb").read())
    for f in ("src/maximilian.cpp", "src/maximilian.h"):
        sha.update(open(os.path.join("/root/reference", f), "rb").read())
    man.setdefault("files", {})["dropin_r4.npz"] = (
        "tests/patches/public_members_patch.cpp (every public member of maxiOsc / maxiFilter / maxiSample / maxiEnv, objects in std::vector, "
        "copy construction and assignment, state members read and written) compiled with the unmodified reference sources and run through "
        "oracle/example_host.cpp: 24000 frames, both channels; sha256 of patch + reference sources " + sha.hexdigest())
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
    print("wrote dropin_r4.npz", a.shape, "peak", np.abs(a).max(axis=0), "finite", np.isfinite(a).all())


if __name__ == "__main__":
    main()
