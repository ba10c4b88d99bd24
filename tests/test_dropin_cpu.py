"""CPU (-m "not gpu"): the drop-in boundary's build contract.  Where /root/reference exists, the reference's example
patches must compile UNMODIFIED against include/maximilian.h (host/Makefile uses the files where they lie), and the
oracle executables built from the same files + the unmodified reference library must reproduce tests/golden/dropin.npz;
the config-1 patch (cpp/commandline/main.cpp: 1x maxiOsc::sinewave, 44 100 frames) also pins SURVEY 8a's known answers."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"
EXAMPLES = {"01": "cpp/commandline/main.cpp", "14": "cpp/commandline/maximilian_examples/14.monosynth/main.cpp",
            "15": "cpp/commandline/maximilian_examples/15.polysynth/main.cpp"}


@pytest.mark.parametrize("ex", sorted(EXAMPLES))
def test_reference_examples_compile_verbatim_against_dropin_header(ex, tmp_path):
    src = os.path.join(REF, EXAMPLES[ex])
    if not os.path.exists(src):
        pytest.skip("/root/reference not present")
    obj = str(tmp_path / "patch.o")
    # the patch file itself, untouched; only the include path decides which maximilian.h it sees
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c", "-I" + os.path.join(ROOT, "include"), "-o", obj, src])
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    assert " T setup()" in syms and " T play(double*)" in syms
    assert "mxg_" in syms       # it reaches the C-ABI, not the reference's arithmetic


@pytest.mark.parametrize("ex,frames", [("01", 44100), ("14", 96000), ("15", 16384)])
def test_oracle_examples_reproduce_golden(ex, frames, golden, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "example_" + ex)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/example_%s not built (needs /root/reference)" % ex)
    out = str(tmp_path / "o.f64")
    r = subprocess.run([exe, str(frames), out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "Msamples/s" in r.stderr
    got = np.fromfile(out, np.float64).reshape(frames, 2)
    assert np.array_equal(got.view(np.uint64), golden("dropin.npz")["ex" + ex].view(np.uint64))


def test_config1_known_answers(golden):
    """samples 0..3 of sinewave(440) with g++ 11 / glibc of this image."""
    ex01 = golden("dropin.npz")["ex01"]
    assert ex01[:4, 0].tolist() == [0.0, 0.062648324178743678, 0.1250505236945281, 0.18696144082725336]
    assert np.array_equal(ex01[:, 0], ex01[:, 1])


def test_device_failure_is_one_printed_line_and_silence(tmp_path):
    """Without a HIP device (this container) a patch built against the drop-in header must behave like the reference after its own one
    runtime complaint (printf("ERROR: Could not load sample."), src/maximilian.cpp:686): ONE line on stderr, no exception, no abort,
    every unit generator returns silence -- and nothing is computed on the CPU instead (the product has no fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the device path works here")
    exe = os.path.join(ROOT, "host", "dropin_01")
    src = os.path.join(REF, EXAMPLES["01"])
    if os.path.exists(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "dropin_01"], stdout=subprocess.DEVNULL)
    if not os.path.exists(exe):
        pytest.skip("host/dropin_01 not built (needs /root/reference)")
    out = str(tmp_path / "o.f64")
    r = subprocess.run([exe, "2000", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("ERROR: maxigpu") == 1 and "no CPU fallback" in r.stderr
    got = np.fromfile(out, np.float64)
    assert got.size == 4000 and not got.any(), "silence, not a CPU rendering"


@pytest.mark.parametrize("flags", [[], ["-DMAXIGPU_THROW"], ["-fno-exceptions", "-DMAXIGPU_NO_EXCEPTIONS"]])
def test_header_builds_in_every_failure_mode(flags):
    for patch in ("public_members_patch.cpp", "convolve_sampler_patch.cpp", "granular_patch.cpp", "refused_call_patch.cpp"):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wno-unused-variable", "-I" + os.path.join(ROOT, "include")] + flags +
                           [os.path.join(ROOT, "tests", "patches", patch)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
