"""CPU (-m "not gpu"): the per-sample engine of include/maximilian.h -- block prediction, derived arguments, rewinds, lock-step groups,
zero-copy renders -- as LOGIC, on the host: tests/host_ps_engine.cpp stubs the handful of C-ABI entry points the engine itself calls
(memory, copies, streams, events) with host memory and renders a simple recurrence on the CPU in a test pool.  Eight objects are called
once per sample with arguments computed from what the earlier calls returned (constants, x*a+b, x1+x2, (x+b)*a, x1*x2, a constant that
is no short decimal, a non-linear map, a parameter that changes now and then, a one-call perturbation every 777 frames); every returned
value must equal the call-by-call evaluation bit for bit, with and without the two mechanisms, and the derivable objects must be served
from predicted blocks.  (The product pools render through libmaxigpu.so; tests/test_gpu_dropin.py checks them against the reference.)"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

FRAMES = 30000


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ps") / "host_ps_engine")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_ps_engine.cpp")])
    return exe


def run(exe, **env):
    r = subprocess.run([exe, str(FRAMES)], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    m = re.search(r"(\d+) frames, (\d+) calls, (\d+) mismatches; (\d+) renders \((\d+) blocks with derived arguments", r.stdout)
    assert m, r.stdout + r.stderr
    misses = {k: int(v) for k, v in re.findall(r"^  (\w): (\d+) of", r.stdout, re.M)}
    return r.returncode, int(m.group(3)), int(m.group(4)), int(m.group(5)), misses, r.stdout


def test_derived_arguments_are_predicted_and_every_value_is_exact(harness):
    rc, bad, renders, derived, misses, out = run(harness)
    assert rc == 0 and bad == 0, out
    assert derived > 100 and renders < 2 * FRAMES, out            # D (the non-linear map) alone costs one render per call
    assert misses["D"] == FRAMES, out
    for k in "ABCEFfG":
        assert misses[k] < FRAMES // 10, out                       # served from predicted blocks


def test_same_values_without_the_prediction_and_without_zero_copy(harness):
    rc, bad, renders, derived, misses, out = run(harness, MXG_PS_DERIVE="0")
    assert rc == 0 and bad == 0 and derived == 0, out
    assert renders > 6 * FRAMES, out                               # "the same arguments as the last call" only: a render per call
    rc, bad, renders, derived, misses, out = run(harness, MXG_PS_ZEROCOPY="0")
    assert rc == 0 and bad == 0 and derived > 100, out


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13, 21, 34])
def test_random_patches_with_setters_replacements_and_perturbations(harness, seed):
    """`fuzz`: a random graph (every argument a constant, a form of earlier objects' outputs, or a sine of one), with random events --
    a one-call perturbation, a changed constant, a setter (settle + state edit), an object destroyed and replaced while other objects'
    forms still name it, a changed recipe -- every returned value bit-identical to the call-by-call evaluation."""
    r = subprocess.run([harness, "fuzz", "12000", str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-1500:]
