"""CPU (-m "not gpu"): error bound of the device's sin/cos for maxiOsc::sinewave / coswave (mxg_sincos.h) measured on
the host over the whole fast-path domain (zero crossings included) against quad-precision references and against glibc's sin/cos."""
import os
import subprocess

import pytest

from conftest import ROOT


def test_sincos_error_bound_on_host(tmp_path):
    if " fma " not in open("/proc/cpuinfo").read():
        pytest.skip("the host build of the kernels relies on FMA contraction (-mfma)")
    exe = str(tmp_path / "sincos_acc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=fast",
                           "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_sincos_accuracy.cpp"), "-lquadmath"])
    r = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "vs glibc sin/cos: max 1 ULP" in r.stdout or "vs glibc sin/cos: max 0 ULP" in r.stdout, r.stdout


def test_log_error_bound_on_host(tmp_path):
    """mxg_log.h (the log of maxiMFCC's log(mb*mb) on the device): < 0.9 ULP against quad precision, <= 1 ULP from glibc's
    log, over the band-energy range, next to 1, across the sqrt(1/2) seam and in every binade; zero / negative / subnormal /
    Inf / NaN take the generic routine."""
    if " fma " not in open("/proc/cpuinfo").read():
        pytest.skip("the host build uses hardware FMA (-mfma)")
    exe = str(tmp_path / "log_acc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-mfma", "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_log_accuracy.cpp"), "-lquadmath"])
    r = subprocess.run([exe, "3000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
