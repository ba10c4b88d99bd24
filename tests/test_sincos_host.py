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
