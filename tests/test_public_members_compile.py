"""CPU (-m "not gpu"): the drop-in header's class surface.  tests/patches/public_members_patch.cpp touches every public member of the
hot-path classes of src/maximilian.h (maxiOsc 169-215, maxiFilter 289-366, maxiSample 602-783, maxiEnv 888-932) and uses them as
value types; it must compile against include/maximilian.h -- and, where the reference is mounted, against the reference itself, which
is what makes it a statement about the REFERENCE's API rather than about ours.  (The GPU suite runs it both ways and compares.)"""
import os
import subprocess

import pytest

from conftest import ROOT

PATCH = os.path.join(ROOT, "tests", "patches", "public_members_patch.cpp")


def test_public_members_patch_compiles_against_the_drop_in_header(tmp_path):
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-I" + os.path.join(ROOT, "include"), PATCH],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


def test_public_members_patch_is_valid_reference_code():
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("the reference is not mounted here")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + ref, PATCH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


def test_no_exceptions_build_of_the_header(tmp_path):
    """-DMAXIGPU_NO_EXCEPTIONS (a host whose audio thread cannot unwind): the header compiles with -fno-exceptions."""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-fno-exceptions", "-DMAXIGPU_NO_EXCEPTIONS", "-Wno-unused-variable",
                        "-I" + os.path.join(ROOT, "include"), PATCH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
