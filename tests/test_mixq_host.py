"""CPU (-m "not gpu"): the mix queue's slot / push / submit / flush / release protocol -- the text the product compiles for
HIP + RCCL (maximilian_amd/csrc/mxg_mixq_core.h, instantiated by comm.hip's mxg_mixq) -- built for the host against a
simulated asynchronous device and driven by TWO ranks with many batches in flight (tests/host_mixq.cpp): streams are worker
threads with random delays, events have hipEvent semantics, the reduce is a two-rank rendezvous that reads its send buffers
slowly.  Rank 0 must see the sum over both ranks for every block through every way of reading a result.

Two mutants of the same header (one event wait compiled out each) must FAIL the same harness: the staging buffer refilled
under a running reduce, and a result overwritten under a declared asynchronous read."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path, name, defs=()):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-Wall"] + list(defs) +
                          ["-I" + os.path.join(ROOT, "maximilian_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "host_mixq.cpp")])
    return exe


def test_mix_queue_protocol_two_ranks_many_batches(tmp_path):
    r = subprocess.run([_build(tmp_path, "mixq"), "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "108 cases, 0 failed" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("mutant", ["MXG_MIXQ_MUTATE_NO_SLOT_WAIT", "MXG_MIXQ_MUTATE_NO_CONSUMED_WAIT"])
def test_mix_queue_harness_catches_a_missing_wait(tmp_path, mutant):
    exe = _build(tmp_path, "mixq_mut", ["-D" + mutant])
    # the mutants are RACES: whether one shows in a given run depends on the worker threads' timing (a loaded machine can hide
    # it for 108 cases).  The harness has to catch it within a few attempts of growing length, not necessarily in the first.
    out = ""
    for rounds in ("12", "24", "48", "96"):
        r = subprocess.run([exe, rounds], capture_output=True, text=True, timeout=600)
        out = r.stdout
        if r.returncode != 0:
            return
    raise AssertionError("the harness did not notice the missing wait (%s) in four attempts\n%s" % (mutant, out[-2000:]))
