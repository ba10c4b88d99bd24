"""CPU (-m "not gpu"): the exact multi-step forms of the granular scheduler (maximilian_amd/csrc/mxg_advance.h, used by
K8a on the device) compiled for the host and fuzzed against the reference's step-by-step recurrences -- hundreds of
thousands of (start, rate, limit) triples incl. rates below half an ulp of the position, starts on the limit, ties and
binade crossings; and next_birth against the sample-by-sample `floor(fmod(counter, cycle)) == 0` walk."""
import os
import subprocess

from conftest import HOST_OPT, ROOT


def test_exact_multi_step_scheduler_forms_on_host(tmp_path):
    exe = str(tmp_path / "sched_fuzz")
    subprocess.check_call(["g++", "-std=c++17"] + HOST_OPT + ["-ffp-contract=off", "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "host_sched_fuzz.cpp")])
    r = subprocess.run([exe, "400000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 mismatches" in r.stdout and "next_birth: 100000 cases, 0 mismatches" in r.stdout, r.stdout


def test_event_driven_walk_equals_the_one_sample_step_on_host(tmp_path):
    """sched_run_events (what K8a runs per stream, mxg_sched.h) against sched_step over random streams of all four modes:
    same births at the same samples with the same (pos0, inc) bits and the same final scheduler state."""
    exe = str(tmp_path / "sched_events")
    subprocess.check_call(["g++", "-std=c++17"] + HOST_OPT + ["-ffp-contract=off", "-I" + os.path.join(ROOT, "maximilian_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "host_sched_events.cpp")])
    r = subprocess.run([exe, "15000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatches 0 0 0 0" in r.stdout, r.stdout
