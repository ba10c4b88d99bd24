"""CPU (-m "not gpu"): the host builds of the device headers (mxg_env.h, mxg_envgen.h, mxg_osc.h, mxg_smp.h, mxg_sched.h /
mxg_advance.h -- plain arithmetic shared with the kernels) once more under UndefinedBehaviorSanitizer."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["test_env_host.py", "test_envgen_host.py", "test_osc_host.py", "test_smp_host.py", "test_sched_host.py"]


def test_host_harnesses_are_ubsan_clean():
    if os.environ.get("MXG_HOST_UBSAN"):
        return  # (we are the inner run)
    env = dict(os.environ, MXG_HOST_UBSAN="1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", s) for s in SUITES],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
