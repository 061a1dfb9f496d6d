"""Sanitizer builds (race detection / memory checks of the reference's world are PhysX-internal; here they are
ours to run).  The CPU oracle and the host build of the per-sample device functions (tests/hostemu: the very templates the GPU
kernels instantiate) are compiled with AddressSanitizer + UndefinedBehaviourSanitizer and the known-answer, golden and
host-emulation parity tests run against those builds in a subprocess that has the sanitizer runtime preloaded.  Any report
aborts the subprocess (-fno-sanitize-recover, ASAN halt_on_error).  The GPU-side counterpart is the `check` build of the
library (tests/test_gpu_check_build.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sanitizer_env():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so here")
    return dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ORACLE_VARIANT="asan", HOSTEMU_VARIANT="asan", OMP_NUM_THREADS="2")


def test_oracle_and_host_emulation_under_asan_ubsan():
    env = sanitizer_env()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostemu"), "-s", "-j3", "asan"], check=True)
    # the oracle's known-answer tests, the contact-scene tests (oracle + host emulation step by step), the golden boundary
    # fixtures and the host-emulation parity tests - the sanitised libraries are picked up through the two *_VARIANT variables
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
                          "tests/test_oracle_kat.py", "tests/test_scene_kat.py", "tests/test_hostemu_parity.py", "tests/test_golden_boundary.py",
                          "tests/test_more_robots.py"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=3000)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in out.stdout
