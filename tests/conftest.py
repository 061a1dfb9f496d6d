import os
import subprocess
import sys

import pytest

# tests/test_gpu_mailbox.py runs eight shard contexts on eight streams of ONE device with kernels that wait for each other: every
# stream needs a hardware queue of its own (the HIP runtime multiplexes streams onto 4 by default and reads this at start-up)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle64():
    from oracle.oracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.oracle import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def hostemu():
    """test-only g++ build of the per-sample device functions (tests/hostemu/hostemu.cpp)."""
    import ctypes as C
    d = os.path.join(ROOT, "tests", "hostemu")
    variant = os.environ.get("HOSTEMU_VARIANT", "")   # "asan": tests/test_sanitizers.py
    subprocess.run(["make", "-C", d, "-s", "-j3"] + ([variant] if variant else []), check=True)
    lib = C.CDLL(os.path.join(d, f"libhostemu{'_' + variant if variant else ''}.so"))
    lib.emu_cost.restype = C.c_float
    return lib
