"""pytest configuration: registers the `gpu` marker and shared helpers.

`-m "not gpu"` runs on any CPU box (oracle vs goldens, host logic, C-ABI symbol
check, gloo world_size-2 partition tests).  `-m gpu` tests are the parity tests
proper: they call the HIP path through the C-ABI on cuda:0 and compare with the
oracle.  Nothing here reads /root/reference at run time; tests that compare the
oracle with the real reference object code skip when oracle/_ref is absent.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build the CPU-side helper libraries if they are missing (seconds, gcc only)."""
    need = [
        (os.path.join(ROOT, "oracle", "libcpu_ref.so"), ["make", "-C", os.path.join(ROOT, "oracle"), "libcpu_ref.so"]),
        (os.path.join(ROOT, "navierstokes_amd", "csrc", "libsynthcsr.so"),
         ["make", "-C", os.path.join(ROOT, "navierstokes_amd", "csrc"), "libsynthcsr.so"]),
    ]
    for path, cmd in need:
        if not os.path.exists(path):
            subprocess.check_call(cmd)


_ensure_built()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


def bits(a):
    """uint64 view for bit-exact comparisons with a readable failure message."""
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = np.nonzero(bits(a) != bits(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} entries differ bitwise, first at {bad[:5]}: {a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}"
