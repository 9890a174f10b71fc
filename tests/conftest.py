import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def emu():
    import ctypes as C
    san = os.environ.get("REDIO_ORACLE_SAN") == "1"  # tests/san_check.sh: the AddressSanitizer + UBSan build of the lane programs
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-s"] + (["SAN=1"] if san else []))
    return C.CDLL(os.path.join(ROOT, "tests", "_build", "libemu_san.so" if san else "libemu.so"))


@pytest.fixture(scope="session")
def redio():
    """The product library.  On the GPU box it must already be built in-tree; no fallback."""
    import libredio_amd as R
    if not os.path.exists(R.LIBREDIO):
        R.build()
    R.lib()
    return R


@pytest.fixture(scope="session")
def gpu(redio):
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    torch.cuda.set_device(0)
    return torch
